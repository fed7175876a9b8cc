"""MI355X-native mirror of the reference's `MerlotModel` surface (model/modeling.py:47-668).

Same constructor arguments, attributes and methods as the reference class -- `MerlotModel(config, is_training,
use_tpu, image, input_ids, mask_input, shuffled_idx_img, img_mask, log_attention_probs)`, `.mask_loss()`,
`.contrastive_loss()`, `.temporal_loss()`, `.allpairs_temporal_logits()`, `.lm_head()`, `.embed_words()`,
`encoder_hidden_states`, `lang_mask_info`, `attention_log`, `B/L/P/...` -- plus three keyword-only extras the TF
graph got implicitly: `params` (the ParamStore standing in for TF's variable store), `noise` (the explicit random
draws, so integer outputs are reproducible bit for bit) and `dist` (the DP context for the in-batch all-gather).

All dense math runs in the hand-written gfx950 kernels of libmerlot_hip.so via merlot_amd.layers; torch is used
for device memory, views/concats and the autograd tape only.  dtype policy = the reference's bf16 policy
(utils/model_utils.py:572-602): fp32 master weights cast to bf16 at use, bf16 residual stream, fp32 LayerNorm
statistics / embedding sums / heads / losses.
"""
import copy
import math

import numpy as np
import torch

from . import layers as L
from . import ops
from .layers import StackW
from .ops import BF16, F32
from .params import ParamStore

MASK = 1          # utils/encode/encoder.py:17
PADDING = 0


def masking_constants(Lseq, config):
    """model/modeling.py:390-419, python-double arithmetic then float32 exactly as the TF graph does."""
    topk_perc = config.get('masking_use_topk_from_attn_perc', 0.20)
    choose_topk_prob = config.get('masking_choose_topk_prob', 0.5)
    masking_rate = config.get('masking_rate', 0.2)
    use_attn = config.get('masking_use_attn', True)
    num_topk = int(Lseq * topk_perc)
    num_to_mask = int(Lseq * masking_rate)
    nontopk_val = 0.01
    topk_val = nontopk_val * choose_topk_prob * (1.0 - topk_perc) / (topk_perc * (1.0 - choose_topk_prob))
    if use_attn:
        w_non = np.float32(nontopk_val)
        w_top = np.float32(1.0) * np.float32(topk_val - nontopk_val) + w_non
        max_w = max(w_top, w_non) if num_topk > 0 else w_non
    else:
        w_non = w_top = max_w = np.float32(1.0)
    return dict(num_topk=num_topk, num_to_mask=num_to_mask, use_attn=use_attn,
                do_spanbert=config.get('masking_do_spanbert', True),
                spanbert_len_probs=config.get('masking_spanbert_len_probs', [0.625, 0.25, 0.125]),
                w_nontopk=float(w_non), w_topk=float(w_top), max_weight=float(max_w),
                log_nontopk=float(np.log(w_non).astype(np.float32)), log_topk=float(np.log(w_top).astype(np.float32)))


def draw_mask_noise(B, Lseq, config, vocab_size, generator):
    """The random draws of mask_inputs (model/modeling.py:445-481, utils/model_utils.py:647) as explicit tensors."""
    c = masking_constants(Lseq, config)
    nm = c['num_to_mask']
    u = torch.rand((B, Lseq), generator=generator, dtype=torch.float64).clamp_(1e-12, 1 - 1e-12)
    probs = torch.tensor(c['spanbert_len_probs'], dtype=torch.float64)
    return {
        'gumbel': (-torch.log(-torch.log(u))).float(),
        'span_lower': torch.multinomial(probs, B * nm, replacement=True, generator=generator).view(B, nm).int(),
        'span_upper': torch.multinomial(probs, B * nm, replacement=True, generator=generator).view(B, nm).int(),
        'random_ids': torch.randint(100, vocab_size, (B * Lseq,), generator=generator).int(),
        'option': torch.multinomial(torch.tensor([0.1, 0.8, 0.1], dtype=torch.float64), B * Lseq, replacement=True,
                                    generator=generator).int(),
    }


class LocalDist(object):
    """Single-replica stand-in for the DP context (utils/model_utils.py:680-683: `return tensor[None], 0`)."""
    rank = 0
    world_size = 1

    def all_gather_cat(self, x):
        return x


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout_apply(x.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout_apply(dy.contiguous(), ctx.p, ctx.seed), None, None


class MerlotModel(object):
    def __init__(self, config, is_training, use_tpu, image, input_ids, mask_input=False, shuffled_idx_img=None,
                 img_mask=None, log_attention_probs=True, *, params=None, noise=None, dist=None, seed=0):
        self.config = copy.deepcopy(dict(config))
        cfg = self.config
        self.is_training = is_training
        self.use_tpu = use_tpu
        if params is None:
            raise ValueError("MerlotModel needs `params` (a merlot_amd.ParamStore): it replaces TF's variable store")
        self.params = params
        self.dist = dist if dist is not None else LocalDist()
        self.seed = int(seed)
        dev = params.device
        self.device = dev

        if cfg.get('resnet_layers') and cfg['patch_size'] != 16:
            raise ValueError("the ResNet-hybrid stem reduces by 16 (utils/vision_transformer.py:208)")
        if cfg.get('num_imgs', 1) != 1 or cfg.get('num_texts', 1) != 1:
            raise NotImplementedError("num_imgs / num_texts > 1 (VCR path) is out of scope")

        input_ids = torch.as_tensor(input_ids).to(dev)
        if input_ids.dim() == 2:                                                  # modeling.py:72-77
            self.num_chunks = 1
            self.num_chunks_in_group = 1
            self.batch_size, self.lang_chunk_length = input_ids.shape
            self.input_ids = input_ids[:, None].int().contiguous()
        elif input_ids.dim() == 3:                                                # :78-82
            self.input_ids = input_ids.int().contiguous()
            self.batch_size, self.num_chunks, self.lang_chunk_length = input_ids.shape
            self.num_chunks_in_group = cfg.get('num_chunks_in_group', self.num_chunks)
            if self.num_chunks % self.num_chunks_in_group != 0:
                raise ValueError("num_chunks must be a multiple of num_chunks_in_group")
        else:
            raise ValueError(f"input_ids must have rank 2 or 3, got shape {tuple(input_ids.shape)}")
        self.num_imgs = 1
        self.num_texts = 1
        self.img_batch_size = self.batch_size
        if not is_training:                                                       # :88-90
            cfg['hidden_dropout_prob'] = 0.0
            cfg['attention_probs_dropout_prob'] = 0.0
        if cfg.get('attention_probs_dropout_prob', 0.0) > 0.0:
            raise NotImplementedError("attention-probability dropout > 0 is not supported (merlot.yaml uses 0.0)")

        H = self.hidden_size
        heads = cfg['num_attention_heads']
        if H != heads * 64:
            raise ValueError("passed in hidden_size={} when size_per_head=64 and num_attention_heads={}".format(H, heads))
        st = params
        st.refresh()
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._token_id_flags = []                            # device bools: embed_words saw an id outside [0, vocab)
        nl_vit = cfg.get('num_vision_transformer_hidden_layers', cfg['num_hidden_layers'])
        nl_enc = max(cfg['num_hidden_layers'], cfg.get('num_lang_transformer_hidden_layers', 0))
        self._vit = StackW(st, 'vision_backbone/vision_transformer', nl_vit)
        if cfg.get('share_params', True):                    # one stack, used by the text-only AND the joint pass
            self._enc = StackW(st, 'encoder', nl_enc)
            self._lang_enc = self._enc
        else:                                                # model/modeling.py:357-362: separate `langonly_encoder`
            self._enc = StackW(st, 'encoder', cfg['num_hidden_layers'])
            self._lang_enc = StackW(st, 'langonly_encoder', cfg['num_lang_transformer_hidden_layers'])
        self.word_embedding_table = st.p('word_embeddings/word_embeddings')
        self._emb = st.lin('word_embeddings/word_embeddings', need_T=True, bias=False)

        # ---------------- vision half (modeling.py:94-133, utils/vision_transformer.py:173-274)
        image = torch.as_tensor(image).to(dev)
        if image.dim() != 4 or image.shape[-1] != 3:
            raise ValueError("image must be [batch*num_chunks, h, w, 3]")
        if image.dtype != BF16:
            image = image.to(BF16)
        image = image.contiguous()
        N, h0, w0, _ = image.shape
        if N != self.batch_size * self.num_chunks:
            raise ValueError(f"image batch {N} != batch_size*num_chunks {self.batch_size * self.num_chunks}")
        Pz = cfg['patch_size']
        assert h0 % Pz == 0 and w0 % Pz == 0
        h1, w1 = h0 // Pz, w0 // Pz
        ncls = cfg.get('num_cls_emb', 2)
        Sv = h1 * w1 + ncls
        vs = 'vision_backbone/vision_transformer'
        if cfg.get('resnet_layers'):                                              # utils/vision_transformer.py:206-223
            conv = L.ResNetStemFn.apply(image, st, cfg, self._anchor)
        else:
            conv = L.PatchEmbedFn.apply(image, st.lin(f'{vs}/conv2d', need_T=False), Pz, self._anchor)
        idx_conv, idx_cls, idx_pos = self._vit_prologue_indices(N, h1, w1, ncls)
        conv_inv = (Sv, ncls)                             # patch j of frame n sits at row n * Sv + ncls + j
        x = L.gather_add(conv, idx_conv,
                         [(st.p(f'{vs}/pos_embs/cls_emb'), st.g(f'{vs}/pos_embs/cls_emb'), idx_cls, Sv),
                          (st.p(f'{vs}/pos_embs/pos_embs'), st.g(f'{vs}/pos_embs/pos_embs'), idx_pos, Sv)],
                         N * Sv, H, self._anchor, act_inv=conv_inv)                # vision_transformer.py:229-233
        x = L.layer_norm(x, st.ln(f'{vs}/LayerNorm_ctx_patches_pre_ln'), out_bf16=True)
        vit_p = cfg.get('vit_hidden_dropout_prob', cfg['hidden_dropout_prob']) if is_training else 0.0
        L.f8_step_begin(st)                               # `fp8_backward`: last step's recorded amaxes become this step's scales (a no-op otherwise)
        hs = L.transformer_stack(x, self._vit, N, Sv, None,
                                 dict(heads=heads, dropout_p=vit_p, seed=self.seed * 4 + 0, num_layers=nl_vit,
                                      fp8=cfg.get('fp8_forward', False), fp8_bwd=cfg.get('fp8_backward', False), f8_site='vit'))
        hs3 = hs.view(N, Sv, H)
        sp = cfg['spatial_pool_size']
        h2, w2 = h1 // sp, w1 // sp
        self.vision_transformer_info = {'hidden_state': hs3, 'cls': hs3[:, :ncls], 'num_h': h2, 'num_w': w2}
        feats, self.img_trg_h = L.ClsAvgPoolFn.apply(hs, N, h1, w1, ncls, sp, 1)  # [N, vl, H] f32 :101-105, hs3[:, 1] f32 :99
        self.vision_transformer_info['seq'] = feats[:, 1:]
        vl = self.viz_chunk_length
        idx_img, idx_fcls, idx_fpos = self._final_pe_indices(N, h2, w2, shuffled_idx_img)
        image_feats = L.gather_add(feats, None,
                                   [(st.p('vision_backbone/img_idx_pe'), st.g('vision_backbone/img_idx_pe'), idx_img),
                                    (st.p('vision_backbone/final_pe/cls_emb'), st.g('vision_backbone/final_pe/cls_emb'), idx_fcls, vl),
                                    (st.p('vision_backbone/final_pe/pos_embs'), st.g('vision_backbone/final_pe/pos_embs'), idx_fpos, vl)],
                                   N * vl, H, self._anchor)                        # :125 + :299-337
        image_feats = L.layer_norm(image_feats, st.ln('vision_backbone/LayerNorm_final_ln'), out_bf16=True)   # :126-128
        if img_mask is None:                                                      # :105-108
            viz_valid = torch.ones((self.B, self.P), device=dev, dtype=torch.bool)
        else:                                                                     # :108, :122 with num_imgs = 1: one flag
            viz_valid = torch.as_tensor(img_mask).to(dev).bool().reshape(self.B, 1).expand(self.B, self.P)   # per group
        self.encoder_pieces = [{'name': 'viz', 'x': image_feats.view(self.B, self.P, H), 'is_valid': viz_valid}]

        # ---------------- language half (:135-149)
        if mask_input:
            self.lang_trg_h, self.lang_transformer_info = self.langonly_reps()
            self.lang_mask_info = self.mask_inputs(noise)
            input_ids_to_use = self.lang_mask_info['masked_ids']
        else:
            input_ids_to_use = self.input_ids
        input_ids_to_use = input_ids_to_use.reshape(self.B, self.L)
        self.encoder_pieces.append({'name': 'lang', 'x': self.embed_words(input_ids_to_use),
                                    'is_valid': input_ids_to_use != 0})

        # ---------------- joint encoder (:151-184)
        encoder_input = torch.cat([p['x'] for p in self.encoder_pieces], 1)
        is_valid = torch.cat([p['is_valid'] for p in self.encoder_pieces], 1)
        Sj = self.P + self.L
        opts = dict(heads=heads, dropout_p=self.dropout_prob if is_training else 0.0, seed=self.seed * 4 + 2,
                    num_layers=cfg['num_hidden_layers'], fp8=cfg.get('fp8_forward', False), fp8_bwd=cfg.get('fp8_backward', False), f8_site='joint')
        if cfg.get('disable_pairwise_lang_attn', False):                          # :160-168, as a segment vector
            opts['seg'] = torch.cat([torch.zeros(self.P, dtype=torch.int32),
                                     1 + torch.arange(self.L, dtype=torch.int32) // self.lang_chunk_length]).to(dev)
        if log_attention_probs:
            log_lo = torch.zeros((self.B, Sj), device=dev, dtype=F32)
            log_hi = torch.zeros((self.B, Sj), device=dev, dtype=F32)
            log_vals = torch.zeros(4, device=dev, dtype=F32)

            P_ = self.P

            def finish_log():                                                     # :186-203 from the fused block sums
                # (closes over tensors and an int only: stored in the encoder's autograd node, a reference to `self` here would be a
                # cycle through the graph that Python's collector cannot see -- one whole step's activations leaked per step)
                tot = log_lo.sum() + log_hi.sum()
                log_vals.copy_(torch.stack([log_hi[:, P_:].sum(), log_lo[:, P_:].sum(), log_hi[:, :P_].sum(), log_lo[:, :P_].sum()]) / tot)
            # `attention_log_in_backward` (an extension key, default off): in a TRAINING step the block sums come out of the
            # attention backward (its dK / dV pass forms P anyway) instead of a second Q K^T walk in each of the joint encoder's
            # forward launches; `attention_log` then holds its values once `loss.backward()` has run (zeros before) -- the
            # reference reads these metrics at the end of the train step too (model/modeling.py:709).  Without a backward
            # (inference, forward-only timing) and by default they are complete right after construction.
            opts.update(log_lo=log_lo, log_hi=log_hi, log_split=self.P, log_done=finish_log,
                        # decided HERE, once (the stack must not silently keep the log in the forward where nobody finishes it)
                        log_in_backward=(bool(cfg.get('attention_log_in_backward', False)) and is_training and torch.is_grad_enabled()))
        enc = L.transformer_stack(encoder_input.reshape(self.B * Sj, H), self._enc, self.B, Sj,
                                  is_valid.to(torch.uint8).contiguous(), opts)
        enc3 = enc.view(self.B, Sj, H)
        self.encoder_info = {'hidden_state': enc3, '_hidden_state_flat': enc}
        cur = 0
        for p in self.encoder_pieces:
            p['start'] = cur
            p['end'] = cur + p['x'].shape[1]
            cur = p['end']
        pieces = L.SplitHiddenFn.apply(enc3, tuple((p['start'], p['end']) for p in self.encoder_pieces))   # :184, f32
        self.encoder_hidden_states = {p['name']: t for p, t in zip(self.encoder_pieces, pieces)}

        if log_attention_probs:
            if not (opts['log_in_backward'] and encoder_input.requires_grad):
                finish_log()
            # sorted keys: lang2lang, lang2viz, viz2lang, viz2viz  (queries -> keys: log_lo = viz queries, log_hi = lang queries)
            self.attention_log = {f'encoder/{k}': log_vals[i] for i, k in enumerate(('lang2lang', 'lang2viz', 'viz2lang', 'viz2viz'))}

    # ------------------------------------------------------------------------------------------------
    # index helpers (host side, cached per shape)
    # ------------------------------------------------------------------------------------------------
    def _vit_prologue_indices(self, N, h1, w1, ncls):
        dev = self.device
        npatch = h1 * w1
        s = torch.arange(ncls + npatch, device=dev)
        is_cls = s < ncls
        g = (s - ncls).clamp(min=0)
        pos = (g // w1) * 64 + (g % w1)                        # rows of pos_embs[0] flattened [64*64, H]
        idx_pos = torch.where(is_cls, torch.full_like(s, -1), pos).repeat(N)
        idx_cls = torch.where(is_cls, s, torch.full_like(s, -1)).repeat(N)
        n = torch.arange(N, device=dev)[:, None]
        idx_conv = torch.where(is_cls[None], torch.full((1, 1), -1, device=dev, dtype=torch.long), n * npatch + g[None])
        return idx_conv.reshape(-1).int().contiguous(), idx_cls.int().contiguous(), idx_pos.int().contiguous()

    def _final_pe_indices(self, N, h2, w2, shuffled_idx_img):
        dev = self.device
        vl = 1 + h2 * w2
        n_in = self.num_chunks_in_group
        if shuffled_idx_img is None:                                              # :312-315
            per_img = torch.arange(N, device=dev) % n_in
        else:                                                                     # :316-323
            per_img = torch.as_tensor(shuffled_idx_img).to(dev).reshape(-1).long()
            if per_img.numel() != N:
                raise ValueError(f"shuffled_idx_img must have {N} entries, got {per_img.numel()}")
        idx_img = per_img[:, None].expand(N, vl).reshape(-1)
        t = torch.arange(vl, device=dev)
        g = (t - 1).clamp(min=0)
        idx_fpos = torch.where(t == 0, torch.full_like(t, -1), (g // w2) * 64 + (g % w2)).repeat(N)
        idx_fcls = torch.where(t == 0, torch.zeros_like(t), torch.full_like(t, -1)).repeat(N)
        return idx_img.int().contiguous(), idx_fcls.int().contiguous(), idx_fpos.int().contiguous()

    # ------------------------------------------------------------------------------------------------
    # shape algebra (modeling.py:226-260)
    # ------------------------------------------------------------------------------------------------
    @property
    def hidden_size(self):
        return self.config['hidden_size']

    @property
    def vocab_size(self):
        return self.config['vocab_size']

    @property
    def B(self):
        return self.batch_size * (self.num_chunks // self.num_chunks_in_group)

    @property
    def L(self):
        return self.lang_chunk_length * self.num_chunks_in_group

    @property
    def viz_chunk_length(self):
        return self.vision_transformer_info['num_h'] * self.vision_transformer_info['num_w'] + 1

    @property
    def P(self):
        return self.viz_chunk_length * self.num_chunks_in_group

    @property
    def dropout_prob(self):
        return self.config['hidden_dropout_prob']

    @property
    def use_bfloat16(self):
        return self.config.get('use_bfloat16', True)

    # ------------------------------------------------------------------------------------------------
    def token_id_flag(self):
        """device bool: any embed_words call of this model saw a token id outside [0, vocab_size) (None if none ran)."""
        if not self._token_id_flags:
            return None
        return torch.stack(self._token_id_flags).any()

    def check_token_ids(self):
        """the range assertion of utils/model_utils.py:256-258 (synchronises with the GPU)."""
        f = self.token_id_flag()
        if f is not None and bool(f):
            raise ValueError("token id out of range")

    def embed_words(self, input_ids_2d, norm_scope_name='position_embeddings'):
        """model/modeling.py:262-297: LN(word_emb[ids] + pos_emb[0:L]) -> dropout -> bf16 [R, L, H]."""
        st = self.params
        H = self.hidden_size
        ids = torch.as_tensor(input_ids_2d).to(self.device)
        R, Lq = ids.shape
        if Lq > self.config['max_position_embeddings']:
            raise ValueError("sequence longer than max_position_embeddings")
        # utils/model_utils.py:256-258 asserts the id range INSIDE the graph (the error surfaces when the step's results are
        # fetched); same here: the flag stays on the device -- no host round trip in the middle of the step, the trainer
        # fetches it one step late (`check_token_ids`) -- and the gather below is made memory-safe by clamping.
        bad = (ids.min() < 0) | (ids.max() > self.vocab_size - 1)
        self._token_id_flags.append(bad)
        if not self.is_training:
            self.check_token_ids()
        idx_w = ids.reshape(-1).clamp(0, self.vocab_size - 1).int().contiguous()
        idx_p = torch.arange(Lq, device=self.device).repeat(R).int().contiguous()
        emb = L.gather_add(None, None,
                           [(st.p('word_embeddings/word_embeddings'), st.g('word_embeddings/word_embeddings'), idx_w),
                            (st.p(f'{norm_scope_name}/position_embeddings'), st.g(f'{norm_scope_name}/position_embeddings'), idx_p, Lq)],
                           R * Lq, H, self._anchor)
        out = L.layer_norm(emb, st.ln(f'{norm_scope_name}/LayerNorm_embed_norm'), out_bf16=True)
        p = self.dropout_prob if self.is_training else 0.0
        if p > 0:
            site = 1 if norm_scope_name == 'position_embeddings' else 2
            out = _DropoutFn.apply(out, p, self.seed * 16 + 8 + site)
        return out.view(R, Lq, H)

    def langonly_reps(self):
        """model/modeling.py:339-379: text-only encoder over the whole transcript; also accumulates the per-key
        attention column sums that mask_inputs consumes (:428-431) instead of stacking [E, layers, S, S]."""
        cfg = self.config
        H = self.hidden_size
        if 'langonly_num_chunks_in_group' in cfg:                                 # :345-351
            g = cfg['langonly_num_chunks_in_group']
            ngroups = self.num_chunks // g
            assert ngroups > 0 and self.num_chunks % g == 0
            ids2d = self.input_ids.reshape(self.batch_size * ngroups, self.lang_chunk_length * g)
        else:
            ids2d = self.input_ids.reshape(self.batch_size, self.lang_chunk_length * self.num_chunks)
        R, Sl = ids2d.shape
        emb = self.embed_words(ids2d, norm_scope_name='langonly_embeddings')
        valid = (ids2d != 0).to(torch.uint8).contiguous()
        summ = torch.zeros((R, Sl), device=self.device, dtype=F32)
        hs = L.transformer_stack(emb.reshape(R * Sl, H), self._lang_enc, R, Sl, valid,
                                 dict(heads=cfg['num_attention_heads'],
                                      dropout_p=self.dropout_prob if self.is_training else 0.0, seed=self.seed * 4 + 1,
                                      num_layers=cfg['num_lang_transformer_hidden_layers'], colsum=summ,
                                      fp8=cfg.get('fp8_forward', False), fp8_bwd=cfg.get('fp8_backward', False), f8_site='lang'))
        pool = hs.view(self.batch_size * self.num_chunks, self.lang_chunk_length, H)[:, 0].float()   # :372-378
        info = {'_hidden_state_flat': hs, 'hidden_state': hs.view(R, Sl, H), 'attention_summs': summ}
        return pool, info

    def mask_inputs(self, noise=None):
        """model/modeling.py:381-489 on the GPU (merlot_mask_inputs); bit-exact given `noise`."""
        cfg = self.config
        ids2d = self.input_ids.reshape(self.B, self.L).contiguous()
        c = masking_constants(self.L, cfg)
        if noise is None:
            gen = torch.Generator().manual_seed(self.seed * 7919 + 17)
            noise = draw_mask_noise(self.B, self.L, cfg, self.vocab_size, gen)
        nz = {k: torch.as_tensor(v).to(self.device).contiguous() for k, v in noise.items()}
        summs = None
        if c['use_attn']:
            summs = self.lang_transformer_info['attention_summs'].reshape(self.B, self.L).contiguous()   # :428-431
        masked_ids, masked_idx = ops.mask_inputs(
            ids2d, summs, nz['gumbel'].float(), nz['span_lower'].int() if c['do_spanbert'] else None,
            nz['span_upper'].int() if c['do_spanbert'] else None, nz['random_ids'].int().reshape(self.B, self.L),
            nz['option'].int().reshape(self.B, self.L), c['num_topk'], c['num_to_mask'], c['w_nontopk'], c['w_topk'],
            c['log_nontopk'], c['log_topk'], c['max_weight'], MASK)
        return {'masked_ids': masked_ids.reshape(self.input_ids.shape), 'masked_idx': masked_idx}

    # ------------------------------------------------------------------------------------------------
    # heads
    # ------------------------------------------------------------------------------------------------
    def lm_head(self, hidden_state):
        """model/modeling.py:205-224 -> logits [T, vocab] (f32).  mask_loss() uses the fused CE form instead."""
        st = self.params
        h = hidden_state
        if self.config.get('do_projection', False):
            h = L.linear(h, st.lin('lm_head/projection'), 'gelu')
            h = L.layer_norm(h, st.ln('lm_head/LayerNorm'))
        logits = L.linear(h, self._emb)
        if self.config.get('do_bias', False):
            logits = logits + self._param_as_leaf('lm_head/output_bias')
        return logits

    def _param_as_leaf(self, name):
        """a differentiable view of a small fp32 parameter for plain-torch glue; its grad is folded into the arena."""
        st = self.params
        p = st.p(name).detach().clone().requires_grad_(True)
        g = st.g(name)
        p.register_hook(lambda gr: (g.add_(gr), None)[1])
        return p

    def mask_loss(self):
        """model/modeling.py:528-551."""
        st = self.params
        cfg = self.config
        H = self.hidden_size
        hflat = self.encoder_hidden_states['lang'].reshape(self.B * self.L, H)
        midx = self.lang_mask_info['masked_idx'].long()
        flat_idx = (midx + torch.arange(self.B, device=self.device)[:, None] * self.L).reshape(-1)
        pooled = hflat[flat_idx]                                                  # gather, never a one-hot matmul
        targets = self.input_ids.reshape(-1)[flat_idx].int().contiguous()
        h = pooled
        if cfg.get('do_projection', False):
            h = L.linear(h, st.lin('lm_head/projection'), 'gelu')
            h = L.layer_norm(h, st.ln('lm_head/LayerNorm'))
        is_valid = (targets != 0).float()
        denom = is_valid.sum() + 1e-5
        rowweight = (is_valid / denom).contiguous()
        has_bias = cfg.get('do_bias', False)
        loss, raw, am = L.VocabCEFn.apply(h, self._emb, st.p('lm_head/output_bias') if has_bias else None,
                                          st.g('lm_head/output_bias') if has_bias else None, targets, rowweight,
                                          self.vocab_size)
        acc = (is_valid * (am == targets).float()).sum() / denom
        return loss, {'loss': loss, 'acc': acc}

    def _project_and_norm(self, x, name, add_intermediate):
        """model/modeling.py:18-44 under scope 'contrastive'."""
        st = self.params
        if add_intermediate:
            x = L.linear(x, st.lin(f'contrastive/{name}_intermediate'), 'gelu')
            x = L.layer_norm(x, st.ln(f'contrastive/LayerNorm_{name}_ln'))
        x = L.linear(x, st.lin(f'contrastive/{name}'))
        return L.L2NormFn.apply(x)

    def contrastive_loss(self):
        """model/modeling.py:491-526; the cross-replica stack (utils/model_utils.py:673-707) is `dist.all_gather_cat`."""
        cfg = self.config
        add = cfg.get('do_projection', False)
        lang_x = self._project_and_norm(self.lang_trg_h, 'lang_proj', add)
        viz_x = self._project_and_norm(self.img_trg_h, 'viz_proj', add)
        both = self.dist.all_gather_cat(torch.stack([lang_x, viz_x], 1))          # [R*N, 2, C] one fused collective
        all_lang, all_viz = both[:, 0], both[:, 1]
        n_local = lang_x.shape[0]
        temp = cfg.get('contrast_temp', 0.05)
        labels = (torch.arange(n_local, device=self.device) + self.dist.rank * n_local).int()      # :519
        losses = {}
        for name, x, y in [('lang_to_viz', lang_x, all_viz), ('viz_to_lang', viz_x, all_lang)]:
            logits = _ContrastiveLogitsFn.apply(x, y.contiguous(), 1.0 / temp)
            raw, _ = L.SoftmaxCEFn.apply(logits, labels, logits.shape[1])
            losses[name] = raw.mean()
        losses['loss_all'] = cfg.get('contrast_coef', 1.0) * (losses['lang_to_viz'] + losses['viz_to_lang']) / 2
        return losses['loss_all'], losses

    def allpairs_temporal_logits(self, xa, xb, scope_name='temporal_paired'):
        """model/modeling.py:553-596: [B, n, H] x [B, n, H] -> logits [B*n*n, 4]."""
        st = self.params
        Bq, n, H = xa.shape
        assert [Bq, n, H] == [Bq, self.num_chunks_in_group, self.hidden_size] and list(xb.shape) == [Bq, n, H]
        xa_t = xa[:, :, None].expand(Bq, n, n, H)
        xb_t = xb[:, None].expand(Bq, n, n, H)
        hj = torch.cat([xa_t, xb_t], 3).reshape(Bq * n * n, 2 * H)
        h0 = L.linear(hj, st.lin(f'{scope_name}/intermediate'), 'gelu')
        h0 = L.layer_norm(h0, st.ln(f'{scope_name}/LayerNorm_ln0'))
        return L.linear(h0, st.lin(f'{scope_name}/logits'))

    def pooled_segments(self):
        """model/modeling.py:631-634."""
        n, H = self.num_chunks_in_group, self.hidden_size
        h_lang = self.encoder_hidden_states['lang'].reshape(self.B, n, self.lang_chunk_length, H)[:, :, 0]
        h_viz = self.encoder_hidden_states['viz'].reshape(self.B, n, self.viz_chunk_length, H)[:, :, 0]
        return h_lang, h_viz

    def allpairs_temporal_labels(self, video_src_ids, shuffled_idx_img=None):
        """model/modeling.py:598-620 (+ the weights of :635,649-652) on the GPU, integer exact."""
        n = self.num_chunks_in_group
        v = torch.as_tensor(video_src_ids).to(self.device).reshape(self.B, n).int().contiguous()
        if shuffled_idx_img is None:
            s = torch.zeros((self.B, n), device=self.device, dtype=torch.int32)
        else:
            s = torch.as_tensor(shuffled_idx_img).to(self.device).reshape(self.B, n).int().contiguous()
        return ops.temporal_labels(v, s, self.B, n)

    def temporal_loss(self, shuffled_idx_img, video_src_ids):
        """model/modeling.py:622-668."""
        h_lang, h_viz = self.pooled_segments()
        labels, label_w = self.allpairs_temporal_labels(video_src_ids, shuffled_idx_img)
        info = {}
        for name, xa, xb in [('lang_viz', h_lang, h_viz), ('viz_viz', h_viz, h_viz)]:
            logits = self.allpairs_temporal_logits(xa, xb, scope_name=f'{name}_temporal')
            raw, am = L.SoftmaxCEFn.apply(logits, labels, 4)
            raw = raw * label_w
            info[f'{name}_loss'] = raw.mean()
            right = (am == labels).float()
            info[f'{name}_acc'] = (right * label_w).sum() / (label_w.sum() + 1e-5)
        info['loss'] = info['lang_viz_loss']
        if self.config.get('image_shuffle_prob', 0) > 0:
            info['loss'] = info['loss'] + info['viz_viz_loss']
        return info['loss'] * self.config.get('temporal_coef', 1.0), info


class _ContrastiveLogitsFn(torch.autograd.Function):
    """All-pairs logits x @ y^T * inv_temp (model/modeling.py:521) on the MFMA GEMM with the scale fused.  The
    embeddings and the logit gradients are fp32 in the reference; they enter the bf16 GEMMs as hi + lo pairs (three
    of the four cross terms, the lo*lo one is below fp32 resolution), fp32 accumulation across launches."""

    @staticmethod
    def forward(ctx, x, y, inv_temp):
        from .layers import split_bf16
        xh, xl = split_bf16(x.contiguous())
        yh, yl = split_bf16(y.contiguous())
        ctx.x, ctx.y, ctx.inv_temp = (xh, xl), (yh, yl), inv_temp
        npad = (y.shape[0] + 3) // 4 * 4
        out = torch.empty((x.shape[0], npad), device=x.device, dtype=F32)
        ops.gemm_nt(xh, yh, out=out, alpha=inv_temp, n=y.shape[0])
        ops.gemm_nt(xl, yh, out=out, alpha=inv_temp, n=y.shape[0], accumulate=True)
        ops.gemm_nt(xh, yl, out=out, alpha=inv_temp, n=y.shape[0], accumulate=True)
        return out[:, :y.shape[0]] if npad != y.shape[0] else out

    @staticmethod
    def backward(ctx, dlog):
        from .layers import split_bf16
        (xh, xl), (yh, yl) = ctx.x, ctx.y
        n_x, n_y = xh.shape[0], yh.shape[0]
        kp = (n_y + 63) // 64 * 64
        dh, dl = split_bf16(dlog.contiguous())

        def pad_cols(t):
            o = torch.zeros((t.shape[0], kp), device=t.device, dtype=BF16)
            o[:, :n_y] = t
            return o
        dh, dl = pad_cols(dh), pad_cols(dl)
        # dx[n_x, C] = dlog @ y : reduction over n_y -> NT with Bt = y^T (padded)
        yth, ytl = pad_cols(yh.t()), pad_cols(yl.t())
        dx = ops.gemm_nt(dh, yth, out_dtype=F32, alpha=ctx.inv_temp)
        ops.gemm_nt(dl, yth, out=dx, alpha=ctx.inv_temp, accumulate=True)
        ops.gemm_nt(dh, ytl, out=dx, alpha=ctx.inv_temp, accumulate=True)
        # dy[n_y, C] = dlog^T @ x : reduction over n_x
        m = n_y + (n_y % 2)
        dy = torch.zeros((m, xh.shape[1]), device=dlog.device, dtype=F32)
        ops.gemm_tn(dh, xh, dy, accumulate=False, alpha=ctx.inv_temp, m=m)
        ops.gemm_tn(dl, xh, dy, accumulate=True, alpha=ctx.inv_temp, m=m)
        ops.gemm_tn(dh, xl, dy, accumulate=True, alpha=ctx.inv_temp, m=m)
        return dx, dy[:n_y], None


def model_fn_builder(config):
    """Mirror of model/modeling.py:671-810: returns model_fn(features, labels, mode, params) -> dict with the
    total loss and the metric dict (the TPUEstimatorSpec / optimizer wiring is the trainer's job here)."""

    def model_fn(features, labels=None, mode='train', params=None):
        store = params['store']
        images = features['images']
        if mode == 'train' and config.model.get('transpose_input', False):      # :683-685 (HWCN infeed layout)
            images = images.permute(3, 0, 1, 2).contiguous()
        elif mode != 'train':                                                   # :686-687
            images = images.reshape([-1] + list(config.model['image_size']) + [3])
        mcfg = config.model
        if mode != 'train' and mcfg.get('attention_log_in_backward', False):
            mcfg = dict(mcfg, attention_log_in_backward=False)                  # no backward follows an eval / predict graph
        model = MerlotModel(config=mcfg, is_training=True,                    # quirk kept: always True (:691-693)
                            image=images, input_ids=features['input_ids'],
                            use_tpu=config.device.get('use_tpu', False),
                            shuffled_idx_img=features.get('shuffled_idx_img', None), mask_input=True,
                            params=store, noise=features.get('noise'), dist=params.get('dist'),
                            seed=params.get('seed', 0))
        lang_loss, lang_losses = model.mask_loss()
        contr_loss, contr_losses = model.contrastive_loss()
        if config.model.get('temporal_coef', 1.0) > 0.0:
            temp_loss, temp_losses = model.temporal_loss(features['shuffled_idx_img'], video_src_ids=features['video_src_ids'])
        else:
            temp_loss, temp_losses = 0.0, {}
        losses = {f'lang/{k}': v for k, v in lang_losses.items()}
        losses.update({f'attn/{k}': v for k, v in model.attention_log.items()})
        losses.update({f'contr/{k}': v for k, v in contr_losses.items()})
        losses.update({f'temporal/{k}': v for k, v in temp_losses.items()})
        loss = lang_loss + contr_loss + temp_loss                               # :713
        return {'loss': loss, 'metrics': losses, 'model': model, 'token_id_out_of_range': model.token_id_flag()}

    return model_fn
