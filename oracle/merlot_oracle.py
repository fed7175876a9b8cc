"""CPU restatement of the MERLOT pretraining hot path (torch-CPU fp32, unfused).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py for the parity status: pinned
against the reference's own program executed under oracle/tf_shim.py, unpinned
only for the semantics of TensorFlow's primitive kernels).  This file follows
the reference op for op and cites the reference file:line each function restates.  It is the
`use_tpu: False, use_bfloat16: False` variant of the graph: fp32 everywhere,
three separate Q/K/V GEMMs, materialised SxS attention, separate LN / GELU /
residual ops.  Dropout is 0 (TF's RNG stream is not reproducible) and every
random draw is an explicit `noise` input.

Weights are a dict keyed by the reference's TF variable names, in TF layouts
(dense kernels [in, out]; conv kernel HWIO).  Gradients come from torch autograd
over this forward.
"""
import math
import re

import numpy as np
import torch

from . import index_oracle as ix


# ----------------------------------------------------------------------------------------------
# utils/model_utils.py
# ----------------------------------------------------------------------------------------------
def gelu(x):
    """utils/model_utils.py:96-110 (exact erf form)."""
    return x * (0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))))


def layer_norm(x, w, scope, eps=1e-5):
    """utils/model_utils.py:113-130.  `scope` is the full LayerNorm_* variable scope."""
    gamma, beta = w[scope + '/gamma'], w[scope + '/beta']
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)          # tf.nn.moments: population variance
    scale = torch.rsqrt(var + eps) * gamma
    return x * scale - mean * scale + beta


def dense(x, w, scope, act=None):
    """tf.layers.dense: kernel [in, out] applied to the last axis."""
    y = x @ w[scope + '/kernel'] + w[scope + '/bias']
    return act(y) if act is not None else y


def position_embedder2d(w, scope, num_h, num_w, num_cls_emb):
    """utils/model_utils.py:710-739 with num_img = max_nimg = 1."""
    pe = w[scope + '/pos_embs'][:1, :num_h, :num_w]
    H = pe.shape[-1]
    full = pe.reshape(1, num_h * num_w, H)
    if num_cls_emb > 0:
        full = torch.cat([w[scope + '/cls_emb'][:1], full], 1)
    return full.reshape(num_cls_emb + num_h * num_w, H)


def raw_cross_entropy_with_logits(logits, labels):
    """utils/model_utils.py:313-332."""
    logp = torch.log_softmax(logits, -1)
    return -logp.gather(-1, labels.long()[..., None])[..., 0]


def l2_normalize(x):
    """tf.math.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))."""
    return x * torch.rsqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))


# ----------------------------------------------------------------------------------------------
# utils/transformer.py
# ----------------------------------------------------------------------------------------------
def attention_layer(x_flat, mask, B, S, w, scope, num_heads):
    """utils/transformer.py:33-138 (no cache, dropout 0).  mask [B, S, S] of {0,1}."""
    H = x_flat.shape[-1]
    d = H // num_heads

    def proj(name):                                          # :8-30
        p = dense(x_flat, w, f'{scope}/{name}')
        return p.reshape(B, S, num_heads, d).permute(0, 2, 1, 3)

    q, k, v = proj('query_layer'), proj('key_layer'), proj('value_layer')
    scores = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(float(d)))            # :98-100
    m = mask[:, None]
    scores = scores * m - 1e10 * (1 - m)                                          # :109-110
    probs = torch.softmax(scores, -1)                                             # :112
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(B * S, H)                       # :120-127
    out = dense(ctx, w, f'{scope}/context_projection_layer')                     # :130-135
    return out, probs


def mlp_block(x, w, scope):
    """utils/transformer.py:141-163."""
    h = dense(x, w, f'{scope}/intermediate', act=gelu)
    return dense(h, w, f'{scope}/output')


def transformer(hidden_state, mask, w, scope, num_layers, num_heads, return_attn_probs=False):
    """utils/transformer.py:171-247 (pre-LN encoder; compress_attn=True head-mean when probs returned)."""
    B, S, H = hidden_state.shape
    h = hidden_state.reshape(B * S, H)
    probs_all = []
    for l in range(num_layers):
        ls = f'{scope}/layer{l:02d}'
        a, probs = attention_layer(layer_norm(h, w, f'{ls}/LayerNorm_attn_ln0'), mask, B, S, w, ls, num_heads)
        if return_attn_probs:
            probs_all.append(probs.mean(1))                                       # :208-209
        h = h + a                                                                 # :214
        h = h + mlp_block(layer_norm(h, w, f'{ls}/LayerNorm_mlp_ln0'), w, ls)    # :216-220
    h = layer_norm(h, w, f'{scope}/LayerNorm_ln_final')                          # :224
    out = {'_hidden_state_flat': h, 'hidden_state': h.reshape(B, S, H)}
    if return_attn_probs:
        out['self_attn_probs'] = torch.stack(probs_all, 1)                        # :237-238  [B, layers, S, S]
    return out


# ----------------------------------------------------------------------------------------------
# utils/vision_transformer.py
# ----------------------------------------------------------------------------------------------
# ResNet-hybrid stem (utils/vision_transformer.py:8-170, utils/model_utils.py:133-222): what merlot.yaml:30
# (`resnet_layers: [3, 4, 9]`) and the released checkpoints use.  NHWC tensors as in the reference.
# ----------------------------------------------------------------------------------------------
# bf16 policy for the stem: the reference's `use_bfloat16: True` graph holds every activation between the stem's ops
# (and the standardised kernels, utils/vision_transformer.py:58-59) in bf16.  23 such layers in sequence move the
# result by ~4 % relative to the fp32 graph (measured), so implementations that follow that policy are compared with
# THIS variant: `with bf16_stem(): ...` rounds at the same places, everything else stays fp32.
_BF16_STEM = [False]


class bf16_stem(object):
    def __enter__(self):
        _BF16_STEM.append(True)

    def __exit__(self, *a):
        _BF16_STEM.pop()


class _RoundBoth(torch.autograd.Function):
    """bf16 tensor in a bf16 graph: the value is rounded going forward and so is its gradient coming back."""
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def _q(t):
    return _RoundBoth.apply(t) if _BF16_STEM[-1] else t


def group_norm(x, w, scope, num_groups=32, eps=1e-4):
    """utils/model_utils.py:133-222 with mean_close_to_zero=True: one-pass moments E[x^2] - E[x]^2 per (sample,
    group) over (H, W, C/groups), then per-channel gamma / beta."""
    N, Hh, Ww, C = x.shape
    g = x.reshape(N, Hh, Ww, num_groups, C // num_groups)
    cnt = Hh * Ww * (C // num_groups)
    mean = g.sum((1, 2, 4), keepdim=True) / cnt                                   # :198-201
    var = (g * g).sum((1, 2, 4), keepdim=True) / cnt - mean * mean
    y = ((g - mean) * torch.rsqrt(var + eps)).reshape(N, Hh, Ww, C)               # :205
    return _q(y * w[scope + '/gamma'] + w[scope + '/beta'])                       # :219 (:220-221 cast back)


def standardize_kernel(k):
    """utils/vision_transformer.py:52-56: per output channel over (kh, kw, cin), population variance, eps 1e-5."""
    mean = k.mean((0, 1, 2), keepdim=True)
    var = ((k - mean) ** 2).mean((0, 1, 2), keepdim=True)
    return (k - mean) * torch.rsqrt(var + 1e-5)


def conv2d_fixed_padding(x, kernel, strides=1):
    """utils/vision_transformer.py:31-63: weight-standardised conv, no bias; stride 1 -> SAME, stride > 1 -> explicit
    (k-1)//2 low / rest high padding then VALID (fixed_padding, :8-19)."""
    k = _q(standardize_kernel(kernel))                                            # :58-59
    ks = k.shape[0]
    lo = (ks - 1) // 2
    hi_ = ks - 1 - lo
    xc = torch.nn.functional.pad(_q(x).permute(0, 3, 1, 2), [lo, hi_, lo, hi_])  # SAME for odd k at stride 1 == this
    out = torch.nn.functional.conv2d(xc, k.permute(3, 2, 0, 1), None, stride=strides)
    return _q(out.permute(0, 2, 3, 1))


def _avg_pool(x, k):
    return torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), k, k).permute(0, 2, 3, 1)


def lite_resnet50(x, w, scope, layers, width=64):
    """utils/vision_transformer.py:114-170 (+ bottleneck_block :66-93, block_group :96-111).  Variable names are the
    ones tf.layers.Conv2D / variable_scope(default_name=...) generate: conv2d, conv2d_1, ... and GroupNorm,
    GroupNorm_1, ... counted per enclosing scope in creation order."""
    rs = f'{scope}/resnet50lite'
    relu = torch.relu
    x = relu(group_norm(conv2d_fixed_padding(x, w[f'{rs}/stem/conv2d/kernel'], 2), w, f'{rs}/stem/GroupNorm_stem0'))
    x = relu(group_norm(conv2d_fixed_padding(x, w[f'{rs}/stem/conv2d_1/kernel']), w, f'{rs}/stem/GroupNorm_stem1'))
    x = relu(group_norm(conv2d_fixed_padding(x, w[f'{rs}/stem/conv2d_2/kernel']), w, f'{rs}/stem/GroupNorm_stem2'))
    c = _avg_pool(x, 2)                                                           # :158
    for i, blocks in enumerate(layers):
        gs = f'{rs}/block_group{i + 1}'
        strides = 1 if i == 0 else 2
        idx = [0, 0]                                                              # next conv2d / GroupNorm suffix

        def nm(kind):
            j = 0 if kind == 'conv2d' else 1
            name = kind if idx[j] == 0 else f'{kind}_{idx[j]}'
            idx[j] += 1
            return f'{gs}/{name}'

        for bi in range(blocks):
            st = strides if bi == 0 else 1
            shortcut = c
            if bi == 0:                                                           # projection shortcut (:75-83)
                sc_in = _avg_pool(c, st) if st > 1 else c
                shortcut = group_norm(conv2d_fixed_padding(sc_in, w[nm('conv2d') + '/kernel']), w, nm('GroupNorm'))
            h = relu(group_norm(conv2d_fixed_padding(c, w[nm('conv2d') + '/kernel']), w, nm('GroupNorm')))
            h = relu(group_norm(conv2d_fixed_padding(h, w[nm('conv2d') + '/kernel']), w, nm('GroupNorm')))
            if st > 1:
                h = _avg_pool(h, st)                                              # :89-90
            h = group_norm(conv2d_fixed_padding(h, w[nm('conv2d') + '/kernel']), w, nm('GroupNorm'))
            c = relu(h + shortcut)                                                # :93
    return c


def resnet_variable_shapes(cfg, scope='vision_backbone/vision_transformer', width=64):
    layers = cfg.get('resnet_layers', [])
    rs = f'{scope}/resnet50lite'
    shapes = {}

    def gn(name, c):
        shapes[name + '/gamma'] = (c,)
        shapes[name + '/beta'] = (c,)
    for j, (ci, co) in enumerate([(3, width // 2), (width // 2, width // 2), (width // 2, width)]):
        shapes[f'{rs}/stem/conv2d' + (f'_{j}' if j else '') + '/kernel'] = (3, 3, ci, co)
        gn(f'{rs}/stem/GroupNorm_stem{j}', co)
    cin = width
    for i, blocks in enumerate(layers):
        f = width * (2 ** i)
        gs = f'{rs}/block_group{i + 1}'
        k = 0

        def add(kh, ci, co):
            nonlocal k
            sfx = f'_{k}' if k else ''
            shapes[f'{gs}/conv2d{sfx}/kernel'] = (kh, kh, ci, co)
            gn(f'{gs}/GroupNorm{sfx}', co)
            k += 1
        for bi in range(blocks):
            if bi == 0:
                add(1, cin, 4 * f)
            add(1, cin, f)
            add(3, f, f)
            add(1, f, 4 * f)
            cin = 4 * f
    shapes[f'{scope}/conv_postresnet_proj/kernel'] = (1, 1, cin, cfg['hidden_size'])
    shapes[f'{scope}/conv_postresnet_proj/bias'] = (cfg['hidden_size'],)
    return shapes


# ----------------------------------------------------------------------------------------------
def vision_transformer_backbone(image, w, cfg, scope='vision_backbone/vision_transformer'):
    """utils/vision_transformer.py:173-274: patch-conv stem (resnet_layers == []) or the ResNet hybrid (:206-223)."""
    P = cfg['patch_size']
    H = cfg['hidden_size']
    num_cls = cfg.get('num_cls_emb', 2)
    N, h0, w0, _ = image.shape
    assert h0 % P == 0 and w0 % P == 0
    h1, w1 = h0 // P, w0 // P
    x = image - 0.5                                                               # :193
    resnet_layers = cfg.get('resnet_layers', [])
    if len(resnet_layers) == 0:
        kern = w[f'{scope}/conv2d/kernel']                                        # HWIO [P, P, 3, H]
        # 16x16 / stride 16 VALID conv == im2col (ph, pw, c) x [P*P*3, H]    :196-205
        patches = x.reshape(N, h1, P, w1, P, 3).permute(0, 1, 3, 2, 4, 5).reshape(N, h1 * w1, P * P * 3)
        x = patches @ kern.reshape(P * P * 3, H) + w[f'{scope}/conv2d/bias']
    else:
        assert P == 16                                                            # :208
        c = lite_resnet50(x, w, scope, resnet_layers)                             # [N, h1, w1, 1024 (3 groups)]
        pk = w[f'{scope}/conv_postresnet_proj/kernel']                            # 1x1 conv to hidden (:213-223)
        x = c.reshape(N, h1 * w1, c.shape[-1]) @ pk.reshape(pk.shape[2], H) + w[f'{scope}/conv_postresnet_proj/bias']
    x = torch.cat([torch.zeros(N, num_cls, H, dtype=x.dtype), x], 1)             # :231
    pos = position_embedder2d(w, f'{scope}/pos_embs', h1, w1, num_cls)           # :232-233
    x = layer_norm(x + pos, w, f'{scope}/LayerNorm_ctx_patches_pre_ln')          # :234
    S = h1 * w1 + num_cls
    mask = torch.ones(N, S, S, dtype=x.dtype)                                     # :239
    nl = cfg.get('num_vision_transformer_hidden_layers', cfg['num_hidden_layers'])
    info = transformer(x, mask, w, scope, nl, cfg['num_attention_heads'])
    info['cls'] = info['hidden_state'][:, :num_cls]                               # :251
    info['seq'] = info['hidden_state'][:, num_cls:]                               # :252
    sp = cfg['spatial_pool_size']
    if sp > 1:                                                                    # :255-267
        seq = info['seq'].reshape(N, h1, w1, H).permute(0, 3, 1, 2)
        seq = torch.nn.functional.avg_pool2d(seq, sp, sp)
        h2, w2 = h1 // sp, w1 // sp
        info['seq'] = seq.permute(0, 2, 3, 1).reshape(N, h2 * w2, H)
    else:
        h2, w2 = h1, w1
    info['num_h'], info['num_w'] = h2, w2
    return info


# ----------------------------------------------------------------------------------------------
# model/modeling.py
# ----------------------------------------------------------------------------------------------
def project_and_norm(x, w, name, add_intermediate):
    """model/modeling.py:18-44 under scope 'contrastive'."""
    if add_intermediate:
        x = dense(x, w, f'contrastive/{name}_intermediate', act=gelu)
        x = layer_norm(x, w, f'contrastive/LayerNorm_{name}_ln')
    return l2_normalize(dense(x, w, f'contrastive/{name}'))


class MerlotOracle(object):
    """Restatement of MerlotModel (model/modeling.py:47-668); same constructor meaning.

    Extra explicit inputs: `weights` (TF-named dict) and `noise` (see index_oracle.mask_inputs).
    """

    def __init__(self, config, weights, image, input_ids, mask_input=False, shuffled_idx_img=None,
                 log_attention_probs=True, noise=None, attention_summs=None, img_mask=None):
        """attention_summs (optional, [B, L] fp32): use THESE per-key attention sums for the top-k step of mask_inputs
        instead of the oracle's own -- lets a test hold the integer masking logic to bit-exactness when the other
        implementation's sums differ in the last bf16 digits (a near-tie then legitimately flips the top-k set)."""
        self.config = dict(config)
        self._summs_override = attention_summs
        self.w = weights
        cfg = self.config
        input_ids = torch.as_tensor(input_ids).long()
        if input_ids.dim() == 2:                                                  # :72-77
            self.num_chunks = 1
            self.num_chunks_in_group = 1
            self.batch_size, self.lang_chunk_length = input_ids.shape
            self.input_ids = input_ids[:, None]
        else:                                                                     # :78-82
            self.input_ids = input_ids
            self.batch_size, self.num_chunks, self.lang_chunk_length = input_ids.shape
            self.num_chunks_in_group = cfg.get('num_chunks_in_group', self.num_chunks)
            assert self.num_chunks % self.num_chunks_in_group == 0
        self.hidden_size = cfg['hidden_size']
        self.vocab_size = cfg['vocab_size']
        H = self.hidden_size
        n = self.num_chunks_in_group

        # ---- vision half (:94-133)
        self.vision_transformer_info = vision_transformer_backbone(image, self.w, cfg)
        vti = self.vision_transformer_info
        self.img_trg_h = vti['cls'][:, 1]                                         # :99
        image_feats = torch.cat([vti['cls'][:, 0, None], vti['seq']], 1)          # :101-104
        image_feats = image_feats.reshape(self.B, self.P, H)                      # :121
        image_feats = image_feats + self.vision_pos_emb(shuffled_idx_img)         # :125
        image_feats = layer_norm(image_feats, self.w, 'vision_backbone/LayerNorm_final_ln')   # :126
        self.image_feats = image_feats
        if img_mask is None:                                                      # :105-108, :122 (num_imgs = 1)
            img_valid = torch.ones(self.B, self.P, dtype=torch.bool)
        else:
            img_valid = torch.as_tensor(img_mask).bool().reshape(self.B, 1).expand(self.B, self.P)

        # ---- language half (:135-149)
        if mask_input:
            self.lang_trg_h, self.lang_transformer_info = self.langonly_reps()
            self.lang_mask_info = self.mask_inputs(noise)
            ids_to_use = self.lang_mask_info['masked_ids']
        else:
            ids_to_use = self.input_ids
        ids_to_use = ids_to_use.reshape(self.B, self.L)
        lang_x = self.embed_words(ids_to_use, 'position_embeddings')
        lang_valid = ids_to_use != 0

        # ---- joint encoder (:151-184)
        enc_in = torch.cat([image_feats, lang_x], 1)
        is_valid = torch.cat([img_valid, lang_valid], 1)
        mask = (is_valid[:, None] & is_valid[:, :, None])                         # :158
        if cfg.get('disable_pairwise_lang_attn', False):                          # :160-168
            seg = torch.cat([torch.zeros(self.P, dtype=torch.long),
                             1 + torch.arange(self.L) // self.lang_chunk_length])
            can = (seg[:, None] == seg[None]) | (seg == 0)[None] | (seg == 0)[:, None]
            mask = mask & can
        mask = mask.to(enc_in.dtype)
        self.encoder_info = transformer(enc_in, mask, self.w, 'encoder', cfg['num_hidden_layers'],
                                        cfg['num_attention_heads'], return_attn_probs=log_attention_probs)
        hs = self.encoder_info['hidden_state']
        self.encoder_hidden_states = {'viz': hs[:, :self.P], 'lang': hs[:, self.P:]}

        if log_attention_probs:                                                   # :186-203
            p = self.encoder_info['self_attn_probs'].mean(1)
            vf = is_valid.to(p.dtype)
            p = p * (vf[:, None] * vf[:, :, None])
            p = p.mean(0)
            p = p / p.sum()
            pieces = [('viz', 0, self.P), ('lang', self.P, self.P + self.L)]
            attns = {}
            for tn, ts, te in pieces:
                for fn, fs, fe in pieces:
                    attns[f'{fn}2{tn}'] = p[ts:te, fs:fe].sum()
            self.attention_log = {f'encoder/{k}': v for k, v in sorted(attns.items())}

    # ---- shape algebra (:234-248)
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

    def embed_words(self, ids2d, norm_scope):
        """model/modeling.py:262-297 (dropout 0)."""
        table = self.w['word_embeddings/word_embeddings']
        L = ids2d.shape[1]
        emb = table[ids2d] + self.w[f'{norm_scope}/position_embeddings'][:L][None]
        return layer_norm(emb, self.w, f'{norm_scope}/LayerNorm_embed_norm')

    def vision_pos_emb(self, shuffled_idx_img):
        """model/modeling.py:299-337 (num_imgs = num_texts = 1)."""
        n, vl, H = self.num_chunks_in_group, self.viz_chunk_length, self.hidden_size
        table = self.w['vision_backbone/img_idx_pe']
        if shuffled_idx_img is None:                                              # :312-315
            my_pe = table[:n][None, :, None].expand(1, n, vl, H).reshape(1, self.P, H)
        else:                                                                     # :316-323
            idx = torch.as_tensor(shuffled_idx_img).long().reshape(-1)
            my_pe = table[idx][:, None].expand(-1, vl, H).reshape(self.B, self.P, H)
        pe2d = position_embedder2d(self.w, 'vision_backbone/final_pe', self.vision_transformer_info['num_h'],
                                   self.vision_transformer_info['num_w'], 1)      # :327-335
        return my_pe + pe2d.repeat(n, 1)[None]                                    # :336

    def langonly_reps(self):
        """model/modeling.py:339-379."""
        cfg = self.config
        if 'langonly_num_chunks_in_group' in cfg:                                 # :345-351
            g = cfg['langonly_num_chunks_in_group']
            ids2d = self.input_ids.reshape(self.batch_size * (self.num_chunks // g), self.lang_chunk_length * g)
        else:
            ids2d = self.input_ids.reshape(self.batch_size, self.lang_chunk_length * self.num_chunks)
        emb = self.embed_words(ids2d, 'langonly_embeddings')
        valid = ids2d != 0
        mask = (valid[:, None] & valid[:, :, None]).to(emb.dtype)
        scope = 'encoder' if cfg.get('share_params', True) else 'langonly_encoder'
        info = transformer(emb, mask, self.w, scope, cfg['num_lang_transformer_hidden_layers'],
                           cfg['num_attention_heads'], return_attn_probs=True)
        pool = info['_hidden_state_flat'].reshape(self.batch_size * self.num_chunks, self.lang_chunk_length,
                                                  self.hidden_size)[:, 0]         # :372-375
        return pool, info

    def attention_summs(self):
        """model/modeling.py:428-431."""
        s = self.lang_transformer_info['self_attn_probs'].sum((1, 2))
        return s.reshape(self.B, self.L).float()

    def mask_inputs(self, noise):
        """model/modeling.py:381-489 via index_oracle.mask_inputs (integer, explicit noise)."""
        ids2d = self.input_ids.reshape(self.B, self.L).numpy().astype(np.int32)
        summ = self.attention_summs().detach().numpy() if self.config.get('masking_use_attn', True) else None
        if summ is not None and self._summs_override is not None:
            summ = np.asarray(self._summs_override, dtype=np.float32).reshape(summ.shape)
        masked_ids, masked_idx = ix.mask_inputs(ids2d, summ, self.config, self.vocab_size, noise)
        return {'masked_ids': torch.from_numpy(masked_ids).long().reshape(self.input_ids.shape),
                'masked_idx': torch.from_numpy(masked_idx).long()}

    def lm_head(self, h):
        """model/modeling.py:205-224."""
        if self.config.get('do_projection', False):
            h = layer_norm(dense(h, self.w, 'lm_head/projection', act=gelu), self.w, 'lm_head/LayerNorm')
        logits = h @ self.w['word_embeddings/word_embeddings'].t()
        if self.config.get('do_bias', False):
            logits = logits + self.w['lm_head/output_bias']
        return logits

    def mask_loss(self):
        """model/modeling.py:528-551."""
        hflat = self.encoder_hidden_states['lang'].reshape(self.B * self.L, self.hidden_size)
        idx = (self.lang_mask_info['masked_idx'] + torch.arange(self.B)[:, None] * self.L).reshape(-1)
        pooled = hflat[idx]
        targets = self.input_ids.reshape(-1)[idx]
        logits = self.lm_head(pooled)
        raw = raw_cross_entropy_with_logits(logits, targets)
        valid = (targets != 0).to(raw.dtype)
        denom = valid.sum() + 1e-5
        loss = (valid * raw).sum() / denom
        acc = (valid * (logits.argmax(-1) == targets).to(raw.dtype)).sum() / denom
        return loss, {'loss': loss, 'acc': acc}

    def contrastive_embeddings(self):
        add = self.config.get('do_projection', False)
        return (project_and_norm(self.lang_trg_h, self.w, 'lang_proj', add),
                project_and_norm(self.img_trg_h, self.w, 'viz_proj', add))

    def contrastive_loss(self, all_lang=None, all_viz=None, my_group_idx=0):
        """model/modeling.py:491-526.  Single replica unless the gathered [R*N, H] sets are given."""
        lang_x, viz_x = self.contrastive_embeddings()
        all_lang = lang_x if all_lang is None else all_lang
        all_viz = viz_x if all_viz is None else all_viz
        n_local = lang_x.shape[0]
        temp = self.config.get('contrast_temp', 0.05)
        labels = torch.arange(n_local) + my_group_idx * n_local                   # :519
        losses = {}
        for name, x, y in [('lang_to_viz', lang_x, all_viz), ('viz_to_lang', viz_x, all_lang)]:
            logits = (x @ y.t()) / temp
            losses[name] = raw_cross_entropy_with_logits(logits, labels).mean()
        losses['loss_all'] = self.config.get('contrast_coef', 1.0) * (losses['lang_to_viz'] + losses['viz_to_lang']) / 2
        return losses['loss_all'], losses

    def allpairs_temporal_logits(self, xa, xb, scope_name):
        """model/modeling.py:553-596."""
        Bq, n, H = xa.shape
        xa_t = xa[:, :, None].expand(Bq, n, n, H).reshape(Bq, n * n, H)
        xb_t = xb[:, None].expand(Bq, n, n, H).reshape(Bq, n * n, H)
        hj = torch.cat([xa_t, xb_t], 2).reshape(Bq * n * n, 2 * H)
        h0 = dense(hj, self.w, f'{scope_name}/intermediate', act=gelu)
        h0 = layer_norm(h0, self.w, f'{scope_name}/LayerNorm_ln0')
        return dense(h0, self.w, f'{scope_name}/logits')

    def pooled_segments(self):
        """model/modeling.py:631-634."""
        n, H = self.num_chunks_in_group, self.hidden_size
        h_lang = self.encoder_hidden_states['lang'].reshape(self.B, n, self.lang_chunk_length, H)[:, :, 0]
        h_viz = self.encoder_hidden_states['viz'].reshape(self.B, n, self.viz_chunk_length, H)[:, :, 0]
        return h_lang, h_viz

    def temporal_loss(self, shuffled_idx_img, video_src_ids):
        """model/modeling.py:622-668."""
        n = self.num_chunks_in_group
        h_lang, h_viz = self.pooled_segments()
        labels = torch.from_numpy(ix.allpairs_temporal_labels(np.asarray(video_src_ids).reshape(self.B, n), n)).long()
        label_w = torch.from_numpy(ix.temporal_label_weights(np.asarray(shuffled_idx_img), n))
        info = {}
        for name, xa, xb in [('lang_viz', h_lang, h_viz), ('viz_viz', h_viz, h_viz)]:
            logits = self.allpairs_temporal_logits(xa, xb, f'{name}_temporal')
            raw = raw_cross_entropy_with_logits(logits, labels) * label_w
            info[f'{name}_loss'] = raw.mean()
            right = (logits.argmax(-1) == labels).to(raw.dtype)
            info[f'{name}_acc'] = (right * label_w).sum() / (label_w.sum() + 1e-5)
        info['loss'] = info['lang_viz_loss']
        if self.config.get('image_shuffle_prob', 0) > 0:
            info['loss'] = info['loss'] + info['viz_viz_loss']
        return info['loss'] * self.config.get('temporal_coef', 1.0), info

    def total_loss(self, shuffled_idx_img, video_src_ids):
        """model/modeling.py:700-713."""
        lang_loss, ll = self.mask_loss()
        contr_loss, cl = self.contrastive_loss()
        if self.config.get('temporal_coef', 1.0) > 0.0:
            temp_loss, tl = self.temporal_loss(shuffled_idx_img, video_src_ids)
        else:
            temp_loss, tl = 0.0, {}
        return lang_loss + contr_loss + temp_loss, {'lang': ll, 'contr': cl, 'temporal': tl}


def sort_story_probs(config, weights, image, input_ids, u_shuffle, duplication_factor=2,
                     faithful_dup_reshape=True):
    """downstream/sort_story/get_zero_shot_logits.py:45-86.

    image [bs, n, H, W, 3], input_ids [bs, n, 32]; u_shuffle [bs*dup*n] uniforms standing in for the
    fixed-seed stateless_uniform (:55-56).  Returns {'lang_viz_probs','viz_viz_probs'} [bs, n, n, 3].
    """
    bs, n = image.shape[0], image.shape[1]
    images = image.repeat(duplication_factor, 1, 1, 1, 1)                         # :45
    sents = torch.as_tensor(input_ids).repeat(duplication_factor, 1, 1)           # :46
    images_resh = images.reshape(bs * duplication_factor * n, *image.shape[2:])
    sidx = ix.sort_story_shuffled_idx(u_shuffle, n)                               # :55-56
    m = MerlotOracle(config, weights, images_resh, sents[:, :, :32], mask_input=False,
                     shuffled_idx_img=sidx.reshape(-1), log_attention_probs=False)
    h_lang, h_viz = m.pooled_segments()
    out = {}
    for name, xa, xb in [('lang_viz', h_lang, h_viz), ('viz_viz', h_viz, h_viz)]:
        logits = m.allpairs_temporal_logits(xa, xb, f'{name}_temporal')
        probs = torch.softmax(logits, -1)[:, 1:]                                  # :80
        # :83-84 -- reference quirk kept: rows are tiled dup-major (tf.tile on axis 0, :45) but the
        # reshape reads them as [batch, dup]; for batch > 1 this averages different stories.
        if faithful_dup_reshape:
            probs = probs.reshape(bs, duplication_factor, n, n, 3).mean(1)
        else:
            probs = probs.reshape(duplication_factor, bs, n, n, 3).mean(0)
        out[f'{name}_probs'] = probs
    return out


# ----------------------------------------------------------------------------------------------
# weight init (fresh-init distributions of the reference, for synthetic runs)
# ----------------------------------------------------------------------------------------------
def variable_shapes(cfg):
    """TF variable name -> shape for the patch-ViT variant."""
    H, I, V = cfg['hidden_size'], cfg['intermediate_size'], cfg['vocab_size']
    P = cfg['patch_size']
    ncls = cfg.get('num_cls_emb', 2)
    shapes = {}

    def ln(scope):
        shapes[scope + '/gamma'] = (H,)
        shapes[scope + '/beta'] = (H,)

    def dn(scope, i, o):
        shapes[scope + '/kernel'] = (i, o)
        shapes[scope + '/bias'] = (o,)

    def stack(scope, nl):
        for l in range(nl):
            ls = f'{scope}/layer{l:02d}'
            ln(f'{ls}/LayerNorm_attn_ln0')
            for nm in ['query_layer', 'key_layer', 'value_layer', 'context_projection_layer']:
                dn(f'{ls}/{nm}', H, H)
            ln(f'{ls}/LayerNorm_mlp_ln0')
            dn(f'{ls}/intermediate', H, I)
            dn(f'{ls}/output', I, H)
        ln(f'{scope}/LayerNorm_ln_final')

    vs = 'vision_backbone/vision_transformer'
    if len(cfg.get('resnet_layers', [])) == 0:
        shapes[f'{vs}/conv2d/kernel'] = (P, P, 3, H)
        shapes[f'{vs}/conv2d/bias'] = (H,)
    else:
        shapes.update(resnet_variable_shapes(cfg, vs))
    shapes[f'{vs}/pos_embs/pos_embs'] = (1, 64, 64, H)
    shapes[f'{vs}/pos_embs/cls_emb'] = (1, ncls, H)
    ln(f'{vs}/LayerNorm_ctx_patches_pre_ln')
    stack(vs, cfg.get('num_vision_transformer_hidden_layers', cfg['num_hidden_layers']))
    shapes['vision_backbone/img_idx_pe'] = (cfg.get('max_vision_pos_embeddings', 1024), H)
    shapes['vision_backbone/final_pe/pos_embs'] = (1, 64, 64, H)
    shapes['vision_backbone/final_pe/cls_emb'] = (1, 1, H)
    ln('vision_backbone/LayerNorm_final_ln')
    shapes['word_embeddings/word_embeddings'] = (V, H)
    for sc in ['langonly_embeddings', 'position_embeddings']:
        shapes[f'{sc}/position_embeddings'] = (cfg['max_position_embeddings'], H)
        ln(f'{sc}/LayerNorm_embed_norm')
    if cfg.get('share_params', True):                        # model/modeling.py:171-172, 357-362
        stack('encoder', max(cfg['num_hidden_layers'], cfg.get('num_lang_transformer_hidden_layers', 0)))
    else:
        stack('encoder', cfg['num_hidden_layers'])
        stack('langonly_encoder', cfg['num_lang_transformer_hidden_layers'])
    dn('lm_head/projection', H, H)
    ln('lm_head/LayerNorm')
    shapes['lm_head/output_bias'] = (V,)
    C = cfg.get('contrastive_size', H)
    for nm in ['lang_proj', 'viz_proj']:
        dn(f'contrastive/{nm}_intermediate', H, C)
        shapes[f'contrastive/LayerNorm_{nm}_ln/gamma'] = (C,)
        shapes[f'contrastive/LayerNorm_{nm}_ln/beta'] = (C,)
        dn(f'contrastive/{nm}', C, C)
    for nm in ['lang_viz_temporal', 'viz_viz_temporal']:
        dn(f'{nm}/intermediate', 2 * H, H)
        ln(f'{nm}/LayerNorm_ln0')
        dn(f'{nm}/logits', H, 4)
    return shapes


def init_weights(cfg, seed=0, perturb=True):
    """Random weights with the reference's init distributions (utils/transformer.py:166-168,
    utils/vision_transformer.py:204, utils/model_utils.py:118-119).  `perturb` jitters LN
    gamma/beta and biases away from 1/0 so parity tests exercise them."""
    g = torch.Generator().manual_seed(seed)
    std = cfg.get('initializer_range', 0.02)
    w = {}
    for name, shp in variable_shapes(cfg).items():
        if name.endswith('/gamma'):
            t = torch.ones(shp) + (0.1 * torch.randn(shp, generator=g) if perturb else 0)
        elif name.endswith('/beta') or name.endswith('/bias') or name.endswith('output_bias'):
            t = (0.02 * torch.randn(shp, generator=g)) if perturb else torch.zeros(shp)
        elif re.search(r'(conv2d(_\d+)?|conv_postresnet_proj)/kernel$', name):       # variance_scaling (fan_in)
            fan_in = shp[0] * shp[1] * shp[2]
            t = torch.randn(shp, generator=g).clamp(-2, 2) * math.sqrt(1.0 / fan_in)
        else:
            t = torch.randn(shp, generator=g).clamp(-2, 2) * std
        w[name] = t.float()
    return w
