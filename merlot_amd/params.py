"""Parameter arena of the MI355X MERLOT path.

All trainable tensors live in ONE flat fp32 master buffer (and one flat fp32 gradient buffer of the same
layout) so that the DP gradient all-reduce and the fused AdamW step each run over a few large contiguous
ranges (RCCL over xGMI is per-link bound: big buckets).  Working copies for the MFMA GEMMs are bf16:
a straight cast of the whole arena (one launch) plus a transposed copy per Linear (for dgrad).

Names follow the reference's TF variable scopes except that
  * dense / conv kernels are stored [out, in] (TF: [in, out] / HWIO) and
  * query/key/value of a layer are fused into `<layer>/qkv/{kernel,bias}` ([3H, H] / [3H]);
`load_tf_weights` / `export_tf_weights` convert to and from the reference's names and layouts.
"""
import math
from collections import OrderedDict

import torch

from . import ops


class Lin(object):
    """Handle of one Linear: fp32 master + grad views, bf16 straight ([out,in]) and transposed ([in,out]) copies."""
    __slots__ = ('name', 'w', 'b', 'gw', 'gb', 'wb', '_wbT', '_store', 'ld_wbT')

    def __init__(self, name, store=None):
        self.name = name
        self.w = self.b = self.gw = self.gb = self.wb = self._wbT = None
        self._store = store

    @property
    def wbT(self):
        """the transposed bf16 copy; Linears registered since the last refresh get theirs in ONE batched launch at the first read
        (a model's first forward registers ~100 of them: one launch, not one launch over the growing table per registration)."""
        if self._store is not None and self._store._t_dirty:
            self._store._flush_transposed()
        return self._wbT

    @wbT.setter
    def wbT(self, v):
        self._wbT = v


class LN(object):
    __slots__ = ('name', 'gamma', 'beta', 'ggamma', 'gbeta')

    def __init__(self, name):
        self.name = name


def _stack_entries(scope, nl, H, I):
    e = []
    for l in range(nl):
        ls = f'{scope}/layer{l:02d}'
        e += [(f'{ls}/LayerNorm_attn_ln0/gamma', (H,), 'ones'), (f'{ls}/LayerNorm_attn_ln0/beta', (H,), 'zeros'),
              (f'{ls}/qkv/kernel', (3 * H, H), 'normal'), (f'{ls}/qkv/bias', (3 * H,), 'zeros'),
              (f'{ls}/context_projection_layer/kernel', (H, H), 'normal'),
              (f'{ls}/context_projection_layer/bias', (H,), 'zeros'),
              (f'{ls}/LayerNorm_mlp_ln0/gamma', (H,), 'ones'), (f'{ls}/LayerNorm_mlp_ln0/beta', (H,), 'zeros'),
              (f'{ls}/intermediate/kernel', (I, H), 'normal'), (f'{ls}/intermediate/bias', (I,), 'zeros'),
              (f'{ls}/output/kernel', (H, I), 'normal'), (f'{ls}/output/bias', (H,), 'zeros')]
    e += [(f'{scope}/LayerNorm_ln_final/gamma', (H,), 'ones'), (f'{scope}/LayerNorm_ln_final/beta', (H,), 'zeros')]
    return e


def resnet_entries(cfg, vs, width=64):
    """Variables of the ResNet-hybrid stem (utils/vision_transformer.py:114-170, 206-223) in creation order, named as the
    reference's scopes name them (conv2d, conv2d_1, ... / GroupNorm, GroupNorm_1, ... per block_group).  Convolution
    kernels keep TF's HWIO layout (they are weight-standardised into a [Co, K] bf16 operand at use)."""
    layers = cfg.get('resnet_layers', [])
    rs = f'{vs}/resnet50lite'
    e = []
    for j, (ci, co) in enumerate([(3, width // 2), (width // 2, width // 2), (width // 2, width)]):
        e += [(f'{rs}/stem/conv2d' + (f'_{j}' if j else '') + '/kernel', (3, 3, ci, co), 'conv_hwio'),
              (f'{rs}/stem/GroupNorm_stem{j}/gamma', (co,), 'ones'), (f'{rs}/stem/GroupNorm_stem{j}/beta', (co,), 'zeros')]
    cin = width
    for i, blocks in enumerate(layers):
        f = width * (2 ** i)
        gs = f'{rs}/block_group{i + 1}'
        k = 0
        for bi in range(blocks):
            convs = ([(1, cin, 4 * f)] if bi == 0 else []) + [(1, cin, f), (3, f, f), (1, f, 4 * f)]
            for kh, ci, co in convs:
                sfx = f'_{k}' if k else ''
                e += [(f'{gs}/conv2d{sfx}/kernel', (kh, kh, ci, co), 'conv_hwio'),
                      (f'{gs}/GroupNorm{sfx}/gamma', (co,), 'ones'), (f'{gs}/GroupNorm{sfx}/beta', (co,), 'zeros')]
                k += 1
            cin = 4 * f
    e += [(f'{vs}/conv_postresnet_proj/kernel', (cfg['hidden_size'], cin), 'conv'),
          (f'{vs}/conv_postresnet_proj/bias', (cfg['hidden_size'],), 'zeros')]
    return e


def param_entries(cfg):
    """[(internal name, shape, init kind)] in arena order."""
    H, I, V = cfg['hidden_size'], cfg['intermediate_size'], cfg['vocab_size']
    P = cfg['patch_size']
    C = cfg.get('contrastive_size', H)
    ncls = cfg.get('num_cls_emb', 2)
    vs = 'vision_backbone/vision_transformer'
    nl_vit = cfg.get('num_vision_transformer_hidden_layers', cfg['num_hidden_layers'])
    nl_enc = max(cfg['num_hidden_layers'], cfg.get('num_lang_transformer_hidden_layers', 0))
    if cfg.get('resnet_layers'):
        e = resnet_entries(cfg, vs)
    else:
        e = [(f'{vs}/conv2d/kernel', (H, P * P * 3), 'conv'), (f'{vs}/conv2d/bias', (H,), 'zeros')]
    e += [(f'{vs}/pos_embs/pos_embs', (1, 64, 64, H), 'normal'), (f'{vs}/pos_embs/cls_emb', (1, ncls, H), 'normal'),
         (f'{vs}/LayerNorm_ctx_patches_pre_ln/gamma', (H,), 'ones'),
         (f'{vs}/LayerNorm_ctx_patches_pre_ln/beta', (H,), 'zeros')]
    e += _stack_entries(vs, nl_vit, H, I)
    e += [('vision_backbone/img_idx_pe', (cfg.get('max_vision_pos_embeddings', 1024), H), 'normal'),
          ('vision_backbone/final_pe/pos_embs', (1, 64, 64, H), 'normal'),
          ('vision_backbone/final_pe/cls_emb', (1, 1, H), 'normal'),
          ('vision_backbone/LayerNorm_final_ln/gamma', (H,), 'ones'),
          ('vision_backbone/LayerNorm_final_ln/beta', (H,), 'zeros'),
          ('word_embeddings/word_embeddings', (V, H), 'normal')]
    for sc in ['langonly_embeddings', 'position_embeddings']:
        e += [(f'{sc}/position_embeddings', (cfg['max_position_embeddings'], H), 'normal'),
              (f'{sc}/LayerNorm_embed_norm/gamma', (H,), 'ones'), (f'{sc}/LayerNorm_embed_norm/beta', (H,), 'zeros')]
    if cfg.get('share_params', True):                        # model/modeling.py:171-172, 357-362
        e += _stack_entries('encoder', nl_enc, H, I)
    else:
        e += _stack_entries('encoder', cfg['num_hidden_layers'], H, I)
        e += _stack_entries('langonly_encoder', cfg['num_lang_transformer_hidden_layers'], H, I)
    e += [('lm_head/projection/kernel', (H, H), 'normal'), ('lm_head/projection/bias', (H,), 'zeros'),
          ('lm_head/LayerNorm/gamma', (H,), 'ones'), ('lm_head/LayerNorm/beta', (H,), 'zeros'),
          ('lm_head/output_bias', (V,), 'zeros')]
    for nm in ['lang_proj', 'viz_proj']:
        e += [(f'contrastive/{nm}_intermediate/kernel', (C, H), 'normal'), (f'contrastive/{nm}_intermediate/bias', (C,), 'zeros'),
              (f'contrastive/LayerNorm_{nm}_ln/gamma', (C,), 'ones'), (f'contrastive/LayerNorm_{nm}_ln/beta', (C,), 'zeros'),
              (f'contrastive/{nm}/kernel', (C, C), 'normal'), (f'contrastive/{nm}/bias', (C,), 'zeros')]
    for nm in ['lang_viz_temporal', 'viz_viz_temporal']:
        e += [(f'{nm}/intermediate/kernel', (H, 2 * H), 'normal'), (f'{nm}/intermediate/bias', (H,), 'zeros'),
              (f'{nm}/LayerNorm_ln0/gamma', (H,), 'ones'), (f'{nm}/LayerNorm_ln0/beta', (H,), 'zeros'),
              (f'{nm}/logits/kernel', (4, H), 'normal'), (f'{nm}/logits/bias', (4,), 'zeros')]
    return e


def _align(n, a=64):
    return (n + a - 1) // a * a


class ParamStore(object):
    def __init__(self, cfg, device, seed=0, init=True):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.entries = param_entries(cfg)
        self.offsets = OrderedDict()
        off = 0
        for name, shape, _ in self.entries:
            n = int(math.prod(shape))
            self.offsets[name] = (off, n, tuple(shape))
            off += _align(n)          # 256-byte aligned starts: every view is 16-B aligned for vector access
        self.numel = off
        self.master = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.bf16 = torch.zeros(off, device=self.device, dtype=torch.bfloat16)
        self._tflat = self._tjobs = None      # flat buffer of transposed bf16 copies + device job table
        self._ttiles = self._tjobs_n = 0
        self._t_dirty = False                 # Linears registered since the transposed copies were last written
        self._lins = {}
        self._lns = {}
        self.version = -1       # bumped by refresh(); compared with master_version
        self.master_version = 0
        self.grad_ready_hook = None   # callable(group_name) set by the DP reducer
        if init:
            self.reset_parameters(seed)

    # ---- views -------------------------------------------------------------------------------------
    def p(self, name):
        o, n, shp = self.offsets[name]
        return self.master[o:o + n].view(shp)

    def g(self, name):
        o, n, shp = self.offsets[name]
        return self.grad[o:o + n].view(shp)

    def b16(self, name):
        o, n, shp = self.offsets[name]
        return self.bf16[o:o + n].view(shp)

    def names(self):
        return list(self.offsets.keys())

    def group_range(self, prefix):
        """[start, end) of the arena covered by entries whose name starts with `prefix` (must be contiguous)."""
        idx = [i for i, (n, _, _) in enumerate(self.entries) if n.startswith(prefix)]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), f"group {prefix} not contiguous"
        first = self.offsets[self.entries[idx[0]][0]]
        last = self.offsets[self.entries[idx[-1]][0]]
        return first[0], last[0] + _align(last[1])

    def reset_parameters(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        std = self.cfg.get('initializer_range', 0.02)
        for name, shape, kind in self.entries:
            if kind == 'ones':
                t = torch.ones(shape)
            elif kind == 'zeros':
                t = torch.zeros(shape)
            elif kind == 'conv':
                t = torch.randn(shape, generator=g).clamp_(-2, 2) * math.sqrt(1.0 / shape[1])
            elif kind == 'conv_hwio':                                        # variance_scaling, fan_in = kh*kw*ci
                t = torch.randn(shape, generator=g).clamp_(-2, 2) * math.sqrt(1.0 / (shape[0] * shape[1] * shape[2]))
            else:
                t = torch.randn(shape, generator=g).clamp_(-2, 2) * std      # truncated normal, transformer.py:166-168
            self.p(name).copy_(t)
        self.master_version += 1

    # ---- bf16 working copies -------------------------------------------------------------------------
    def refresh(self, force=False):
        """Re-derive every bf16 working copy from the fp32 masters (call after each optimizer step): one cast launch for
        the whole arena and ONE batched transpose launch for the transposed copies of all registered Linears."""
        if self.version == self.master_version and not force:
            return
        ops.cast_bf16(self.master, self.bf16)
        if self._lins:
            if self._tjobs is None or self._tjobs_n != len(self._lins) or self._t_dirty:
                self._build_transpose_jobs()
            self._t_dirty = False
            ops.cast_transpose_batched(self.master, self._tflat, self._tjobs, self._ttiles)
        self.version = self.master_version

    def _t_shape(self, name):
        out_dim, in_dim = self.offsets[name][2]
        ld = _align(out_dim, 128) if name == 'word_embeddings/word_embeddings' else out_dim   # padded reduction dim (LM-head dgrad)
        return in_dim, out_dim, ld

    def _build_transpose_jobs(self):
        """flat bf16 buffer holding every transposed copy + the device job table of merlot_cast_transpose_batched."""
        rows, off, tiles = [], 0, 0
        views = {}
        for name in self._lins:
            in_dim, out_dim, ld = self._t_shape(name)
            views[name] = (off, in_dim, ld)
            rows.append([self.offsets[name][0], off, out_dim, in_dim, ld, tiles])
            tiles += ((out_dim + 63) // 64) * ((in_dim + 63) // 64)
            off += _align(in_dim * ld, 64)
        self._tflat = torch.zeros(off, device=self.device, dtype=torch.bfloat16)
        self._tjobs = torch.tensor(rows, dtype=torch.int64).to(self.device)
        self._ttiles, self._tjobs_n = tiles, len(self._lins)
        for name, (o, in_dim, ld) in views.items():
            self._lins[name].wbT = self._tflat[o:o + in_dim * ld].view(in_dim, ld)

    def _flush_transposed(self):
        """Linears were registered after the last refresh: rebuild the table and fill every copy (one launch)."""
        self._t_dirty = False
        self._build_transpose_jobs()
        ops.cast_transpose_batched(self.master, self._tflat, self._tjobs, self._ttiles)

    def lin(self, scope, need_T=True, bias=True):
        """Linear handle for `<scope>/kernel` (+ `<scope>/bias`)."""
        key = scope + '/kernel' if (scope + '/kernel') in self.offsets else scope
        h = self._lins.get(key) if need_T else None
        if h is None:
            h = Lin(key, self)
            h.w, h.gw, h.wb = self.p(key), self.g(key), self.b16(key)
            bname = scope + '/bias'
            if bias and bname in self.offsets:
                h.b, h.gb = self.p(bname), self.g(bname)
            if need_T:
                self._lins[key] = h
                if self.version >= 0:
                    self._t_dirty = True                     # its transposed copy is made at the first read of any .wbT
        return h

    def ln(self, scope):
        h = self._lns.get(scope)
        if h is None:
            h = LN(scope)
            h.gamma, h.beta = self.p(scope + '/gamma'), self.p(scope + '/beta')
            h.ggamma, h.gbeta = self.g(scope + '/gamma'), self.g(scope + '/beta')
            self._lns[scope] = h
        return h

    def zero_grad(self):
        self.grad.zero_()

    def notify_ready(self, group):
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(group)

    # ---- TF-name interchange ---------------------------------------------------------------------------
    def view(self, flat, name):
        """the slice of an arena-shaped flat tensor (master, grad, an optimizer slot) that belongs to `name`."""
        o, n, shp = self.offsets[name]
        return flat[o:o + n].view(shp)

    def tf_names(self, name):
        """the TF variable names an arena entry is stored under in a reference checkpoint (q/k/v are fused here)."""
        if name.endswith('/qkv/kernel') or name.endswith('/qkv/bias'):
            base, leaf = name.rsplit('/qkv/', 1)
            return [f'{base}/{nm}/{leaf}' for nm in ('query_layer', 'key_layer', 'value_layer')]
        return [name]

    def load_tf_weights(self, tf_weights, strict=True, getter=None, suffix=''):
        """tf_weights: {TF variable name (+ suffix): tensor in TF layout} (see oracle.merlot_oracle.variable_shapes).
        strict=False is `tf.train.init_from_checkpoint` with the assignment map of utils/model_utils.py:388-413: only
        the variables the checkpoint holds are overwritten.  getter/suffix select another arena-shaped destination
        (e.g. the optimizer's `<name>/adam_m` slots).  -> names in tf_weights that nothing consumed."""
        H = self.cfg['hidden_size']
        P = self.cfg['patch_size']
        getter = getter or self.p
        used = set()
        self.initialized_from_checkpoint = getattr(self, 'initialized_from_checkpoint', set())

        def fetch(tf_name):
            key = tf_name + suffix
            if key not in tf_weights:
                if strict:
                    raise KeyError(key)
                return None
            used.add(key)
            self.initialized_from_checkpoint.add(key)
            t = tf_weights[key]
            return (t if isinstance(t, torch.Tensor) else torch.as_tensor(t)).float()

        for name in self.names():
            dst = getter(name)
            if name.endswith('/qkv/kernel') or name.endswith('/qkv/bias'):
                leaf = name.rsplit('/', 1)[1]
                for i, tf_name in enumerate(self.tf_names(name)):
                    t = fetch(tf_name)
                    if t is not None:
                        dst[i * H:(i + 1) * H].copy_(t.t() if leaf == 'kernel' else t)
                continue
            t = fetch(name)
            if t is None:
                continue
            if '/resnet50lite/' in name:
                dst.copy_(t.reshape(dst.shape))                          # HWIO kernels / GroupNorm, as is
            elif name.endswith('conv_postresnet_proj/kernel'):
                dst.copy_(t.reshape(-1, H).t())                          # [1, 1, C, H]
            elif name.endswith('conv2d/kernel'):
                dst.copy_(t.reshape(P * P * 3, H).t())                   # HWIO [P,P,3,H]
            elif name.endswith('/kernel'):
                dst.copy_(t.t())                                         # [in, out]
            else:
                dst.copy_(t.reshape(dst.shape))
        self.master_version += 1
        return sorted(set(tf_weights.keys()) - used)

    def _export(self, getter, keep_dtype=False):
        H = self.cfg['hidden_size']
        P = self.cfg['patch_size']
        out = {}
        for name in self.names():
            t = getter(name).detach().cpu()
            t = t if keep_dtype else t.float()
            if name.endswith('/qkv/kernel') or name.endswith('/qkv/bias'):
                base, leaf = name.rsplit('/qkv/', 1)
                for i, nm in enumerate(['query_layer', 'key_layer', 'value_layer']):
                    part = t[i * H:(i + 1) * H]
                    out[f'{base}/{nm}/{leaf}'] = part.t().contiguous() if leaf == 'kernel' else part.clone()
            elif '/resnet50lite/' in name:
                out[name] = t.clone()
            elif name.endswith('conv_postresnet_proj/kernel'):
                out[name] = t.t().reshape(1, 1, -1, H).contiguous()
            elif name.endswith('conv2d/kernel'):
                out[name] = t.t().reshape(P, P, 3, H).contiguous()
            elif name.endswith('/kernel'):
                out[name] = t.t().contiguous()
            else:
                out[name] = t.clone()
        return out

    def export_tf_weights(self):
        return self._export(self.p)

    def export_tf_grads(self):
        return self._export(self.g)
