"""autograd glue between the model code and the HIP kernels.

Granularity is deliberately coarse: one autograd.Function per transformer STACK (forward and backward walk the
12 layers calling the C-ABI directly, activations kept in bf16), one per Linear / LayerNorm / embedding site of
the small heads.  Parameter gradients are not returned to autograd: every backward ACCUMULATES straight into the
flat fp32 gradient arena of `ParamStore` (wgrad GEMMs with beta=1, atomically-added bias / LayerNorm grads) and
then signals `store.notify_ready(group)` so the DP reducer can launch that bucket's all-reduce while the rest of
the backward is still running.

Reference being replaced: utils/transformer.py:171-247 (stack), :33-138 (attention), :141-163 (MLP),
utils/model_utils.py:113-130 (LayerNorm), tf.gradients at utils/optimization.py:176.
"""
import torch

from . import ops
from .ops import BF16, F32, EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU


class LayerW(object):
    __slots__ = ('ln1', 'qkv', 'proj', 'ln2', 'fc1', 'fc2', 'name')


class StackW(object):
    """Weights of one transformer stack (`scope` = 'encoder' or 'vision_backbone/vision_transformer')."""

    def __init__(self, store, scope, num_layers):
        self.store = store
        self.scope = scope
        self.layers = []
        for l in range(num_layers):
            ls = f'{scope}/layer{l:02d}'
            w = LayerW()
            w.name = ls
            w.ln1 = store.ln(f'{ls}/LayerNorm_attn_ln0')
            w.qkv = store.lin(f'{ls}/qkv')
            w.proj = store.lin(f'{ls}/context_projection_layer')
            w.ln2 = store.ln(f'{ls}/LayerNorm_mlp_ln0')
            w.fc1 = store.lin(f'{ls}/intermediate')
            w.fc2 = store.lin(f'{ls}/output')
            self.layers.append(w)
        self.ln_final = store.ln(f'{scope}/LayerNorm_ln_final')


def _fwd_linear(x, lin, fp8, x8=None, row_scale=None, **kw):
    """One forward Linear of the stack.  fp8 (BASELINE config #5, `model.fp8_forward`): e4m3 operands, fp32 accumulation on the
    fp8 kernel; bias, epilogue and output dtype are those of the bf16 path.  The activation arrives either already quantised
    with per-ROW scales by the LayerNorm that produced it (x8, row_scale: QKV and fc1) or is quantised here with one
    per-tensor scale (two passes over x: fc2); weights: per-tensor, per call.  The backward always consumes the bf16
    tensors (x is saved as before)."""
    if not fp8:
        return ops.gemm_nt(x, lin.wb, bias=lin.b, **kw)
    w8, sw = _w8(lin._store, lin.wb) if lin._store is not None else ops.quantize_e4m3(lin.wb)
    if x8 is not None:
        return ops.gemm_fp8_nt(x8, None, w8, sw, bias=lin.b, a_row_scale=row_scale, **kw)
    x8, sx = ops.quantize_e4m3(x)
    return ops.gemm_fp8_nt(x8, sx, w8, sw, bias=lin.b, **kw)


def _w8(store, wt):
    """the e4m3 copy + scale block of a bf16 weight view (per-tensor, current scaling), made ONCE per version of the working copies: the weights change at
    the optimizer step only, while a step reads each of them two or three times (forward, and the transposed copy in the backward) -- until round 6 every use
    ran its own amax + convert launches (~2.8 ms per config-#5 step)."""
    cache = store.__dict__.setdefault('_w8_cache', {})
    key = (wt.data_ptr(), tuple(wt.shape))
    ent = cache.get(key)
    if ent is None or ent[0] != store.version:
        w8, sw = ops.quantize_e4m3(wt, out=ent[1] if ent is not None else None)
        ent = cache[key] = (store.version, w8, sw)
    return ent[1], ent[2]


class F8Scales(object):
    """Per-tensor scale blocks of the 8-bit backward (BASELINE config #5, `model.fp8_backward`), one per (stack use, layer, tensor) site, kept on the ParamStore
    across steps in ONE pooled tensor (block = float[4] {s, 1/s, amax s was made from, amax recorded since}).  A site's FIRST step is a calibration: its
    tensor is produced in bf16 as without the option and quantised by a "current" pass (ops.quantize_f8: amax pass + convert pass) -- from then on
    the PRODUCING launch writes the 8-bit copy itself with the block's scale and records the tensor's amax (delayed scaling; ops.ln_fwd_q8t,
    ops.gemm_fp8_nt_q8, ops.gemm_nt_q8, ops.ln_bwd(db8_block=...)), or, without 'fuse', a one-pass quantisation does.  step_begin() (once per model =
    once per step, in front of the first stack) turns the recorded amaxes into the step's scales -- never between a forward that quantised with a
    block and the backward that dequantises with it.  refresh(): forget the history (the next step re-calibrates)."""
    CAP = 2048

    def __init__(self, device):
        self.pool = torch.zeros((self.CAP, 4), device=device, dtype=F32)
        self.fmts = torch.zeros(self.CAP, device=device, dtype=torch.int32)
        self.index = {}
        self.calibrated = set()
        self.dirty = False

    def block(self, key, fmt):
        i = self.index.get(key)
        if i is None:
            i = self.index[key] = len(self.index)
            if i >= self.CAP:
                raise RuntimeError("F8Scales: more sites than blocks")
            if fmt != 0:
                self.fmts[i] = int(fmt)
        return self.pool[i]

    def ready(self, key):
        return key in self.calibrated

    def calibrate(self, key, x, fmt, out=None):
        """the site's calibration step: current scaling into its block -> (x8, block)"""
        blk = self.block(key, fmt)
        y, _ = ops.quantize_f8(x, fmt, current_into=blk, out=out)
        self.calibrated.add(key)
        self.dirty = True
        return y, blk

    def quantize(self, key, x, fmt):
        """the stand-alone pass (no producer writes this tensor's copy): current in the site's first step, delayed after"""
        if not self.ready(key):
            return self.calibrate(key, x, fmt)
        blk = self.block(key, fmt)
        y, _ = ops.quantize_f8(x, fmt, scale=blk)
        return y, blk

    def step_begin(self):
        if self.dirty and self.index:
            ops.f8_scale_rotate(self.pool, len(self.index), self.fmts)
            self.dirty = False

    def refresh(self):
        self.calibrated.clear()


def f8_scales(store, create=True):
    f8 = getattr(store, 'f8_scales', None)
    if f8 is None and create:
        f8 = store.f8_scales = F8Scales(store.grad.device)
    return f8


def f8_step_begin(store):
    """called by the model in front of its first stack (once per step): the amaxes the last step's producers recorded become this step's scales"""
    f8 = f8_scales(store, create=False)
    if f8 is not None:
        f8.step_begin()


def _f8_modes(opt):
    """`model.fp8_backward`: False / None, or a comma-separated list of what runs on 8-bit operands in the backward: w1, w2, wqkv, wproj = that weight
    gradient through merlot_gemm_f8_tn (gradient operand e5m2, activation operand e4m3; 'e4m3' in the list: gradients in e4m3 too); 'fuse': the 8-bit
    copies of x1 / x2 (LayerNorm), a (fc1's GELU epilogue), du (the GELU' epilogue) and the branch gradients (LayerNorm backward) come out of the launches
    that produce those tensors instead of quantising passes (needs fp8_forward and at least 2 048 rows; the attention output has no such
    producer, dqkv only where the tiled attention backward runs: they keep their pass); 'noa': with 'fuse' and w2, fc1 does not store its bf16 output at all (fc2 reads the e4m3 copy); 'dgradqkv': with 'fuse' and wqkv, the QKV input-gradient GEMM (K = 2 304) reads dqkv's copy too (then the one quantising pass over dqkv pays);
    'dgrad1': with 'fuse' and w1, fc1's input-gradient GEMM
    (K = 3 072) reads du's 8-bit copy as well and the GELU' epilogue does not store du in bf16 at all.  True = 'w1,w2,fuse'."""
    if not opt:
        return frozenset()
    if opt is True:
        return frozenset(('w1', 'w2', 'fuse'))
    return frozenset(t.strip() for t in str(opt).split(',') if t.strip())


def _wgrad(f8, key, dy, x, gw, gfmt, **kw):
    """gw[out, in] += dy[T, out]^T x[T, in]: bf16 (f8 None) or on 8-bit copies quantised here (f8 = the stack's F8Scales)."""
    if f8 is None or dy.shape[0] < 2048:
        return ops.gemm_tn(dy, x, gw, **kw)
    dy8, sdy = f8.quantize(key + '/dy', dy, gfmt)
    x8, sx = f8.quantize(key + '/x', x, ops.F8_E4M3)
    colsum_a = kw.pop('colsum_a', None)
    if colsum_a is not None:                            # (the 8-bit kernel does not sum its A operand's columns: the stand-alone pass)
        ops.colsum_bf16(dy[:, :colsum_a.numel()], colsum_a)
    return ops.gemm_f8_tn(dy8, sdy, x8, sx, gw, **kw)


# The LayerNorm behind each residual GEMM from that GEMM's own launch (merlot_gemm_bf16_nt_ln, ABI v8).  OFF: built, parity-tested (tests/test_gemm_ln_gpu.py) and
# measured -- the fused launch costs what the stand-alone LayerNorm kernel costs (proj + LN: -43 ... -59 us, fc2 + LN: +53 ... +73 us per launch at the bench's
# ViT row count; the step 0.2 - 0.4 % SLOWER on three boxes, profiles/r06_g_ln_fold_ab.txt, r06_g_bench_*.json): every tile pays 5 - 9 k cycles of segment
# statistics in an epilogue during which the matrix pipe idles anyway, and one CU normalises a 256 x 768 block alone in 27 k cycles where the separate kernel
# spreads it over the chip at 4.9 TB/s.  bench.py --ln-fold switches it on.
FUSE_LN = False
# The fused-QKV bias gradient (its Q third) from the weight-gradient launch's own A fragments (merlot_gemm_bf16_tn_cs) instead of a pass over dQKV.
TN_COLSUM = True


def _site_seed(seed, layer, site):
    return (int(seed) * 1000003 + layer * 16 + site) & 0xFFFFFFFFFFFFFFFF


class TransformerStackFn(torch.autograd.Function):
    """hidden [B*S, H] bf16 -> LN_final(stack(hidden)) [B*S, H] bf16.

    opts: dict(heads, dropout_p, seed, fp8 (True: QKV / fc1 / fc2 forward GEMMs on e4m3 operands; 'ln': only the two fed by a
               LayerNorm, whose e4m3 copy costs no extra pass; 'all' (rounds 2-5: + the attention forward on e4m3, removed in round 6) = True), colsum (f32 [B,S] accumulated over layers, all query rows, weight 1/heads),
               log_lo / log_hi (f32 [B,S], valid pairs only, queries < / >= log_split), num_layers,
               log_in_backward (True: when a backward will run, log_lo / log_hi are filled by the attention BACKWARD -- its dK / dV
               pass forms P anyway -- instead of a second Q K^T walk in every forward launch; `log_done` (callable) is invoked once
               the last layer's backward has been queued.  Metrics only: nothing in the forward reads them),
               seg (int32 [S]: the disable_pairwise_lang_attn block mask, model/modeling.py:160-168))
    """

    @staticmethod
    def forward(ctx, h, stack, B, S, valid, opts):
        heads = opts['heads']
        p = float(opts.get('dropout_p', 0.0))
        seed = opts.get('seed', 0)
        nl = opts.get('num_layers', len(stack.layers))
        colsum = opts.get('colsum')
        log_lo, log_hi = opts.get('log_lo'), opts.get('log_hi')
        seg = opts.get('seg')                              # int32 [S] block mask of disable_pairwise_lang_attn, or None
        fp8 = bool(opts.get('fp8', False))
        fp8_fc2 = fp8 and opts.get('fp8') != 'ln'
        need_bwd = ctx.needs_input_grad[0]
        log_bwd = bool(opts.get('log_in_backward', False)) and need_bwd and log_lo is not None
        ctx.log = (log_lo, log_hi, opts.get('log_split'), opts.get('log_done')) if log_bwd else None
        if log_bwd:
            log_lo = log_hi = None                        # the forward launches carry no log side output
        saved = []
        h = h.contiguous()
        # Round 6: the LayerNorm behind a residual GEMM (proj -> LN2, fc2 -> the next layer's LN1 / the final LN) comes out of that GEMM's launch
        # (ops.gemm_nt_ln, ABI v8) -- bf16 path only; the fp8 path's LayerNorm also emits the e4m3 copy and stays a launch of its own
        fuse_ln = FUSE_LN and not fp8 and h.shape[1] % 256 == 0
        nxt = None                                         # (x1, mean1, rstd1) of this layer when the previous layer's fc2 launch produced them
        # the 8-bit backward (`fp8_backward`): with 'fuse' the forward's producers write the e4m3 copies the weight gradients will read
        f8m = _f8_modes(opts.get('fp8_bwd')) if need_bwd else frozenset()
        site = opts.get('f8_site', stack.scope)           # which use of the stack (the joint and the text-only pass share weights, not activations)
        T = h.shape[0]
        fuse8 = 'fuse' in f8m and fp8 and T >= 2048       # (any row count: the copies are allocated in whole K-tiles of 128 rows, the padding zero)
        f8 = f8_scales(stack.store) if fuse8 else None
        f8_x1, f8_x2, f8_a = fuse8 and 'wqkv' in f8m, fuse8 and 'w1' in f8m, fuse8 and 'w2' in f8m
        no_a = f8_a and 'noa' in f8m

        def ln_q8t(key, hin, ln):
            """LayerNorm whose e4m3 copy carries ONE per-tensor factor: -> (x8, block, mean, rstd); no bf16 output once the site is calibrated"""
            if f8.ready(key):
                _, x8, mean, rstd = ops.ln_fwd_q8t(hin, ln.gamma, ln.beta, f8.block(key, ops.F8_E4M3))
                f8.dirty = True
                return x8, f8.block(key, ops.F8_E4M3), mean, rstd
            x16, _, mean, rstd = ops.ln_fwd(hin, ln.gamma, ln.beta)
            x8, blk = f8.calibrate(key, x16, ops.F8_E4M3)
            return x8, blk, mean, rstd

        for l in range(nl):
            w = stack.layers[l]
            x1q = sx1 = sx2 = a8 = sa = None
            if f8_x1:
                x1q, sx1, mean1, rstd1 = ln_q8t(f'{site}/{l}/x1', h, w.ln1)
                x1 = None
                w8, sw = _w8(stack.store, w.qkv.wb)
                qkv = ops.gemm_fp8_nt(x1q[:T], sx1, w8, sw, bias=w.qkv.b)
            elif fp8:
                x1, x1q, rs1, mean1, rstd1 = ops.ln_fwd_q8(h, w.ln1.gamma, w.ln1.beta)
                qkv = _fwd_linear(x1, w.qkv, True, x8=x1q, row_scale=rs1)
                x1q = None
            else:
                if nxt is not None:
                    x1, mean1, rstd1 = nxt
                else:
                    x1, _, mean1, rstd1 = ops.ln_fwd(h, w.ln1.gamma, w.ln1.beta)
                qkv = _fwd_linear(x1, w.qkv, False)
            # the attention-probability side outputs (a7) come out of the forward launch: K is still resident in LDS
            if colsum is not None and log_lo is None:
                ctx_, lse = ops.attention_fwd(qkv, B, S, heads, valid, seg=seg, colsum_lo=colsum, valid_q_only=False,
                                              weight=1.0 / heads)
            elif log_lo is not None and colsum is None:
                ctx_, lse = ops.attention_fwd(qkv, B, S, heads, valid, seg=seg, colsum_lo=log_lo, colsum_hi=log_hi,
                                              qsplit=opts['log_split'], valid_q_only=True, weight=1.0 / heads)
            else:
                ctx_, lse = ops.attention_fwd(qkv, B, S, heads, valid, seg=seg)
                if colsum is not None:
                    ops.attention_colsum(qkv, lse, B, S, heads, colsum, valid=valid, valid_q_only=False, weight=1.0 / heads,
                                         seg=seg)
                if log_lo is not None:
                    ops.attention_colsum(qkv, lse, B, S, heads, log_lo, log_hi, qsplit=opts['log_split'], valid=valid,
                                         valid_q_only=True, weight=1.0 / heads, seg=seg)
            if fuse_ln:
                h_mid, x2, mean2, rstd2 = ops.gemm_nt_ln(ctx_, w.proj.wb, w.ln2.gamma, w.ln2.beta, bias=w.proj.b, aux_in=h, dropout_p=p,
                                                         dropout_seed=_site_seed(seed, l, 0))
                x2q = rs2 = None
            else:
                h_mid = ops.gemm_nt(ctx_, w.proj.wb, bias=w.proj.b, epilogue=EPI_RESIDUAL, aux_in=h, dropout_p=p,
                                    dropout_seed=_site_seed(seed, l, 0))
                if f8_x2:
                    x2q, sx2, mean2, rstd2 = ln_q8t(f'{site}/{l}/x2', h_mid, w.ln2)
                    x2, rs2 = None, None
                elif fp8:
                    x2, x2q, rs2, mean2, rstd2 = ops.ln_fwd_q8(h_mid, w.ln2.gamma, w.ln2.beta)
                else:
                    x2, _, mean2, rstd2 = ops.ln_fwd(h_mid, w.ln2.gamma, w.ln2.beta)
                    x2q = rs2 = None
            u = torch.empty((T, w.fc1.wb.shape[0]), device=h.device, dtype=BF16)
            if f8_a:
                # fc1 on e4m3 operands whose GELU epilogue ALSO writes the e4m3 copy of a (and, with 'noa', nothing else of it)
                key = f'{site}/{l}/a'
                w8, sw = _w8(stack.store, w.fc1.wb)
                if f8.ready(key):
                    sa = f8.block(key, ops.F8_E4M3)
                    a, a8 = ops.gemm_fp8_nt_q8(x2q[:T], sx2, w8, sw, sa, bias=w.fc1.b, aux_out=u, a_row_scale=rs2, keep_bf16=not no_a)
                    f8.dirty = True
                else:
                    a = ops.gemm_fp8_nt(x2q[:T], sx2, w8, sw, bias=w.fc1.b, a_row_scale=rs2, epilogue=EPI_GELU, aux_out=u)
                    a8, sa = f8.calibrate(key, a, ops.F8_E4M3)
                    if no_a:
                        a = None
            elif f8_x2:
                w8, sw = _w8(stack.store, w.fc1.wb)
                a = ops.gemm_fp8_nt(x2q[:T], sx2, w8, sw, bias=w.fc1.b, epilogue=EPI_GELU, aux_out=u)
            else:
                a = _fwd_linear(x2, w.fc1, fp8, x8=x2q, row_scale=rs2, epilogue=EPI_GELU, aux_out=u)
            if not f8_x2:
                x2q = None
            if fuse_ln:
                ln_next = stack.layers[l + 1].ln1 if l + 1 < nl else stack.ln_final
                h_out, xn, meann, rstdn = ops.gemm_nt_ln(a, w.fc2.wb, ln_next.gamma, ln_next.beta, bias=w.fc2.b, aux_in=h_mid, dropout_p=p,
                                                         dropout_seed=_site_seed(seed, l, 1))
                nxt = (xn, meann, rstdn)
            elif a8 is not None and (fp8_fc2 or no_a):
                w8, sw = _w8(stack.store, w.fc2.wb)          # fc2 reads the copy fc1's epilogue wrote: no quantising pass over a
                h_out = ops.gemm_fp8_nt(a8[:T], sa, w8, sw, bias=w.fc2.b, epilogue=EPI_RESIDUAL, aux_in=h_mid, dropout_p=p,
                                        dropout_seed=_site_seed(seed, l, 1))
            else:
                h_out = _fwd_linear(a, w.fc2, fp8_fc2, epilogue=EPI_RESIDUAL, aux_in=h_mid, dropout_p=p,
                                    dropout_seed=_site_seed(seed, l, 1))
            if need_bwd:
                saved.append((h, mean1, rstd1, x1, qkv, ctx_, lse, h_mid, mean2, rstd2, x2, u, a, (x1q, sx1, x2q, sx2, a8, sa)))
            h = h_out
        if fuse_ln and nxt is not None:
            y, meanf, rstdf = nxt
        else:
            y, _, meanf, rstdf = ops.ln_fwd(h, stack.ln_final.gamma, stack.ln_final.beta)
        ctx.stack, ctx.B, ctx.S, ctx.valid, ctx.heads, ctx.p, ctx.seed, ctx.nl = stack, B, S, valid, heads, p, seed, nl
        ctx.seg = seg
        ctx.f8m, ctx.f8_site, ctx.fuse8 = f8m, site, fuse8
        ctx.saved = saved
        ctx.final = (h, meanf, rstdf)
        return y

    @staticmethod
    def backward(ctx, dy):
        stack, B, S, valid, heads, p, seed = ctx.stack, ctx.B, ctx.S, ctx.valid, ctx.heads, ctx.p, ctx.seed
        store = stack.store
        hL, meanf, rstdf = ctx.final
        dy = dy.contiguous()
        nl = ctx.nl
        f8m = ctx.f8m
        gfmt = ops.F8_E4M3 if 'e4m3' in f8m else ops.F8_E5M2
        f8 = f8_scales(store) if f8m else None            # the scale history lives on the ParamStore (the StackW objects are per model = per step)
        site, fuse8 = ctx.f8_site, ctx.fuse8

        def ln_bwd8(key, want8, *a, **kw):
            """ln_bwd whose branch gradient also comes as an 8-bit copy (from the same launch once the site is calibrated): -> (dx, d_branch, d_branch8, block)"""
            if not (want8 and fuse8):
                dx, dbr = ops.ln_bwd(*a, **kw)
                return dx, dbr, None, None
            if f8.ready(key):
                blk = f8.block(key, gfmt)
                dx, dbr, dbr8 = ops.ln_bwd(*a, db8_block=blk, db8_fmt=gfmt, **kw)
                f8.dirty = True
                return dx, dbr, dbr8, blk
            dx, dbr = ops.ln_bwd(*a, **kw)
            dbr8, blk = f8.calibrate(key, dbr, gfmt)
            return dx, dbr, dbr8, blk

        # every ln_bwd also emits the branch gradient of the sub-layer BELOW it (dropout mask regenerated from the
        # counter hash) and accumulates that sub-layer's bias gradient -- no separate dropout / column-sum passes.
        dh, db2, db2_8, sdb2 = ln_bwd8(f'{site}/{nl - 1}/db2', 'w2' in f8m, dy, hL, meanf, rstdf, stack.ln_final.gamma, stack.ln_final.ggamma,
                                       stack.ln_final.gbeta, branch_bias_grad=stack.layers[nl - 1].fc2.gb, drop_p=p, drop_seed=_site_seed(seed, nl - 1, 1))
        for l in range(nl - 1, -1, -1):
            w = stack.layers[l]
            h, mean1, rstd1, x1, qkv, ctx_, lse, h_mid, mean2, rstd2, x2, u, a, (x1q, sx1, x2q, sx2, a8, sa) = ctx.saved[l]
            ctx.saved[l] = None
            big = h.shape[0] >= 2048                      # (merlot_gemm_f8_tn's floor; below it the bf16 kernel)
            # ---- MLP branch: h_out = h_mid + drop(fc2(gelu(fc1(LN2(h_mid)))))          db2 = d(fc2 output)
            if 'w2' in f8m and big and a8 is not None:    # both operands came out of their producers (or this step calibrates them)
                ops.gemm_f8_tn(db2_8, sdb2, a8, sa, w.fc2.gw)
            elif 'w2' in f8m and big:
                _wgrad(f8, f'{site}/{l}/w2', db2, a, w.fc2.gw, gfmt)
            else:
                ops.gemm_tn(db2, a, w.fc2.gw)             # dW2[H, I]
            a = a8 = db2_8 = None
            du8 = sdu = None
            dg1 = 'dgrad1' in f8m and fuse8 and 'w1' in f8m      # fc1's input gradient reads du's copy too: du itself is never stored
            if 'w1' in f8m and fuse8:
                key = f'{site}/{l}/du'
                if f8.ready(key):
                    sdu = f8.block(key, gfmt)
                    du, du8 = ops.gemm_nt_q8(db2, w.fc2.wbT, sdu, gfmt, epilogue=EPI_DGELU, aux_in=u, colsum_out=w.fc1.gb, keep_bf16=not dg1)
                    f8.dirty = True
                else:
                    du = ops.gemm_nt(db2, w.fc2.wbT, epilogue=EPI_DGELU, aux_in=u, colsum_out=w.fc1.gb)
                    du8, sdu = f8.calibrate(key, du, gfmt)
            else:
                du = ops.gemm_nt(db2, w.fc2.wbT, epilogue=EPI_DGELU, aux_in=u, colsum_out=w.fc1.gb)   # [T, I] + fc1's bias grad
            if du8 is not None and x2q is not None:
                ops.gemm_f8_tn(du8, sdu, x2q, sx2, w.fc1.gw)
            elif 'w1' in f8m and big:
                _wgrad(f8, f'{site}/{l}/w1', du, x2, w.fc1.gw, gfmt)
            else:
                ops.gemm_tn(du, x2, w.fc1.gw)             # dW1[I, H]
            if dg1:
                wt8, swt = _w8(store, w.fc1.wbT)
                dx2 = ops.gemm_f8_nt(du8[:h.shape[0]], sdu, wt8, swt)
            else:
                dx2 = ops.gemm_nt(du, w.fc1.wbT)
            du = du8 = x2q = None
            dh_mid, db1, db1_8, sdb1 = ln_bwd8(f'{site}/{l}/db1', 'wproj' in f8m, dx2, h_mid, mean2, rstd2, w.ln2.gamma, w.ln2.ggamma, w.ln2.gbeta, dres=dh,
                                               branch_bias_grad=w.proj.gb, drop_p=p, drop_seed=_site_seed(seed, l, 0))
            # ---- attention branch: h_mid = h + drop(proj(attn(qkv(LN1(h)))))             db1 = d(proj output)
            if db1_8 is not None:                          # (the attention output has no producer that writes its copy: one pass)
                c8, sc8 = f8.quantize(f'{site}/{l}/wproj/x', ctx_, ops.F8_E4M3)
                ops.gemm_f8_tn(db1_8, sdb1, c8, sc8, w.proj.gw)
                c8 = db1_8 = None
            else:
                _wgrad(f8 if 'wproj' in f8m else None, f'{site}/{l}/wproj', db1, ctx_, w.proj.gw, gfmt)
            # bias gradients of the fused QKV projection without a pass over dQKV [T, 3D]:
            #   V: sum_k dV_k = sum_q dO_q because every softmax row sums to 1 -> the column sums of dO, from the epilogue of the
            #      GEMM that produces it;  K: identically 0 (adding a constant to every key shifts each query's scores by a
            #      constant: softmax does not move);  Q: the column sums of the first third of dQKV only.
            D = heads * 64
            exact_rows = True                             # (False: sum all three thirds -- what the e4m3 attention forward of rounds 2-5 needed; it is gone)
            dctx = ops.gemm_nt(db1, w.proj.wbT, colsum_out=w.qkv.gb[2 * D:3 * D] if exact_rows else None)
            akw = dict(seg=ctx.seg)
            if ctx.log is not None:
                akw.update(log_lo=ctx.log[0], log_hi=ctx.log[1], log_split=ctx.log[2], log_weight=1.0 / heads)
            dq8 = sdq = None
            qkey = f'{site}/{l}/wqkv/dy'
            # dqkv's 8-bit copy from the attention backward's own launches where the tiled kernel pair runs (config #5's 578 / 2 832 tokens); elsewhere a pass below
            if 'wqkv' in f8m and big and x1q is not None and f8.ready(qkey) and ops.attention_bwd_writes_q8(S, ctx.seg is not None):
                sdq = f8.block(qkey, gfmt)
                dqkv, dq8 = ops.attention_bwd(qkv, ctx_, dctx, lse, B, S, heads, valid, q8_block=sdq, q8_fmt=gfmt, **akw)
                f8.dirty = True
            else:
                dqkv = ops.attention_bwd(qkv, ctx_, dctx, lse, B, S, heads, valid, **akw)
            # (round 6: the Q third's column sums come out of the weight-gradient launch below -- its A fragments are dQKV)
            cs_q = w.qkv.gb[:D] if exact_rows else w.qkv.gb
            if 'wqkv' in f8m and big and x1q is not None:  # x1's copy came out of its LayerNorm; dqkv's out of the attention backward where the tiled pair
                if dq8 is None:                            # runs -- elsewhere (the fused / persistent kernels of <= 512 tokens) one pass (delayed scale), which
                    dq8, sdq = f8.quantize(qkey, dqkv, gfmt)   # pays once BOTH of dqkv's consumers read the copy ('dgradqkv')
                ops.colsum_bf16(dqkv[:, :cs_q.numel()], cs_q)
                ops.gemm_f8_tn(dq8, sdq, x1q, sx1, w.qkv.gw)
                x1q = None
            elif TN_COLSUM:
                _wgrad(f8 if 'wqkv' in f8m else None, f'{site}/{l}/wqkv', dqkv, x1, w.qkv.gw, gfmt, colsum_a=cs_q)
            else:
                ops.colsum_bf16(dqkv[:, :D] if exact_rows else dqkv, cs_q)
                ops.gemm_tn(dqkv, x1, w.qkv.gw)
            if dq8 is not None and 'dgradqkv' in f8m:      # the QKV input gradient (K = 2 304) on the same copy
                wt8, swt = _w8(store, w.qkv.wbT)
                dx1 = ops.gemm_f8_nt(dq8[:dqkv.shape[0]], sdq, wt8, swt)
            else:
                dx1 = ops.gemm_nt(dqkv, w.qkv.wbT)
            dq8 = None
            if l > 0:
                dh, db2, db2_8, sdb2 = ln_bwd8(f'{site}/{l - 1}/db2', 'w2' in f8m, dx1, h, mean1, rstd1, w.ln1.gamma, w.ln1.ggamma, w.ln1.gbeta, dres=dh_mid,
                                               branch_bias_grad=stack.layers[l - 1].fc2.gb, drop_p=p, drop_seed=_site_seed(seed, l - 1, 1))
            else:
                dh = ops.ln_bwd(dx1, h, mean1, rstd1, w.ln1.gamma, w.ln1.ggamma, w.ln1.gbeta, dres=dh_mid)
            store.notify_ready(w.name)
        store.notify_ready(stack.scope + '/LayerNorm_ln_final')
        if ctx.log is not None and ctx.log[3] is not None:
            ctx.log[3]()                                 # every layer's log sums are queued: the owner finalises its metrics
        ctx.log = None                                   # (the callback belongs to the model: do not keep the model alive through this node)
        ctx.saved = None
        ctx.final = None
        return dh, None, None, None, None, None


def transformer_stack(h, stack, B, S, valid, opts):
    return TransformerStackFn.apply(h, stack, B, S, valid, opts)


def split_bf16(x):
    """fp32 -> (hi, lo) bf16 with hi + lo == x to ~2^-17 relative: lets the bf16 MFMA GEMMs carry fp32 activations
    of the small heads (the reference keeps contrastive / temporal / lm-head activations fp32, SURVEY 8 dtype policy)."""
    hi = ops.cast_bf16(x)
    lo = ops.cast_bf16((x - hi.float()).contiguous())
    return hi, lo


class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) for the small heads.  x: [T, in] f32 or bf16 -> y f32 (heads are kept fp32 outside
    the MFMA contraction, model/modeling.py:184).  act in {None, 'gelu'}.

    fp32 inputs take the two-term path: activations and their gradients enter the bf16 MFMA GEMMs as hi + lo pairs
    (x_hi W + x_lo W, fp32 accumulation across the launches), weights as their bf16 working copy -- the reference's
    own policy (fp32 head activations, bf16-cast variables).  The heads are a few hundred rows: the extra launches
    are noise in the step, while single-term bf16 made gradients that cancel heavily (everything behind
    l2-normalise at temperature 0.05) lose a digit."""

    @staticmethod
    def forward(ctx, x, lin, act):
        two = x.dtype == F32
        if two:
            xb, xlo = split_bf16(x.contiguous())
            y = ops.gemm_nt(xb, lin.wb, bias=lin.b, out_dtype=F32)
            ops.gemm_nt(xlo, lin.wb, out=y, accumulate=True)
        else:
            xb, xlo = x, None
            y = ops.gemm_nt(xb, lin.wb, bias=lin.b, out_dtype=F32)
        ctx.lin, ctx.act, ctx.xb, ctx.xlo, ctx.x_dtype = lin, act, xb, xlo, x.dtype
        if act == 'gelu':
            ctx.pre = y
            y = ops.gelu_fwd(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lin = ctx.lin
        dy = dy.contiguous()
        if ctx.act == 'gelu':
            dy = ops.gelu_bwd(dy, ctx.pre)
        two = ctx.xlo is not None
        if two:
            dyb, dylo = split_bf16(dy)
        else:
            dyb, dylo = ops.cast_bf16(dy), None
        if lin.gb is not None:
            ops.colsum_bf16(dyb, lin.gb)
            if two:
                ops.colsum_bf16(dylo, lin.gb)
        out_dim = lin.w.shape[0]
        if out_dim % 2 != 0:
            raise ValueError("LinearFn: odd output width unsupported")
        ops.gemm_tn(dyb, ctx.xb, lin.gw)
        if two:
            ops.gemm_tn(dylo, ctx.xb, lin.gw)
            ops.gemm_tn(dyb, ctx.xlo, lin.gw)
        dx = None
        if ctx.needs_input_grad[0]:
            out_dt = F32 if ctx.x_dtype == F32 else BF16
            # K of the dgrad = out_dim must be a multiple of 64; the 4-wide temporal logits pad through wbT rows
            if out_dim % 64 == 0:
                dx = ops.gemm_nt(dyb, lin.wbT, out_dtype=out_dt)
                if two:
                    ops.gemm_nt(dylo, lin.wbT, out=dx, accumulate=True)
            else:
                kp = (out_dim + 63) // 64 * 64
                wp = torch.zeros((lin.wbT.shape[0], kp), device=dyb.device, dtype=BF16)
                wp[:, :out_dim] = lin.wbT
                dx = None
                for part in ((dyb, dylo) if two else (dyb,)):
                    dyp = torch.zeros((part.shape[0], kp), device=part.device, dtype=BF16)
                    dyp[:, :out_dim] = part
                    if dx is None:
                        dx = ops.gemm_nt(dyp, wp, out_dtype=out_dt)
                    else:
                        ops.gemm_nt(dyp, wp, out=dx, accumulate=True)
        return dx, None, None


def linear(x, lin, act=None):
    return LinearFn.apply(x, lin, act)


class LayerNormFn(torch.autograd.Function):
    """LN on f32 rows; output f32 or bf16."""

    @staticmethod
    def forward(ctx, x, ln, out_bf16):
        x = x.contiguous()
        y16, y32, mean, rstd = ops.ln_fwd(x, ln.gamma, ln.beta, out_bf16=out_bf16, out_f32=not out_bf16)
        ctx.ln, ctx.x, ctx.mean, ctx.rstd = ln, x, mean, rstd
        return y16 if out_bf16 else y32

    @staticmethod
    def backward(ctx, dy):
        ln = ctx.ln
        dx = ops.ln_bwd(dy.contiguous(), ctx.x, ctx.mean, ctx.rstd, ln.gamma, ln.ggamma, ln.gbeta, dx_dtype=ctx.x.dtype)
        return dx, None, None


def layer_norm(x, ln, out_bf16=False):
    return LayerNormFn.apply(x, ln, out_bf16)


class GatherAddFn(torch.autograd.Function):
    """out[r] = sum_k table_k[idx_k[r]]   (f32).  `act` (optional) is an activation tensor source (bf16 or f32,
    differentiable through autograd); tables are (param, grad, idx[, period]) tuples from the store, gradients
    scatter-added (`period`: idx repeats with that period -> the backward reduces over the repeats first)."""

    @staticmethod
    def forward(ctx, act, act_idx, tables, rows, H, anchor, act_inv=None):
        srcs = []
        if act is not None:
            srcs.append((act.contiguous().view(-1, H), act_idx))
        for ent in tables:
            srcs.append((ent[0].view(-1, H), ent[2]))
        while len(srcs) < 4:
            srcs.append((None, None))
        # slot `a` is the only one that may be bf16: put the activation there
        (a, ia), (b, ib), (c, ic), (d, id_) = srcs[:4]
        out = ops.gather_add4(rows, H, a=a, ia=ia, b=b, ib=ib, c=c, ic=ic, d=d, id_=id_)
        ctx.act_shape = None if act is None else act.shape
        ctx.act_dtype = None if act is None else act.dtype
        ctx.act_idx, ctx.tables, ctx.H, ctx.act_inv = act_idx, tables, H, act_inv
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        H = ctx.H
        reds = {}                                            # one reduction per period, shared by the tables that repeat with it
        for ent in ctx.tables:
            tab, gtab, idx = ent[0], ent[1], ent[2]
            period = ent[3] if len(ent) > 3 else None
            if period is not None and dout.shape[0] > period:
                # the index pattern repeats every `period` rows (position tables): reduce over the repeats first, so
                # the scatter issues `period` rows of atomics instead of one per token (fp32 atomics are slow here)
                if period not in reds:
                    reds[period] = dout.view(-1, period, H).sum(0)
                ops.scatter_add_rows(reds[period], idx[:period].contiguous(), gtab.view(-1, H))
            else:
                ops.scatter_add_rows(dout, idx, gtab.view(-1, H))
        dact = None
        if ctx.act_shape is not None and ctx.needs_input_grad[0]:
            if ctx.act_idx is None:
                dact = dout.view(ctx.act_shape).to(ctx.act_dtype)
            elif isinstance(ctx.act_inv, tuple):
                # source row (n, j) sits at output row n * period + skip + j: the gradient is a strided slice, cast in ONE pass
                period, skip = ctx.act_inv
                dact = dout.view(-1, period, H)[:, skip:].to(ctx.act_dtype).reshape(ctx.act_shape)
            elif ctx.act_inv is not None:
                # injective placement (every source row lands in exactly one output row): the gradient is a GATHER
                dact = dout.index_select(0, ctx.act_inv).view(ctx.act_shape).to(ctx.act_dtype)
            else:
                n_src = 1
                for s in ctx.act_shape[:-1]:
                    n_src *= s
                buf = torch.zeros((n_src, H), device=dout.device, dtype=F32)
                ops.scatter_add_rows(dout, ctx.act_idx, buf)
                dact = buf.view(ctx.act_shape).to(ctx.act_dtype)
        return dact, None, None, None, None, None, None


def gather_add(act, act_idx, tables, rows, H, anchor, act_inv=None):
    """`anchor`: the store's dummy requires-grad tensor, so a gather of parameters only is still a graph root.
    `act_inv` (optional): for an injective act_idx, the output row each source row was placed in (int64 tensor), or the tuple
    (period, skip) when source row (n, j) sits at output row n * period + skip + j."""
    return GatherAddFn.apply(act, act_idx, tables, rows, H, anchor, act_inv)


class SplitHiddenFn(torch.autograd.Function):
    """The joint encoder's output [B, S, H] bf16 -> one f32 tensor [B, end - start, H] per piece (model/modeling.py:184: the hidden states
    handed to the heads, cast to f32).  One node instead of a slice + cast per piece: the backward writes every piece's gradient into ITS
    slice of one bf16 buffer (a cast-copy per piece) where autograd's slice nodes would each zero-fill a full [B, S, H] tensor, copy their
    slice in and add the results."""

    @staticmethod
    def forward(ctx, enc3, bounds):
        ctx.shape, ctx.bounds, ctx.dtype = enc3.shape, bounds, enc3.dtype
        return tuple(enc3[:, s:e].float() for s, e in bounds)

    @staticmethod
    def backward(ctx, *douts):
        # (autograd materialises the gradient of an unused piece as zeros; the pieces tile [0, S))
        d = torch.empty(ctx.shape, device=douts[0].device, dtype=ctx.dtype)
        assert sum(e - s for s, e in ctx.bounds) == ctx.shape[1]
        for (s, e), g in zip(ctx.bounds, douts):
            d[:, s:e].copy_(g)
        return d, None


class ClsAvgPoolFn(torch.autograd.Function):
    """[n_img, S, H] bf16 ViT output -> (f32 [n_img, 1 + h2*w2, H] = [cls slot 0 ; 2x2 avg-pooled grid], f32 [n_img, H] = the tokens of
    row `extra_row` (the contrastive head's image representation, model/modeling.py:99)).  Both consumers of the ViT output in ONE node:
    a separate `hs[:, 1].float()` costs the backward a zero-filled [n_img, S, H] tensor and a full-size add to join two rows."""

    @staticmethod
    def forward(ctx, x, n_img, h1, w1, cls_skip, pool, extra_row):
        ctx.args = (n_img, h1, w1, cls_skip, pool, extra_row)
        x = x.contiguous()
        return ops.cls_avgpool_fwd(x, n_img, h1, w1, cls_skip, pool), x.view(n_img, cls_skip + h1 * w1, -1)[:, extra_row].float()

    @staticmethod
    def backward(ctx, dout, dextra):
        n_img, h1, w1, cls_skip, pool, extra_row = ctx.args
        dx = ops.cls_avgpool_bwd(dout.contiguous(), n_img, h1, w1, cls_skip, pool)
        if dextra is not None:
            dx[:, extra_row] += dextra.to(dx.dtype)
        return dx.view(n_img * (cls_skip + h1 * w1), -1), None, None, None, None, None, None


class PatchEmbedFn(torch.autograd.Function):
    """image NHWC bf16 -> [n_img*h1*w1, H] bf16 (utils/vision_transformer.py:193-205): im2col(image - 0.5) + MFMA GEMM;
    the patch matrix is kept for the weight gradient.  The image gets no gradient."""

    @staticmethod
    def forward(ctx, image, lin, patch, anchor):
        out, patches = ops.patch_embed_fwd(image, lin.wb, lin.b, patch)
        ctx.patches, ctx.lin = patches, lin
        return out

    @staticmethod
    def backward(ctx, dy):
        lin = ctx.lin
        dy = dy.contiguous()
        ops.colsum_bf16(dy, lin.gb)
        ops.patch_embed_wgrad(ctx.patches, dy, lin.gw, accumulate=True)
        ctx.patches = None
        return None, None, None, None


class StemWeights(object):
    """Every convolution kernel of the ResNet-hybrid stem, weight-standardised (utils/vision_transformer.py:52-56) in ONE launch per
    forward and back-propagated into the master gradients in ONE launch per backward (`merlot_weight_std_fwd_batched` /
    `_bwd_batched`: 52 kernels of a few KB to a few MB were 52 launches of 18 / 34 us each way).  Per kernel: khat fp32 [K, Co], rstd [Co],
    wb bf16 [Co, Kp] (NT operand), wbT bf16 [Kp, Cop] (1x1 input-gradient operand), for 3x3 kernels wdg bf16 [Cin, 9 Co] (the
    input-gradient operand of the implicit convolution: taps flipped), and dk fp32 [Co (+1), Kp], the slot the layer's weight-gradient
    GEMM writes.  Flat buffers allocated once; paddings are zero from the allocation and never written."""

    def __init__(self, store, scope):
        self.store = store
        names = [n for n, (_, _, shp) in store.offsets.items() if n.startswith(scope + '/') and n.endswith('/kernel') and len(shp) == 4]
        dev = store.device
        fj, bj, self.meta = [], [], {}
        o = dict(khat=0, rstd=0, wb=0, wbT=0, wdg=0, dk=0)
        blocks = 0
        for name in names:
            off, _, (kh, kw, ci, co) = store.offsets[name]
            K = kh * kw * ci
            Kp, Cop, co2 = (K + 63) // 64 * 64, (co + 63) // 64 * 64, co + co % 2
            dg = o['wdg'] if kh == 3 and ci % 8 == 0 else -1
            self.meta[name] = (K, co, Kp, Cop, ci, co2, dict(o), dg)
            fj.append([off, K, co, o['khat'], o['rstd'], o['wb'], Kp, o['wbT'], Cop, dg, ci, blocks])
            bj.append([o['dk'], Kp, o['khat'], o['rstd'], K, co, off, blocks])
            blocks += (co + 15) // 16
            o['khat'] += K * co
            o['rstd'] += (co + 7) // 8 * 8
            o['wb'] += co * Kp
            o['wbT'] += Kp * Cop
            o['dk'] += co2 * Kp
            if dg >= 0:
                o['wdg'] += 9 * ci * co
        self.blocks = blocks
        self.khat = torch.zeros(o['khat'], device=dev, dtype=F32)
        self.rstd = torch.zeros(o['rstd'], device=dev, dtype=F32)
        self.wb = torch.zeros(o['wb'], device=dev, dtype=BF16)
        self.wbT = torch.zeros(o['wbT'], device=dev, dtype=BF16)
        self.wdg = torch.zeros(max(o['wdg'], 8), device=dev, dtype=BF16)
        self.dk = torch.zeros(o['dk'], device=dev, dtype=F32)
        self.fjobs = torch.tensor(fj, dtype=torch.int64).to(dev)
        self.bjobs = torch.tensor(bj, dtype=torch.int64).to(dev)

    def standardise(self):
        ops.weight_std_fwd_batched(self.store.master, self.fjobs, self.blocks, self.khat, self.rstd, self.wb, self.wbT, self.wdg)
        self.generation = getattr(self, 'generation', 0) + 1
        self.std_of_master = self.store.master_version

    def restore_for(self, generation, master_version):
        """ONE set of buffers serves every graph of the stem (tape entries hold views of it).  A second forward between a graph's forward
        and its backward (an evaluation pass inside a training step, two live graphs) re-standardises in place: from the same master
        weights that rewrites identical values, and the older graph's backward only has to make sure of it; after a weight update the
        older graph's standardised kernels are gone, and its backward refuses instead of back-propagating with the wrong ones."""
        if generation == self.generation:
            return
        if master_version != self.store.master_version:
            raise RuntimeError("ResNet-hybrid stem: the master weights changed between this graph's forward and its backward and another "
                               "stem forward has overwritten the shared standardised kernels; run backward before the optimizer step")
        if self.std_of_master != self.store.master_version:
            self.standardise()

    def backward(self):
        """store.grad[kernel] += the standardisation's backward of every dk slot (call once, after the last weight-gradient GEMM)."""
        ops.weight_std_bwd_batched(self.dk, self.bjobs, self.blocks, self.khat, self.rstd, self.store.grad)

    def get(self, name):
        """-> khat [K, Co], wb [Co, Kp], wbT [Kp, Cop], wdg [Cin, 9 Co] or None, dk [Co (+1), Kp]"""
        K, co, Kp, Cop, ci, co2, o, dg = self.meta[name]
        return (self.khat[o['khat']:o['khat'] + K * co].view(K, co), self.wb[o['wb']:o['wb'] + co * Kp].view(co, Kp),
                self.wbT[o['wbT']:o['wbT'] + Kp * Cop].view(Kp, Cop),
                self.wdg[dg:dg + 9 * ci * co].view(ci, 9 * co) if dg >= 0 else None,
                self.dk[o['dk']:o['dk'] + co2 * Kp].view(co2, Kp))


class ResNetStemFn(torch.autograd.Function):
    """ResNet-hybrid stem (utils/vision_transformer.py:114-170, 206-223): image NHWC bf16 ->
    [n_img*h1*w1, H] bf16 tokens, the drop-in for PatchEmbedFn when `resnet_layers` is set.

    Convolutions run on the MFMA GEMMs: 1x1 directly on the [N*H*W, C] activations, 3x3 (stride 1, C % 32 == 0: all but the
    root) as implicit GEMMs -- `merlot_conv3x3_bf16` forward and, with flipped taps, input gradient, `merlot_conv3x3_wgrad_bf16`
    weight gradient; no patch matrix -- or, for the 3-channel stride-2 root and with `model.resnet_implicit_conv: false`, on an
    im2col matrix (`merlot_im2col3x3`, dgrad through `merlot_col2im3x3`); GroupNorm + ReLU (+ the bottleneck's residual add)
    and the 2x2 average pools are the kernels of csrc/conv.hip.  Kernels are weight-standardised (:52-56) from the fp32 masters
    every step -- a few MB of parameters, done with torch ops on the device -- and the standardisation is
    back-propagated into the master gradient.  Activations saved for the backward: conv inputs (or their im2col
    matrices), conv outputs, GroupNorm statistics and outputs.
    """

    @staticmethod
    def forward(ctx, image, store, cfg, anchor):
        vs = 'vision_backbone/vision_transformer'
        rs = f'{vs}/resnet50lite'
        tape = []
        implicit = bool(cfg.get('resnet_implicit_conv', True))

        sw = getattr(store, '_stem_weights', None)
        if sw is None:
            sw = store._stem_weights = StemWeights(store, rs)
        sw.standardise()                                  # every kernel of the stem: one launch
        ctx.stem_generation, ctx.stem_master_version = sw.generation, store.master_version

        def conv(x, name, stride=1, shift=0.0):
            kh, _, _, co = store.offsets[name + '/kernel'][2]
            khat, wb, wbT, wdg, dk = sw.get(name + '/kernel')
            N, Hh, Ww, C = x.shape
            if kh == 3 and implicit and stride == 1 and shift == 0.0 and C % 32 == 0 and co % 32 == 0:
                # implicit GEMM (csrc/conv_gemm.hip): no patch matrix in HBM; the tape keeps x, the backward gathers from it
                y = ops.conv3x3(x, wb, co)
                tape.append(('conv', name, None, khat, (wdg, dk), wbT, (N, Hh, Ww, C), stride, kh, x))
                return y
            if kh == 1:
                a = x.reshape(N * Hh * Ww, C)
                Ho, Wo = Hh, Ww
            else:
                a = ops.im2col3x3(x, stride, shift)
                Ho, Wo = Hh // stride, Ww // stride
            y = ops.gemm_nt(a, wb).view(N, Ho, Wo, co)
            tape.append(('conv', name, a, khat, (wdg, dk), wbT, (N, Hh, Ww, C), stride, kh, None))
            return y

        def gn(x, name, relu=True, res=None):
            g_, b_ = store.p(name + '/gamma'), store.p(name + '/beta')
            y, stats = ops.groupnorm_fwd(x, g_, b_, res=res, relu=relu)
            # the backward's ReLU mask: recomputed from x for a layer without a residual add (no second read of y), from y otherwise
            tape.append(('gn', name, x, y if (relu and res is not None) else None, stats, relu, res is not None))
            return y

        def pool(x):
            tape.append(('pool',))
            return ops.avgpool2_fwd(x)

        x = gn(conv(image, f'{rs}/stem/conv2d', 2, -0.5), f'{rs}/stem/GroupNorm_stem0')     # image - 0.5 (:193) in the gather
        x = gn(conv(x, f'{rs}/stem/conv2d_1'), f'{rs}/stem/GroupNorm_stem1')
        x = gn(conv(x, f'{rs}/stem/conv2d_2'), f'{rs}/stem/GroupNorm_stem2')
        c = pool(x)
        for i, blocks in enumerate(cfg['resnet_layers']):
            gs = f'{rs}/block_group{i + 1}'
            k = [0]

            def nm():
                n_ = '' if k[0] == 0 else f'_{k[0]}'
                k[0] += 1
                return f'{gs}/conv2d{n_}', f'{gs}/GroupNorm{n_}'
            for bi in range(blocks):
                st = (1 if i == 0 else 2) if bi == 0 else 1
                tape.append(('block_begin', bi == 0, st))
                shortcut = c
                if bi == 0:
                    cn, gnn = nm()
                    sc_in = pool(c) if st > 1 else c
                    shortcut = gn(conv(sc_in, cn), gnn, relu=False)
                tape.append(('main_begin',))
                cn, gnn = nm()
                h = gn(conv(c, cn), gnn)
                cn, gnn = nm()
                h = gn(conv(h, cn), gnn)
                if st > 1:
                    h = pool(h)
                cn, gnn = nm()
                c = gn(conv(h, cn), gnn, relu=True, res=shortcut)       # relu(gn(conv) + shortcut), :92-93
                tape.append(('block_end',))
        N, h1, w1, C = c.shape
        lin = store.lin(f'{vs}/conv_postresnet_proj')
        a = c.reshape(N * h1 * w1, C)
        out = ops.gemm_nt(a, lin.wb, bias=lin.b)
        ctx.tape, ctx.store, ctx.lin, ctx.a_final, ctx.c_shape, ctx.sw = tape, store, lin, a, (N, h1, w1, C), sw
        return out

    @staticmethod
    def backward(ctx, dy):
        store, tape, lin = ctx.store, ctx.tape, ctx.lin
        store._stem_weights.restore_for(ctx.stem_generation, ctx.stem_master_version)
        dy = dy.contiguous()
        ops.colsum_bf16(dy, lin.gb)
        ops.gemm_tn(dy, ctx.a_final, lin.gw)
        N, h1, w1, C = ctx.c_shape
        d = ops.gemm_nt(dy, lin.wbT).view(N, h1, w1, C)

        def conv_bwd(entry, dyc, need_dx=True, add=None):
            """`add` (1x1 convolutions only): a gradient of the convolution's INPUT arriving on another path (the bottleneck block's
            shortcut), added in the input-gradient GEMM's residual epilogue instead of by a separate pass over the tensor."""
            _, name, a, khat, (wdg, dk), wbT, xshape, stride, kh, x_in = entry
            Nn, Hh, Ww, Cc = xshape
            co = khat.shape[1]
            dyf = dyc.reshape(-1, co)
            # dKhat^T [Co, Kp] into this kernel's slot of the stem's dk buffer; the weight standardisation's backward (khat = (k - mean) *
            # rstd per output channel) runs once for all kernels at the end of the stem's backward (StemWeights.backward)
            if x_in is not None:                                                        # implicit 3x3: gathered from x, no patch matrix
                ops.conv3x3_wgrad(dyf, x_in, dk)
            else:
                ops.gemm_tn(dyf, a, dk, accumulate=False, m=co + (co % 2))
            if not need_dx:
                return None
            if x_in is not None:
                # dX = the same convolution of dY with the taps flipped: wdg[ci][(ky', kx', co)] = khat[(2-ky', 2-kx', ci), co]
                # -- one fp32 accumulation over the nine taps, no [T, 9 C] product in HBM and no col2im pass over it
                out = ops.conv3x3(dyc.reshape(Nn, Hh, Ww, co).contiguous(), wdg, Cc)
                return out if add is None else out + add
            kp_co = wbT.shape[1]
            if kp_co != co:                                                             # pad the reduction dim to 64
                dyp = torch.zeros((dyf.shape[0], kp_co), device=dyf.device, dtype=BF16)
                dyp[:, :co] = dyf
                dyf = dyp
            if add is not None and kh == 1 and wbT.shape[0] == Cc:
                dp = ops.gemm_nt(dyf, wbT, epilogue=EPI_RESIDUAL, aux_in=add.reshape(-1, Cc))   # [T, C] + the other path's gradient
                return dp.view(Nn, Hh, Ww, Cc)
            dp = ops.gemm_nt(dyf, wbT)                                                  # [T, Kp]
            if kh == 1:
                out = dp[:, :Cc].reshape(Nn, Hh, Ww, Cc) if dp.shape[1] != Cc else dp.view(Nn, Hh, Ww, Cc)
            else:
                out = ops.col2im3x3(dp, Nn, Hh, Ww, Cc, stride)
            return out if add is None else out + add

        def gn_bwd(entry, dyg):
            _, name, x, y, stats, relu, has_res = entry
            dx, dres = ops.groupnorm_bwd(dyg.contiguous(), y, x, stats, store.p(name + '/gamma'), store.g(name + '/gamma'),
                                         store.g(name + '/beta'), beta=store.p(name + '/beta'), relu=relu, want_dres=has_res)
            return dx, dres

        # walk the tape backwards; the bottleneck blocks need their two branches' gradients joined
        i = len(tape) - 1

        def pop():
            nonlocal i
            e = tape[i]
            i -= 1
            return e

        while i >= 0 and tape[i][0] == 'block_end':
            pop()
            # main branch, last to first: gn(res) <- conv <- [pool] <- gn <- conv <- gn <- conv
            e = pop(); d_h, d_short = gn_bwd(e, d)
            d_h = conv_bwd(pop(), d_h)
            if tape[i][0] == 'pool':
                pop(); d_h = ops.avgpool2_bwd(d_h.contiguous())
            e = pop(); d_h, _ = gn_bwd(e, d_h)
            d_h = conv_bwd(pop(), d_h)
            e = pop(); d_h, _ = gn_bwd(e, d_h)
            conv1 = pop()                                                                # its input gradient joins the shortcut's below
            assert pop()[0] == 'main_begin'
            # shortcut branch
            if tape[i][0] == 'gn':                                                       # projection shortcut
                e = pop(); d_s, _ = gn_bwd(e, d_short)
                d_s = conv_bwd(pop(), d_s)
                if tape[i][0] == 'pool':
                    pop(); d_s = ops.avgpool2_bwd(d_s.contiguous())
            else:
                d_s = d_short
            assert pop()[0] == 'block_begin'
            d = conv_bwd(conv1, d_h, add=d_s.contiguous())       # d(block input) = conv1's input gradient + the shortcut's, one GEMM
        assert pop()[0] == 'pool'
        d = ops.avgpool2_bwd(d.contiguous())
        for j in range(3):                                                               # the three stem convolutions
            e = pop(); d, _ = gn_bwd(e, d)
            d = conv_bwd(pop(), d, need_dx=(j < 2))
        assert i == -1
        ctx.sw.backward()                                # every kernel's gradient through the standardisation: one launch
        store.notify_ready('vision_backbone/vision_transformer/resnet50lite')
        ctx.tape = None
        return None, None, None, None


class L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, inv = ops.l2norm_fwd(x.contiguous())
        ctx.save_for_backward(y, inv)                    # y is this node's OUTPUT: as a plain ctx attribute it is a reference cycle
        return y                                         # (y -> grad_fn -> ctx -> y) that keeps the whole upstream graph alive until a gc pass

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return ops.l2norm_bwd(dy, y, inv)


class SoftmaxCEFn(torch.autograd.Function):
    """per-row cross entropy of f32 logits [rows, >=C]; returns (loss[rows], argmax[rows])."""

    @staticmethod
    def forward(ctx, logits, labels, C):
        loss, am, dl = ops.softmax_ce(logits, labels, C, dlogits_dtype=F32, ld_dl=logits.shape[1])
        ctx.dl = dl
        ctx.mark_non_differentiable(am)
        return loss, am

    @staticmethod
    def backward(ctx, dloss, _dam):
        return ctx.dl * dloss[:, None], None, None


class VocabCEFn(torch.autograd.Function):
    """Masked-LM head tail (model/modeling.py:217-223, 539-545): logits = h @ E^T + output_bias, CE vs targets,
    weighted row sum.  h: [T, H] f32 ; emb: Lin handle of the tied word-embedding table ; rowweight f32 [T].
    Returns (sum_r rowweight[r] * ce[r]  (differentiable scalar), ce[T], argmax[T])."""

    @staticmethod
    def forward(ctx, h, emb, out_bias, gout_bias, targets, rowweight, vocab):
        hb = ops.cast_bf16(h.contiguous())
        vpad = emb.wbT.shape[1]
        # logits GEMM + softmax cross-entropy as ONE C-ABI call (merlot_vocab_ce_fwd) over a caller-owned fp32 scratch
        loss, am, dl = ops.vocab_ce(hb, emb.wb, out_bias, targets, rowweight, vocab, vpad)
        ctx.hb, ctx.emb, ctx.dl, ctx.gout_bias, ctx.vocab = hb, emb, dl, gout_bias, vocab
        ctx.mark_non_differentiable(loss, am)
        return (loss * rowweight).sum(), loss, am

    @staticmethod
    def backward(ctx, g, _dl, _dam):
        emb, vocab, dl = ctx.emb, ctx.vocab, ctx.dl          # dl already carries rowweight; g is the scalar upstream
        if ctx.gout_bias is not None:
            tmp = torch.zeros_like(ctx.gout_bias)
            ops.colsum_bf16(dl, tmp, accumulate=False, n=vocab)
            ctx.gout_bias.add_(tmp * g)
        hg = (ctx.hb.float() * g).to(BF16)
        ops.gemm_tn(dl, hg, emb.gw, m=vocab)                             # dE[V, H] += dlogits^T h
        dh = ops.gemm_nt(dl, emb.wbT, out_dtype=F32) * g                 # [T, H]
        ctx.dl = None
        return dh, None, None, None, None, None, None
