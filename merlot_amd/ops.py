"""Tensor-level wrappers over the C-ABI (raw pointers + the current HIP stream).  No autograd here.

PyTorch is plumbing only: it owns device memory and the stream; every op below is one or more
hand-written gfx950 kernels in libmerlot_hip.so.  There is no eager / CPU fallback.
"""
import torch

from .lib import call, LIB

EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU = 0, 1, 2, 3
BF16 = torch.bfloat16
F32 = torch.float32


class KernelTimer(object):
    """Optional per-kernel HIP-event timing (bench.py roofline): set ops.TIMER = KernelTimer() and every gemm_nt
    launch is bracketed by events on the launch stream; .summary() -> (flops, seconds, launches) per key."""

    def __init__(self):
        self.records = []

    def time(self, key, flops, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.records.append((key, flops, e0, e1))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, flops, e0, e1 in self.records:
            f, t, n = out.get(key, (0.0, 0.0, 0))
            out[key] = (f + flops, t + e0.elapsed_time(e1) * 1e-3, n + 1)
        return out


TIMER = None


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_NT_WS = {}


def _nt_ws():
    """(pointer, bytes) of the tile-claim counter block the persistent NT GEMMs need (include/merlot_hip.h: caller-owned, zero
    on entry, left zero): one block per (device, stream) -- launches of one stream run in order, so they share it."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    buf = _NT_WS.get(key)
    if buf is None:
        buf = _NT_WS[key] = torch.zeros(LIB.query('merlot_gemm_nt_workspace_bytes') // 4, device='cuda', dtype=torch.int32)
    return buf.data_ptr(), buf.numel() * 4


_ATTN_WS = {}


def _attn_ws():
    """(pointer, bytes) of the item-claim counter block of the persistent attention kernels (include/merlot_hip.h, ABI v7: caller-owned, zero on
    entry, left zero): one block per (device, stream), separate from the GEMMs'."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    buf = _ATTN_WS.get(key)
    if buf is None:
        buf = _ATTN_WS[key] = torch.zeros(LIB.query('merlot_attention_workspace_bytes') // 4, device='cuda', dtype=torch.int32)
    return buf.data_ptr(), buf.numel() * 4


def reset_workspaces():
    """Zero every claim-counter block handed out so far.  The persistent kernels' contract is "zero on entry, left zero" and they
    carry no launch epoch: a launch that failed or was aborted half-way may leave a counter non-zero, and the NEXT launch on that block would then
    skip its first items silently.  Registered as the binding's error hook (every failed C-ABI call runs it before raising) and callable by hand
    after anything that may have killed a kernel.  One block belongs to ONE stream: concurrent launches sharing a block are undefined behaviour
    (include/merlot_hip.h), which is why the blocks are keyed by (device, stream) here."""
    for buf in list(_NT_WS.values()) + list(_ATTN_WS.values()) + list(_LN_WS.values()):
        buf.zero_()


LIB.on_error.append(reset_workspaces)


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: last dim must be contiguous")


def gemm_nt(a, bt, *, bias=None, epilogue=EPI_NONE, out=None, out_dtype=BF16, accumulate=False, alpha=1.0,
            aux_in=None, aux_out=None, dropout_p=0.0, dropout_seed=0, n=None, colsum_out=None):
    """C[M,N] = epi(alpha * a[M,K] @ bt[N,K]^T).  a, bt bf16 2-D (row stride = leading dim).
    colsum_out (f32 [N], accumulated): column sums of the stored bf16 C from the same launch (a bias gradient)."""
    _chk(a, BF16, 'a'); _chk(bt, BF16, 'bt'); _chk(bias, F32, 'bias'); _chk(aux_in, BF16, 'aux_in'); _chk(aux_out, BF16, 'aux_out')
    _chk(colsum_out, F32, 'colsum_out')
    M, K = a.shape
    N = bt.shape[0] if n is None else n
    if bt.shape[1] != K:
        raise ValueError(f"gemm_nt: K mismatch {a.shape} vs {bt.shape}")
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=out_dtype)
    _chk(out, out.dtype, 'out')

    def launch():
        call('merlot_gemm_bf16_nt', _p(a), a.stride(0), _p(bt), bt.stride(0), _p(out), out.stride(0), M, N, K,
             float(alpha), int(epilogue), 1 if out.dtype == F32 else 0, 1 if accumulate else 0, _p(bias), _p(aux_in),
             aux_in.stride(0) if aux_in is not None else 0, _p(aux_out), aux_out.stride(0) if aux_out is not None else 0,
             float(dropout_p), int(dropout_seed) & 0xFFFFFFFFFFFFFFFF, _p(colsum_out), *_nt_ws(), _stream())

    if TIMER is not None:
        TIMER.time('gemm_nt', 2.0 * M * N * K, launch)
    else:
        launch()
    return out


_LN_WS = {}


def _ln_ws(M, N):
    """(pointer, bytes) of the LayerNorm-fold workspace (include/merlot_hip.h, ABI v8: arrival counters zero on entry and left zero + scratch for the
    segment statistics)."""
    # one block per (device, stream, M): the block's layout depends on M (ceil(M / 256) arrival counters, then the scratch), so a block that served another M
    # holds scratch where this launch expects zeroed counters (the first model-level run did exactly that: LayerNorm passes that never ran, NaN activations)
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, int(M), int(N))
    buf = _LN_WS.get(key)
    if buf is None:
        need = LIB.query('merlot_gemm_nt_ln_workspace_bytes', M, N)
        buf = _LN_WS[key] = torch.zeros((need + 3) // 4, device='cuda', dtype=torch.int32)
    return buf.data_ptr(), buf.numel() * 4


def gemm_nt_ln(a, bt, gamma, beta, *, bias=None, aux_in=None, dropout_p=0.0, dropout_seed=0, alpha=1.0, eps=1e-5, save_stats=True):
    """h = aux_in + dropout(alpha * a @ bt^T + bias) and LayerNorm(h) from ONE call (merlot_gemm_bf16_nt_ln) -> (h, y, mean, rstd)."""
    _chk(a, BF16, 'a'); _chk(bt, BF16, 'bt'); _chk(bias, F32, 'bias'); _chk(aux_in, BF16, 'aux_in'); _chk(gamma, F32, 'gamma'); _chk(beta, F32, 'beta')
    M, K = a.shape
    N = bt.shape[0]
    h = torch.empty((M, N), device=a.device, dtype=BF16)
    y = torch.empty((M, N), device=a.device, dtype=BF16)
    mean = torch.empty(M, device=a.device, dtype=F32) if save_stats else None
    rstd = torch.empty(M, device=a.device, dtype=F32) if save_stats else None

    def launch():
        call('merlot_gemm_bf16_nt_ln', _p(a), a.stride(0), _p(bt), bt.stride(0), _p(h), N, M, N, K, float(alpha), _p(bias), _p(aux_in),
             aux_in.stride(0), float(dropout_p), int(dropout_seed) & 0xFFFFFFFFFFFFFFFF, _p(gamma), _p(beta), _p(y), N, _p(mean), _p(rstd),
             float(eps), *_ln_ws(M, N), *_nt_ws(), _stream())

    if TIMER is not None:
        TIMER.time('gemm_nt', 2.0 * M * N * K, launch)
    else:
        launch()
    return h, y, mean, rstd


FP8 = torch.float8_e4m3fn


def quantize_e4m3(x, out=None):
    """x bf16 [rows, cols] -> (y e4m3 [rows, cols], scale f32[3] = {s, 1/s, max|x|}), s = 448 / max|x| over the whole tensor."""
    _chk(x, BF16, 'x')
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), device=x.device, dtype=FP8)
    _chk(out, FP8, 'out')
    scale = torch.empty(3, device=x.device, dtype=F32)
    call('merlot_quantize_e4m3', _p(x), rows, cols, x.stride(0), _p(out), out.stride(0), _p(scale), _stream())
    return out, scale


def gemm_fp8_nt(a8, a_scale, bt8, b_scale, *, bias=None, epilogue=EPI_NONE, out=None, out_dtype=BF16, alpha=1.0,
                aux_in=None, aux_out=None, dropout_p=0.0, dropout_seed=0, a_row_scale=None):
    """C[M,N] = epi(alpha / (sa * sb) * a8[M,K] @ bt8[N,K]^T): gemm_nt on e4m3 operands; a_scale / b_scale are the f32[3]
    tensors quantize_e4m3 returned (their [1] entry, the dequantisation factor, is read on the device)."""
    _chk(a8, FP8, 'a8'); _chk(bt8, FP8, 'bt8'); _chk(a_scale, F32, 'a_scale'); _chk(b_scale, F32, 'b_scale')
    _chk(a_row_scale, F32, 'a_row_scale')
    if a_scale is None and a_row_scale is None:
        raise ValueError("gemm_fp8_nt: a8 needs its per-tensor scale or per-row scales")
    if a_row_scale is not None and a_row_scale.numel() != a8.shape[0]:
        raise ValueError("gemm_fp8_nt: a_row_scale must have one entry per row of a8")
    _chk(bias, F32, 'bias'); _chk(aux_in, BF16, 'aux_in'); _chk(aux_out, BF16, 'aux_out')
    M, K = a8.shape
    N = bt8.shape[0]
    if bt8.shape[1] != K:
        raise ValueError(f"gemm_fp8_nt: K mismatch {a8.shape} vs {bt8.shape}")
    if out is None:
        out = torch.empty((M, N), device=a8.device, dtype=out_dtype)
    _chk(out, out.dtype, 'out')

    def launch():
        call('merlot_gemm_fp8_nt', _p(a8), a8.stride(0), a_scale.data_ptr() + 4 if a_scale is not None else None, _p(a_row_scale),
             _p(bt8), bt8.stride(0), b_scale.data_ptr() + 4,
             _p(out), out.stride(0), M, N, K, float(alpha), int(epilogue), 1 if out.dtype == F32 else 0, _p(bias), _p(aux_in),
             aux_in.stride(0) if aux_in is not None else 0, _p(aux_out), aux_out.stride(0) if aux_out is not None else 0,
             float(dropout_p), int(dropout_seed) & 0xFFFFFFFFFFFFFFFF, *_nt_ws(), _stream())

    if TIMER is not None:
        TIMER.time('gemm_fp8_nt', 2.0 * M * N * K, launch)
    else:
        launch()
    return out


def gemm_tn(a, b, out, *, accumulate=True, alpha=1.0, m=None, n=None, colsum_a=None):
    """out[M,N] (f32) (+)= alpha * a[R,M]^T @ b[R,N].  colsum_a (f32 [m'], accumulated): += the column sums of a[:, :m'] from the same launch."""
    _chk(a, BF16, 'a'); _chk(b, BF16, 'b'); _chk(out, F32, 'out'); _chk(colsum_a, F32, 'colsum_a')
    R = a.shape[0]
    M = a.shape[1] if m is None else m
    N = b.shape[1] if n is None else n
    nbytes = LIB.query('merlot_gemm_bf16_tn_workspace_bytes', M, N, R)
    ws = torch.empty(nbytes // 4, device=a.device, dtype=F32) if nbytes else None   # caller-owned split-R partials

    def launch():
        call('merlot_gemm_bf16_tn_cs', _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, R,
             float(alpha), 1 if accumulate else 0, _p(colsum_a), colsum_a.numel() if colsum_a is not None else 0, _p(ws), nbytes, _stream())

    if TIMER is not None:
        TIMER.time('gemm_tn', 2.0 * M * N * R, launch)
    else:
        launch()
    return out


F8_E4M3, F8_E5M2 = 0, 1
_F8_DTYPES = {F8_E4M3: torch.float8_e4m3fn, F8_E5M2: torch.float8_e5m2}


def quantize_f8(x, fmt=F8_E4M3, *, scale=None, out=None, row_multiple=128, current_into=None):
    """x bf16 [rows, cols] -> (y f8 [rows padded to row_multiple, cols], scale f32[4] = {s, 1/s, amax s came from, amax of x}).  Per-tensor scaling;
    scale=None: "current" (s from this tensor, two passes; current_into: write the block there instead of a fresh tensor); scale = the block a previous
    call returned: "delayed" (one pass, s from the amax that call recorded, this tensor's amax recorded for the next).  Padding rows are zeros
    (merlot_gemm_f8_tn's K-tile is 128 reduction rows)."""
    _chk(x, BF16, 'x')
    rows, cols = x.shape
    rows_pad = (rows + row_multiple - 1) // row_multiple * row_multiple
    dt = _F8_DTYPES[fmt]
    if out is None:
        out = torch.empty((rows_pad, cols), device=x.device, dtype=dt)
    _chk(out, dt, 'out')
    if out.shape[0] < rows_pad or out.shape[1] != cols:
        raise ValueError(f"quantize_f8: out {tuple(out.shape)} does not hold [{rows_pad}, {cols}]")
    delayed = scale is not None
    if scale is None:
        scale = current_into if current_into is not None else torch.empty(4, device=x.device, dtype=F32)
    _chk(scale, F32, 'scale')
    call('merlot_quantize_f8', _p(x), rows, cols, x.stride(0), _p(out), out.stride(0), rows_pad, int(fmt), 1 if delayed else 0, _p(scale), _stream())
    return out, scale


def gemm_f8_nt(a8, a_scale, bt8, b_scale, *, bias=None, alpha=1.0):
    """C[M,N] bf16 = alpha / (sa * sb) * a8[M,K] @ bt8[N,K]^T (+ bias): a8 e4m3 or e5m2 (by dtype), bt8 e4m3; scales: the blocks quantize_f8 / F8Scales hold."""
    fa = 0 if a8.dtype == torch.float8_e4m3fn else 1
    _chk(a8, _F8_DTYPES[fa], 'a8'); _chk(bt8, FP8, 'bt8'); _chk(a_scale, F32, 'a_scale'); _chk(b_scale, F32, 'b_scale'); _chk(bias, F32, 'bias')
    M, K = a8.shape
    N = bt8.shape[0]
    out = torch.empty((M, N), device=a8.device, dtype=BF16)

    def launch():
        call('merlot_gemm_f8_nt', _p(a8), a8.stride(0), fa, a_scale.data_ptr() + 4, _p(bt8), bt8.stride(0), b_scale.data_ptr() + 4, _p(out), N, M, N, K,
             float(alpha), _p(bias), *_nt_ws(), _stream())

    if TIMER is not None:
        TIMER.time('gemm_fp8_nt', 2.0 * M * N * K, launch)
    else:
        launch()
    return out


def _f8_rows(rows, cols, dtype, device, row_multiple=128):
    """an 8-bit tensor for `rows` rows whose allocation is padded to a multiple of 128 rows (merlot_gemm_f8_tn's K-tile), the padding zeroed: producers
    write rows [0, rows), the weight gradient reads all of it"""
    rp = (rows + row_multiple - 1) // row_multiple * row_multiple
    t = torch.empty((rp, cols), device=device, dtype=dtype)
    if rp > rows:
        t[rows:].zero_()
    return t


def f8_scale_rotate(blocks, n, fmts):
    """blocks f32 [>= n, 4], fmts int32 [>= n]: every block whose producers recorded an amax ([3] > 0) gets {s, 1/s, amax} from it, the record is cleared."""
    _chk(blocks, F32, 'blocks'); _chk(fmts, torch.int32, 'fmts')
    call('merlot_f8_scale_rotate', _p(blocks), int(n), _p(fmts), _stream())


def ln_fwd_q8t(x, gamma, beta, block, *, out_bf16=False, eps=1e-5):
    """ln_fwd that also emits the e4m3 copy of its bf16-rounded output scaled by block[0] (per tensor, delayed) and records max|y| in block[3]:
    -> (y16 or None, y8, mean, rstd)."""
    assert x.dtype in (BF16, F32) and x.is_contiguous()
    _chk(gamma, F32, 'gamma'); _chk(beta, F32, 'beta'); _chk(block, F32, 'block')
    H = x.shape[-1]
    rows = x.numel() // H
    y16 = torch.empty(x.shape, device=x.device, dtype=BF16) if out_bf16 else None
    y8 = _f8_rows(rows, H, FP8, x.device)                # (rows padded to 128, padding zero)
    mean = torch.empty(rows, device=x.device, dtype=F32)
    rstd = torch.empty(rows, device=x.device, dtype=F32)
    call('merlot_ln_fwd_q8t', _p(x), 1 if x.dtype == F32 else 0, _p(gamma), _p(beta), _p(y16), _p(y8), _p(block), _p(mean), _p(rstd),
         rows, H, float(eps), _stream())
    return y16, y8, mean, rstd


def gemm_nt_q8(a, bt, block, fmt, *, epilogue, aux_in, bias=None, colsum_out=None, alpha=1.0, keep_bf16=True):
    """gemm_nt (DGELU epilogue) that also writes the 8-bit float copy of its bf16-rounded output, scaled by block[0]; max|C| -> block[3].
    -> (C bf16 or None, C8)."""
    _chk(a, BF16, 'a'); _chk(bt, BF16, 'bt'); _chk(aux_in, BF16, 'aux_in'); _chk(bias, F32, 'bias'); _chk(colsum_out, F32, 'colsum_out'); _chk(block, F32, 'block')
    M, K = a.shape
    N = bt.shape[0]
    out = torch.empty((M, N), device=a.device, dtype=BF16) if keep_bf16 else None
    out8 = _f8_rows(M, N, _F8_DTYPES[fmt], a.device)

    def launch():
        call('merlot_gemm_bf16_nt_q8', _p(a), a.stride(0), _p(bt), bt.stride(0), _p(out), N, M, N, K, float(alpha), int(epilogue), _p(bias),
             _p(aux_in), aux_in.stride(0), _p(colsum_out), _p(out8), N, int(fmt), _p(block), *_nt_ws(), _stream())

    if TIMER is not None:
        TIMER.time('gemm_nt', 2.0 * M * N * K, launch)
    else:
        launch()
    return out, out8


def gemm_fp8_nt_q8(a8, a_scale, bt8, b_scale, block, *, bias=None, aux_out, a_row_scale=None, alpha=1.0, keep_bf16=True):
    """gemm_fp8_nt (GELU epilogue, pre-activation to aux_out) that also writes the e4m3 copy of its bf16-rounded output, scaled by block[0];
    max|C| -> block[3].  -> (C bf16 or None, C8)."""
    _chk(a8, FP8, 'a8'); _chk(bt8, FP8, 'bt8'); _chk(aux_out, BF16, 'aux_out'); _chk(bias, F32, 'bias'); _chk(block, F32, 'block')
    M, K = a8.shape
    N = bt8.shape[0]
    out = torch.empty((M, N), device=a8.device, dtype=BF16) if keep_bf16 else None
    out8 = _f8_rows(M, N, FP8, a8.device)

    def launch():
        call('merlot_gemm_fp8_nt_q8', _p(a8), a8.stride(0), a_scale.data_ptr() + 4 if a_scale is not None else None, _p(a_row_scale),
             _p(bt8), bt8.stride(0), b_scale.data_ptr() + 4, _p(out), N, M, N, K, float(alpha), int(EPI_GELU), _p(bias),
             _p(aux_out), aux_out.stride(0), _p(out8), N, 0, _p(block), *_nt_ws(), _stream())

    if TIMER is not None:
        TIMER.time('gemm_fp8_nt', 2.0 * M * N * K, launch)
    else:
        launch()
    return out, out8


def gemm_f8_tn(a8, a_scale, b8, b_scale, out, *, accumulate=True, alpha=1.0, m=None, n=None):
    """out[M,N] (f32) (+)= alpha / (sa * sb) * a8[R,M]^T @ b8[R,N] on 8-bit float operands (e4m3 / e5m2 by dtype), R % 128 == 0; a_scale / b_scale: the
    blocks quantize_f8 returned (their [1] entry is read on the device)."""
    fa = 0 if a8.dtype == torch.float8_e4m3fn else 1
    fb = 0 if b8.dtype == torch.float8_e4m3fn else 1
    _chk(a8, _F8_DTYPES[fa], 'a8'); _chk(b8, _F8_DTYPES[fb], 'b8'); _chk(out, F32, 'out'); _chk(a_scale, F32, 'a_scale'); _chk(b_scale, F32, 'b_scale')
    R = a8.shape[0]
    if b8.shape[0] != R:
        raise ValueError(f"gemm_f8_tn: reduction rows differ {tuple(a8.shape)} vs {tuple(b8.shape)}")
    M = a8.shape[1] if m is None else m
    N = b8.shape[1] if n is None else n
    nbytes = LIB.query('merlot_gemm_f8_tn_workspace_bytes', M, N, R)
    ws = torch.empty(nbytes // 4, device=a8.device, dtype=F32) if nbytes else None   # caller-owned split-R partials

    def launch():
        call('merlot_gemm_f8_tn', _p(a8), a8.stride(0), fa, a_scale.data_ptr() + 4, _p(b8), b8.stride(0), fb, b_scale.data_ptr() + 4,
             _p(out), out.stride(0), M, N, R, float(alpha), 1 if accumulate else 0, _p(ws), nbytes, _stream())

    if TIMER is not None:
        TIMER.time('gemm_f8_tn', 2.0 * M * N * R, launch)
    else:
        launch()
    return out


def patch_embed_fwd(image, wt, bias, patch):
    """-> (out [rows, hidden] bf16, patches [rows, P*P*3] bf16 = im2col(image - 0.5), kept for the weight gradient)."""
    _chk(image, BF16, 'image'); _chk(wt, BF16, 'wt'); _chk(bias, F32, 'bias')
    n, H, W, c = image.shape
    assert c == 3 and image.is_contiguous()
    hidden = wt.shape[0]
    rows = n * (H // patch) * (W // patch)
    patches = torch.empty((rows, patch * patch * 3), device=image.device, dtype=BF16)
    out = torch.empty((rows, hidden), device=image.device, dtype=BF16)
    call('merlot_patch_embed_fwd', _p(image), n, H, W, patch, _p(wt), _p(bias), _p(patches), _p(out), hidden, *_nt_ws(), _stream())
    return out, patches


def patch_embed_wgrad(patches, dy, dwt, accumulate=True):
    _chk(patches, BF16, 'patches'); _chk(dy, BF16, 'dy'); _chk(dwt, F32, 'dwt')
    assert dy.is_contiguous() and dwt.is_contiguous() and patches.is_contiguous()
    rows, K = patches.shape
    hidden = dwt.shape[0]
    nbytes = LIB.query('merlot_gemm_bf16_tn_workspace_bytes', hidden, K, rows)
    ws = torch.empty(nbytes // 4, device=dy.device, dtype=F32) if nbytes else None
    call('merlot_patch_embed_wgrad', _p(patches), rows, K, _p(dy), _p(dwt), hidden, 1 if accumulate else 0, _p(ws), nbytes,
         _stream())


def ln_fwd(x, gamma, beta, *, out_bf16=True, out_f32=False, save_stats=True, eps=1e-5):
    assert x.dtype in (BF16, F32) and x.is_contiguous()
    _chk(gamma, F32, 'gamma'); _chk(beta, F32, 'beta')
    H = x.shape[-1]
    rows = x.numel() // H
    y16 = torch.empty(x.shape, device=x.device, dtype=BF16) if out_bf16 else None
    y32 = torch.empty(x.shape, device=x.device, dtype=F32) if out_f32 else None
    mean = torch.empty(rows, device=x.device, dtype=F32) if save_stats else None
    rstd = torch.empty(rows, device=x.device, dtype=F32) if save_stats else None
    call('merlot_ln_fwd', _p(x), 1 if x.dtype == F32 else 0, _p(gamma), _p(beta), _p(y16), _p(y32), _p(mean), _p(rstd),
         rows, H, float(eps), _stream())
    return y16, y32, mean, rstd


def ln_fwd_q8(x, gamma, beta, *, out_bf16=True, save_stats=True, eps=1e-5):
    """ln_fwd that also emits the per-row e4m3 copy of its (bf16-rounded) output: -> (y16, y8, row_scale f32[rows], mean, rstd)."""
    assert x.dtype in (BF16, F32) and x.is_contiguous()
    _chk(gamma, F32, 'gamma'); _chk(beta, F32, 'beta')
    H = x.shape[-1]
    rows = x.numel() // H
    y16 = torch.empty(x.shape, device=x.device, dtype=BF16) if out_bf16 else None
    y8 = torch.empty(x.shape, device=x.device, dtype=FP8)
    rs = torch.empty(rows, device=x.device, dtype=F32)
    mean = torch.empty(rows, device=x.device, dtype=F32) if save_stats else None
    rstd = torch.empty(rows, device=x.device, dtype=F32) if save_stats else None
    call('merlot_ln_fwd_q8', _p(x), 1 if x.dtype == F32 else 0, _p(gamma), _p(beta), _p(y16), _p(y8), _p(rs), _p(mean), _p(rstd),
         rows, H, float(eps), _stream())
    return y16, y8, rs, mean, rstd


def ln_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, *, dres=None, dx_dtype=None, branch_bias_grad=None, drop_p=0.0,
           drop_seed=0, db8_block=None, db8_fmt=F8_E5M2):
    """dx = LN'(dy) (+dres); dgamma/dbeta accumulated in place.  With `branch_bias_grad` (f32 [H]) the kernel also
    accumulates the column sums of d_branch = dropout'(dx) into it and returns (dx, d_branch) (d_branch is dx when
    drop_p == 0).  db8_block (f32[4], with branch_bias_grad): also the 8-bit float copy of d_branch, scaled by block[0], max|d_branch| -> block[3];
    returns (dx, d_branch, d_branch8)."""
    assert dy.is_contiguous() and x.is_contiguous()
    H = x.shape[-1]
    rows = x.numel() // H
    dx_dtype = dx_dtype or x.dtype
    dx = torch.empty(x.shape, device=x.device, dtype=dx_dtype)
    dx_drop = None
    if branch_bias_grad is not None and drop_p > 0:
        dx_drop = torch.empty(x.shape, device=x.device, dtype=BF16)
    args = (_p(dy), 1 if dy.dtype == F32 else 0, _p(x), 1 if x.dtype == F32 else 0, _p(mean), _p(rstd),
            _p(gamma), _p(dres), 1 if (dres is not None and dres.dtype == F32) else 0, _p(dx), 1 if dx_dtype == F32 else 0,
            _p(dgamma), _p(dbeta), rows, H, _p(dx_drop), float(drop_p), int(drop_seed) & 0xFFFFFFFFFFFFFFFF,
            _p(branch_bias_grad))
    if db8_block is not None:
        _chk(db8_block, F32, 'db8_block')
        db8 = _f8_rows(rows, H, _F8_DTYPES[db8_fmt], x.device)
        call('merlot_ln_bwd_q8', *args, _p(db8), int(db8_fmt), _p(db8_block), _stream())
        return dx, (dx_drop if dx_drop is not None else dx), db8
    call('merlot_ln_bwd', *args, _stream())
    if branch_bias_grad is None:
        return dx
    return dx, (dx_drop if dx_drop is not None else dx)


def attention_fwd(qkv, B, S, heads, valid=None, need_lse=True, seg=None, colsum_lo=None, colsum_hi=None, qsplit=None,
                  valid_q_only=False, weight=1.0):
    """colsum_lo / colsum_hi (f32 [B,S], accumulated in place): the attention_colsum side outputs from the same launch."""
    _chk(qkv, BF16, 'qkv'); _chk(seg, torch.int32, 'seg'); _chk(colsum_lo, F32, 'colsum_lo'); _chk(colsum_hi, F32, 'colsum_hi')
    out = torch.empty((B * S, heads * 64), device=qkv.device, dtype=BF16)
    lse = torch.empty((B, heads, S), device=qkv.device, dtype=F32) if need_lse else None
    call('merlot_attention_fwd', _p(qkv), qkv.stride(0), _p(out), out.stride(0), _p(lse), _p(valid), _p(seg), B, S,
         heads, 0.125, _p(colsum_lo), _p(colsum_hi), S if qsplit is None else qsplit, 1 if valid_q_only else 0, float(weight),
         *_attn_ws(), _stream())
    return out, lse


def amax_groups(x, groups):
    """max|x| per equal column group of a bf16 [rows, cols] tensor, one pass -> f32[groups] on the device."""
    _chk(x, BF16, 'x')
    rows, cols = x.shape
    out = torch.empty(groups, device=x.device, dtype=F32)
    call('merlot_amax_bf16', _p(x), rows, cols, x.stride(0), int(groups), _p(out), _stream())
    return out


def attention_bwd_writes_q8(S, has_seg):
    """whether merlot_attention_bwd_q8 exists for this shape (the tiled dQ / dK dV kernel pair: > 512 tokens, a segment mask, or <= 64 tokens)"""
    return bool(LIB.query('merlot_attention_bwd_writes_q8', int(S), 1 if has_seg else 0))


def attention_bwd(qkv, out, dout, lse, B, S, heads, valid=None, seg=None, log_lo=None, log_hi=None, log_split=None, log_weight=1.0,
                  q8_block=None, q8_fmt=F8_E5M2):
    """log_lo / log_hi (f32 [B, S], accumulated): the attention LOG side output (valid pairs only, queries below / from log_split),
    taken from the backward's own P instead of a second Q K^T walk in the forward (merlot_hip.h, merlot_attention_bwd).
    q8_block (f32[4]; shapes with attention_bwd_writes_q8 only): also the 8-bit float copy of dqkv, scaled by block[0], max|dqkv| -> block[3]; returns (dqkv, dqkv8)."""
    _chk(dout, BF16, 'dout'); _chk(seg, torch.int32, 'seg'); _chk(log_lo, F32, 'log_lo'); _chk(log_hi, F32, 'log_hi')
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, heads, S), device=qkv.device, dtype=F32)
    args = (_p(qkv), qkv.stride(0), _p(out), out.stride(0), _p(dout), dout.stride(0), _p(lse),
            _p(valid), _p(seg), _p(dqkv), dqkv.stride(0), _p(delta), B, S, heads, 0.125, _p(log_lo), _p(log_hi),
            S if log_split is None else int(log_split), float(log_weight))
    if q8_block is not None:
        _chk(q8_block, F32, 'q8_block')
        dqkv8 = _f8_rows(qkv.shape[0], qkv.shape[1], _F8_DTYPES[q8_fmt], qkv.device)
        call('merlot_attention_bwd_q8', *args, _p(dqkv8), dqkv8.stride(0), int(q8_fmt), _p(q8_block), *_attn_ws(), _stream())
        return dqkv, dqkv8
    call('merlot_attention_bwd', *args, *_attn_ws(), _stream())
    return dqkv


def attention_colsum(qkv, lse, B, S, heads, colsum_lo, colsum_hi=None, *, qsplit=None, valid=None, valid_q_only=False,
                     weight=1.0, seg=None):
    _chk(seg, torch.int32, 'seg')
    call('merlot_attention_colsum', _p(qkv), qkv.stride(0), _p(lse), _p(valid), _p(seg), _p(colsum_lo), _p(colsum_hi),
         S if qsplit is None else qsplit, 1 if valid_q_only else 0, float(weight), B, S, heads, 0.125, _stream())


def cast_bf16(src, dst=None):
    _chk(src, F32, 'src')
    assert src.is_contiguous()
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=BF16)
    call('merlot_cast_f32_bf16', _p(src), _p(dst), src.numel(), _stream())
    return dst


def cast_transpose_bf16(src, dst=None, ld_dst=None):
    _chk(src, F32, 'src')
    assert src.dim() == 2 and src.is_contiguous()
    R, C = src.shape
    if dst is None:
        dst = torch.empty((C, R), device=src.device, dtype=BF16)
    call('merlot_cast_transpose_f32_bf16', _p(src), _p(dst), R, C, R if ld_dst is None else ld_dst, _stream())
    return dst


def cast_transpose_batched(src_base, dst_base, jobs, total_tiles):
    """jobs: device int64 [n, 6] (see include/merlot_hip.h); one launch for all transposed bf16 copies."""
    _chk(src_base, F32, 'src_base'); _chk(dst_base, BF16, 'dst_base'); _chk(jobs, torch.int64, 'jobs')
    call('merlot_cast_transpose_batched', _p(src_base), _p(dst_base), _p(jobs), jobs.shape[0], int(total_tiles), _stream())


def colsum_bf16(x, out, accumulate=True, n=None):
    _chk(x, BF16, 'x'); _chk(out, F32, 'out')
    T = x.shape[0]
    N = x.shape[1] if n is None else n
    call('merlot_colsum_bf16', _p(x), x.stride(0), _p(out), T, N, 1 if accumulate else 0, _stream())


def gather_add4(rows, H, a=None, ia=None, b=None, ib=None, c=None, ic=None, d=None, id_=None):
    dev = next(t for t in (a, b, c, d) if t is not None).device
    out = torch.empty((rows, H), device=dev, dtype=F32)
    a_bf16 = 1 if (a is not None and a.dtype == BF16) else 0
    for t in (b, c, d):
        _chk(t, F32, 'table')
    for t in (ia, ib, ic, id_):
        _chk(t, torch.int32, 'index')
    call('merlot_gather_add4', _p(a), a_bf16, _p(ia), _p(b), _p(ib), _p(c), _p(ic), _p(d), _p(id_), _p(out), rows, H,
         _stream())
    return out


def scatter_add_rows(src, idx, table):
    """table[idx[r]] += src[r].  Many rows onto a table (the word-embedding gradient: 65 536 rows onto ~20 000 tokens): sorted by index
    on the device and reduced per run of equal indices (merlot_scatter_add_sorted) instead of one fp32 atomic per element."""
    _chk(src, F32, 'src'); _chk(table, F32, 'table'); _chk(idx, torch.int32, 'idx')
    assert src.is_contiguous() and table.is_contiguous()
    H = src.shape[-1]
    rows = src.numel() // H
    if idx is not None and rows >= 4096 and H % 4 == 0:
        sidx, perm = torch.sort(idx.reshape(-1), stable=True)
        perm = perm.to(torch.int32)
        call('merlot_scatter_add_sorted', _p(src), _p(perm), _p(sidx), _p(table), rows, H, _stream())
        return
    call('merlot_scatter_add_rows', _p(src), _p(idx), _p(table), rows, H, _stream())


def dropout_apply(x, p, seed):
    _chk(x, BF16, 'x')
    assert x.is_contiguous()
    y = torch.empty_like(x)
    N = x.shape[-1]
    call('merlot_dropout_apply', _p(x), _p(y), x.numel() // N, N, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream())
    return y


def cls_avgpool_fwd(x, n_img, h1, w1, cls_skip, pool):
    _chk(x, BF16, 'x')
    H = x.shape[-1]
    vl = 1 + (h1 // pool) * (w1 // pool)
    out = torch.empty((n_img, vl, H), device=x.device, dtype=F32)
    call('merlot_cls_avgpool_fwd', _p(x), _p(out), n_img, h1, w1, cls_skip, pool, H, _stream())
    return out


def cls_avgpool_bwd(dout, n_img, h1, w1, cls_skip, pool):
    _chk(dout, F32, 'dout')
    assert dout.is_contiguous()
    H = dout.shape[-1]
    dx = torch.empty((n_img, cls_skip + h1 * w1, H), device=dout.device, dtype=BF16)
    call('merlot_cls_avgpool_bwd', _p(dout), _p(dx), n_img, h1, w1, cls_skip, pool, H, _stream())
    return dx


def softmax_ce(logits, labels, C, *, rowscale=None, dlogits_dtype=None, ld_dl=None, want_argmax=True, out=None):
    """out: optional (loss, argmax, dlogits) views to fill (row chunks of caller-owned tensors) instead of fresh tensors."""
    _chk(logits, F32, 'logits'); _chk(labels, torch.int32, 'labels'); _chk(rowscale, F32, 'rowscale')
    rows = logits.shape[0]
    if out is not None:
        loss, am, dl = out
        _chk(loss, F32, 'loss'); _chk(am, torch.int32, 'argmax'); _chk(dl, dlogits_dtype, 'dlogits')
        ld_dl = dl.stride(0)
        if loss.shape[0] != rows or dl.shape[0] != rows:
            raise ValueError("softmax_ce: out views must have one row per logits row")
    else:
        loss = torch.empty(rows, device=logits.device, dtype=F32)
        am = torch.empty(rows, device=logits.device, dtype=torch.int32) if want_argmax else None
        dl = None
        if dlogits_dtype is not None:
            ld_dl = ld_dl or logits.shape[1]
            dl = torch.empty((rows, ld_dl), device=logits.device, dtype=dlogits_dtype)
    call('merlot_softmax_ce', _p(logits), logits.stride(0), _p(labels), _p(loss), _p(am), _p(rowscale), _p(dl),
         1 if dlogits_dtype == BF16 else 0, ld_dl or 0, rows, C, _stream())
    return loss, am, dl


def vocab_ce(hb, table, out_bias, targets, rowscale, vocab, ld_dl):
    """MLM head tail in one C-ABI call (merlot_vocab_ce_fwd): logits GEMM + softmax cross-entropy through an fp32 scratch that
    lives for this call only (a caching-allocator hit: the 2.6 GB block is free for the backward's tensors afterwards)
    -> (loss f32 [T], argmax int32 [T], dlogits bf16 [T, ld_dl])."""
    _chk(hb, BF16, 'h'); _chk(table, BF16, 'table'); _chk(out_bias, F32, 'out_bias'); _chk(targets, torch.int32, 'targets')
    _chk(rowscale, F32, 'rowscale')
    T, K = hb.shape
    need = LIB.query('merlot_vocab_ce_scratch_bytes', T, vocab)
    scratch = torch.empty((need + 3) // 4, device=hb.device, dtype=F32)
    loss = torch.empty(T, device=hb.device, dtype=F32)
    am = torch.empty(T, device=hb.device, dtype=torch.int32)
    dl = torch.empty((T, ld_dl), device=hb.device, dtype=BF16)
    call('merlot_vocab_ce_fwd', _p(hb), hb.stride(0), _p(table), table.stride(0), _p(out_bias), _p(targets), _p(rowscale), _p(loss),
         _p(am), _p(dl), ld_dl, T, vocab, K, _p(scratch), scratch.numel() * 4, *_nt_ws(), _stream())
    return loss, am, dl


def l2norm_fwd(x):
    _chk(x, F32, 'x')
    assert x.is_contiguous()
    y = torch.empty_like(x)
    inv = torch.empty(x.shape[0], device=x.device, dtype=F32)
    call('merlot_l2norm_fwd', _p(x), _p(y), _p(inv), x.shape[0], x.shape[1], _stream())
    return y, inv


def l2norm_bwd(dy, y, inv):
    dx = torch.empty_like(y)
    call('merlot_l2norm_bwd', _p(dy.contiguous()), _p(y), _p(inv), _p(dx), y.shape[0], y.shape[1], _stream())
    return dx


def gelu_fwd(x):
    _chk(x, F32, 'x')
    y = torch.empty_like(x)
    call('merlot_gelu_fwd', _p(x), _p(y), x.numel(), _stream())
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    call('merlot_gelu_bwd', _p(dy.contiguous()), _p(x), _p(dx), x.numel(), _stream())
    return dx


def mask_inputs(ids, summs, gumbel, span_lower, span_upper, random_ids, option, num_topk, num_to_mask, w_nontopk, w_topk,
                log_nontopk, log_topk, max_weight, mask_token=1):
    B, L = ids.shape
    for t in (ids, span_lower, span_upper, random_ids, option):
        _chk(t, torch.int32, 'int input')
    _chk(summs, F32, 'summs'); _chk(gumbel, F32, 'gumbel')
    masked_ids = torch.empty_like(ids)
    masked_idx = torch.empty((B, num_to_mask), device=ids.device, dtype=torch.int32)
    call('merlot_mask_inputs', _p(ids), _p(summs), _p(gumbel), _p(span_lower), _p(span_upper), _p(random_ids), _p(option),
         _p(masked_ids), _p(masked_idx), B, L, num_topk, num_to_mask, float(w_nontopk), float(w_topk), float(log_nontopk),
         float(log_topk), float(max_weight), mask_token, _stream())
    return masked_ids, masked_idx


def temporal_labels(video_src_ids, shuffled_idx, B, n):
    _chk(video_src_ids, torch.int32, 'video_src_ids'); _chk(shuffled_idx, torch.int32, 'shuffled_idx')
    labels = torch.empty(B * n * n, device=video_src_ids.device, dtype=torch.int32)
    weights = torch.empty(B * n * n, device=video_src_ids.device, dtype=F32)
    call('merlot_temporal_labels', _p(video_src_ids), _p(shuffled_idx), _p(labels), _p(weights), B, n, _stream())
    return labels, weights


def shuffled_idx(num_shuffle, u_select, u_perm, B, n, offset=16):
    _chk(num_shuffle, torch.int32, 'num_shuffle'); _chk(u_select, F32, 'u_select'); _chk(u_perm, F32, 'u_perm')
    out = torch.empty(B * n, device=num_shuffle.device, dtype=torch.int32)
    call('merlot_shuffled_idx', _p(num_shuffle), _p(u_select), _p(u_perm), _p(out), B, n, offset, _stream())
    return out


def adamw_step(param, grad, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0, wd_flags=None):
    _chk(param, F32, 'param'); _chk(grad, F32, 'grad'); _chk(wd_flags, torch.uint8, 'wd_flags')
    state_bf16 = 1 if m.dtype in (BF16, torch.int16, torch.uint16) else 0
    call('merlot_adamw_step', _p(param), _p(grad), _p(m), _p(v), param.numel(), float(lr), float(beta1), float(beta2),
         float(eps), float(weight_decay), float(grad_scale), _p(wd_flags), state_bf16, _stream())


# ---- ResNet-hybrid stem ------------------------------------------------------------------------------
def im2col3x3(x, stride=1, shift=0.0):
    """x NHWC bf16 -> (patches [N*Ho*Wo, Kp] bf16, Kp); k = (ky, kx, c), Kp = 9*C rounded up to 64 (zero columns)."""
    _chk(x, BF16, 'x')
    N, H, W, C = x.shape
    assert x.is_contiguous()
    Kp = (9 * C + 63) // 64 * 64
    out = torch.empty((N * (H // stride) * (W // stride), Kp), device=x.device, dtype=BF16)
    call('merlot_im2col3x3', _p(x), _p(out), N, H, W, C, int(stride), Kp, float(shift), _stream())
    return out


_ZEROS = {}


def _zero_page(device):
    """64 zeroed bf16 on `device` (the source of a tap outside the image in merlot_conv3x3_bf16), allocated once per device."""
    z = _ZEROS.get(device)
    if z is None:
        z = _ZEROS[device] = torch.zeros(64, device=device, dtype=BF16)
    return z


def conv3x3(x, w, co=None):
    """3x3 / stride 1 / SAME convolution as an implicit GEMM: x [N,H,W,C] bf16 (C % 32 == 0), w [>= Co, ld >= 9*C] bf16 with
    k = (ky, kx, c) -> y [N,H,W,Co] bf16.  Bit-identical to gemm_nt(im2col3x3(x), w) without the patch matrix in HBM."""
    _chk(x, BF16, 'x'); _chk(w, BF16, 'w')
    N, H, W, C = x.shape
    Co = w.shape[0] if co is None else co
    assert x.is_contiguous() and w.stride(1) == 1 and w.shape[0] >= Co and w.shape[1] >= 9 * C
    y = torch.empty((N, H, W, Co), device=x.device, dtype=BF16)
    call('merlot_conv3x3_bf16', _p(x), _p(w), w.stride(0), _p(y), Co, N, H, W, C, Co, _p(_zero_page(x.device)), _stream())
    return y


def conv3x3_wgrad(dy, x, dw, accumulate=False):
    """Kernel gradient of conv3x3 without the patch matrix: dw[co][(ky,kx,c)] (+)= sum_pixels dy[pix][co] * x[pix + tap][c].
    dy [N*H*W, Co] bf16 (row stride % 8 == 0), x [N,H,W,C] bf16, dw fp32 [>= Co, ld >= 9*C]."""
    _chk(dy, BF16, 'dy'); _chk(x, BF16, 'x'); _chk(dw, F32, 'dw')
    N, H, W, C = x.shape
    Co = dy.shape[1]
    assert x.is_contiguous() and dy.stride(1) == 1 and dw.stride(1) == 1 and dy.shape[0] == N * H * W
    assert dw.shape[0] >= Co and dw.shape[1] >= 9 * C
    nbytes = LIB.query('merlot_conv3x3_wgrad_workspace_bytes', N, H, W, C, Co)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=F32) if nbytes else None       # caller-owned partial tiles of the pixel ranges
    call('merlot_conv3x3_wgrad_bf16', _p(dy), dy.stride(0), _p(x), _p(dw), dw.stride(0), N, H, W, C, Co, 1 if accumulate else 0,
         _p(_zero_page(x.device)), _p(ws), nbytes, _stream())
    return dw


def col2im3x3(dpatches, N, H, W, C, stride=1):
    _chk(dpatches, BF16, 'dpatches')
    assert dpatches.is_contiguous()
    dx = torch.empty((N, H, W, C), device=dpatches.device, dtype=BF16)
    call('merlot_col2im3x3', _p(dpatches), _p(dx), N, H, W, C, int(stride), dpatches.shape[1], _stream())
    return dx


# GroupNorm in one launch per direction (merlot_groupnorm_*_fused, ABI v10 / v11), measured per shape of the as-shipped stem at 896 frames (profiles/r06_z3_gn_fused_fwd.txt,
# r06_z4_gn_fused_fwd.txt, r06_z5_gn_fused_bwd.txt): the FORWARD wins on all thirteen shapes (-3 ... -50 %: x is read once); the BACKWARD loses on twelve of them (x1.15 ... x2.2
# on per-slice slots, ABI v11; x1.4 ... x15 with its first version's returning atomics) and a call at 66 slices per sample has been seen to take seconds.
# 'fwd' (default) = one-launch forward, two-launch backward; True = both one-launch (A/B only); False = both two-launch (bench.py --gn-fused / --no-gn-fused).
GN_FUSED = 'fwd'
GN_FUSED_MAX_SLICES = 40                                   # csrc/conv.hip: both one-launch entries refuse more slices per sample (see _gn_fused_slices)


def _gn_ws(device, N, C, groups):
    """caller-owned workspace of one fused GroupNorm call (zeroed by the call itself, on its stream)."""
    return torch.empty(LIB.query('merlot_groupnorm_fused_workspace_bytes', N, C, groups) // 4, device=device, dtype=torch.int32)


def _gn_fused_slices(HW, C, res, bwd=False):
    """slices per sample of the one-launch GroupNorm entries (csrc/conv.hip: 256-thread blocks, a thread holds 8 or 16 positions of 8 channels); they refuse more than 40."""
    cpr = C // 8
    threads = max(cpr, (256 // cpr) * cpr)
    pstep = threads // cpr
    s8 = -(-HW // (8 * pstep))
    if bwd or (res and s8 <= GN_FUSED_MAX_SLICES):
        return s8
    return -(-HW // (16 * pstep))


def groupnorm_fwd(x, gamma, beta, *, res=None, relu=True, groups=32, eps=1e-4):
    _chk(x, BF16, 'x'); _chk(gamma, F32, 'gamma'); _chk(beta, F32, 'beta'); _chk(res, BF16, 'res')
    N, H, W, C = x.shape
    assert x.is_contiguous() and (res is None or res.is_contiguous())
    y = torch.empty_like(x)
    stats = torch.empty((N, groups, 2), device=x.device, dtype=F32)
    if GN_FUSED and _gn_fused_slices(H * W, C, res is not None) <= GN_FUSED_MAX_SLICES:      # one launch, x read once (ABI v10); above 40 slices per sample (unmeasured) two launches
        ws = _gn_ws(x.device, N, C, groups)
        call('merlot_groupnorm_fwd_fused', _p(x), _p(gamma), _p(beta), _p(res), _p(y), _p(stats), N, H, W, C, groups, float(eps),
             1 if relu else 0, _p(ws), ws.numel() * 4, _stream())
        return y, stats
    call('merlot_groupnorm_fwd', _p(x), _p(gamma), _p(beta), _p(res), _p(y), _p(stats), N, H, W, C, groups, float(eps),
         1 if relu else 0, _stream())
    return y, stats


def groupnorm_bwd(dy, y, x, stats, gamma, dgamma, dbeta, *, beta=None, relu=True, want_dres=False, groups=32, eps=1e-4):
    """y = None with relu (layers without a residual add; needs beta): the ReLU mask is recomputed from x instead of read from y."""
    _chk(dy, BF16, 'dy'); _chk(x, BF16, 'x'); _chk(stats, F32, 'stats'); _chk(dgamma, F32, 'dgamma'); _chk(dbeta, F32, 'dbeta')
    _chk(beta, F32, 'beta')
    N, H, W, C = x.shape
    assert dy.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    gsum = torch.empty((N, groups, 2), device=x.device, dtype=F32)
    if GN_FUSED is True and _gn_fused_slices(H * W, C, True, bwd=True) <= GN_FUSED_MAX_SLICES:      # one launch, x | dy read once (per-slice slots in the workspace: ABI v11); A/B only
        ws = torch.empty(LIB.query('merlot_groupnorm_bwd_fused_workspace_bytes', N, H, W, C, groups) // 4, device=x.device, dtype=torch.int32)
        call('merlot_groupnorm_bwd_fused', _p(dy), _p(y), _p(x), _p(stats), _p(gamma), _p(beta), _p(dgamma), _p(dbeta), _p(gsum), _p(dx),
             _p(dres), N, H, W, C, groups, float(eps), 1 if relu else 0, _p(ws), ws.numel() * 4, _stream())
        return dx, dres
    call('merlot_groupnorm_bwd', _p(dy), _p(y), _p(x), _p(stats), _p(gamma), _p(beta), _p(dgamma), _p(dbeta), _p(gsum), _p(dx), _p(dres),
         N, H, W, C, groups, float(eps), 1 if relu else 0, _stream())
    return dx, dres


def avgpool2_fwd(x):
    _chk(x, BF16, 'x')
    N, H, W, C = x.shape
    assert x.is_contiguous()
    y = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=BF16)
    call('merlot_avgpool2_fwd', _p(x), _p(y), N, H, W, C, _stream())
    return y


def avgpool2_bwd(dy):
    _chk(dy, BF16, 'dy')
    N, Ho, Wo, C = dy.shape
    assert dy.is_contiguous()
    dx = torch.empty((N, 2 * Ho, 2 * Wo, C), device=dy.device, dtype=BF16)
    call('merlot_avgpool2_bwd', _p(dy), _p(dx), N, 2 * Ho, 2 * Wo, C, _stream())
    return dx


# ---- input pipeline: frame preprocessing --------------------------------------------------------------
def image_frames(src, jobs_host, jobs_dev, n_img, out_h, out_w):
    """decoded frames (uint8, device, concatenated) + job table -> bf16 [n_img, out_h, out_w, 3] (csrc/image.hip)."""
    _chk(src, torch.uint8, 'src'); _chk(jobs_dev, torch.uint8, 'jobs_dev')
    if jobs_host.device.type != 'cpu' or jobs_host.dtype != torch.uint8 or jobs_host.numel() != n_img * 56:
        raise ValueError('jobs_host: expected a CPU uint8 tensor holding n_img merlot_image_job_t entries')
    if jobs_dev.numel() != jobs_host.numel():
        raise ValueError('jobs_dev must mirror jobs_host')
    nbytes = LIB.query('merlot_image_frames_workspace_bytes', n_img, out_h, out_w)
    ws = torch.empty(nbytes // 4, device=src.device, dtype=F32)
    out = torch.empty((n_img, out_h, out_w, 3), device=src.device, dtype=BF16)
    call('merlot_image_frames', _p(src), src.numel(), jobs_host.data_ptr(), _p(jobs_dev), n_img, _p(out), out_h, out_w,
         _p(ws), nbytes, _stream())
    return out


def jpeg_idct_rgb(coef, infos, plane_bytes, dst_bytes, dst=None):
    """coef: int16 device tensor (all images); infos: numpy array of merlot_jpeg_info_t (coef_base / plane_offset / dst_offset
    set) -> flat uint8 device tensor of dst_bytes holding the RGB frames (`dst`: write into this existing buffer)."""
    import numpy as np
    assert coef.is_cuda and coef.dtype == torch.int16
    infos_host = torch.from_numpy(np.ascontiguousarray(infos).view(np.uint8).reshape(-1).copy())
    infos_dev = infos_host.to(coef.device, non_blocking=True)
    ws = torch.empty(max(int(plane_bytes), 16), device=coef.device, dtype=torch.uint8)
    if dst is None:
        dst = torch.empty(max(int(dst_bytes), 16), device=coef.device, dtype=torch.uint8)
    assert dst.is_cuda and dst.dtype == torch.uint8 and dst.is_contiguous()
    call('merlot_jpeg_idct_rgb', _p(coef), infos_host.data_ptr(), _p(infos_dev), len(infos), _p(ws), ws.numel(), _p(dst), dst.numel(),
         _stream())
    return dst


def weight_std_fwd(k2d, Kp, Cop):
    """k2d: fp32 [K, Co] (HWIO kernel flattened) -> (khat fp32 [K, Co], rstd [Co], wb bf16 [Co, Kp], wbT bf16 [Kp, Cop])."""
    _chk(k2d, F32, 'k2d')
    assert k2d.is_contiguous()
    K, Co = k2d.shape
    khat = torch.empty_like(k2d)
    rstd = torch.empty(Co, device=k2d.device, dtype=F32)
    wb = torch.zeros((Co, Kp), device=k2d.device, dtype=BF16) if Kp != K else torch.empty((Co, Kp), device=k2d.device, dtype=BF16)
    wbT = torch.zeros((Kp, Cop), device=k2d.device, dtype=BF16) if (Kp != K or Cop != Co) else torch.empty((Kp, Cop), device=k2d.device, dtype=BF16)
    call('merlot_weight_std_fwd', _p(k2d), K, Co, _p(khat), _p(rstd), _p(wb), Kp, _p(wbT), Cop, _stream())
    return khat, rstd, wb, wbT


def weight_std_fwd_batched(k_base, jobs, total_blocks, khat, rstd, wb, wbT, wdg):
    """every job of `jobs` (device int64 [n, 12], include/merlot_hip.h) in one launch: the standardised kernels of a whole stem."""
    _chk(k_base, F32, 'k_base'); _chk(khat, F32, 'khat'); _chk(rstd, F32, 'rstd'); _chk(wb, BF16, 'wb'); _chk(wbT, BF16, 'wbT'); _chk(wdg, BF16, 'wdg')
    call('merlot_weight_std_fwd_batched', _p(k_base), _p(jobs), jobs.shape[0], int(total_blocks), _p(khat), _p(rstd), _p(wb), _p(wbT),
         _p(wdg), _stream())


def weight_std_bwd_batched(dk, jobs, total_blocks, khat, rstd, gk_base):
    """gk_base[job's offset] += the standardisation's backward of every job's dkhat_t (device int64 [n, 8]) in one launch."""
    _chk(dk, F32, 'dk'); _chk(khat, F32, 'khat'); _chk(rstd, F32, 'rstd'); _chk(gk_base, F32, 'gk_base')
    call('merlot_weight_std_bwd_batched', _p(dk), _p(jobs), jobs.shape[0], int(total_blocks), _p(khat), _p(rstd), _p(gk_base), _stream())


def weight_std_bwd(dkhat_t, khat, rstd, gk2d):
    """gk2d [K, Co] (a view of the gradient arena) += the standardisation's backward of dkhat_t [Co(+pad), ld >= K]."""
    _chk(dkhat_t, F32, 'dkhat_t'); _chk(khat, F32, 'khat'); _chk(rstd, F32, 'rstd'); _chk(gk2d, F32, 'gk2d')
    assert gk2d.is_contiguous() and khat.is_contiguous()
    K, Co = khat.shape
    call('merlot_weight_std_bwd', _p(dkhat_t), dkhat_t.stride(0), _p(khat), _p(rstd), K, Co, _p(gk2d), _stream())
