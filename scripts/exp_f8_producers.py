"""Round 6: what the 8-bit copies cost the launches that produce them ('fuse'), at the config-#5 ViT row count.
python scripts/exp_f8_producers.py [rows]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_amd import ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 48 * 16 * 578
dev = 'cuda'


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def blk(s):
    return torch.tensor([s, 1.0 / s, 1.0, 0.0], device=dev)


print(torch.cuda.get_device_name(0), 'rows', T)
H, I = 768, 3072
g = torch.Generator(device=dev).manual_seed(0)
db2 = (torch.randn(T, H, device=dev, generator=g) * 1e-3).bfloat16()
w2t = (torch.randn(I, H, device=dev, generator=g) * 0.02).bfloat16()
u = torch.randn(T, I, device=dev, generator=g).bfloat16()
cs = torch.zeros(I, device=dev)
t0 = timeit(lambda: ops.gemm_nt(db2, w2t, epilogue=ops.EPI_DGELU, aux_in=u, colsum_out=cs))
b = blk(1e6)
t1 = timeit(lambda: ops.gemm_nt_q8(db2, w2t, b, 1, epilogue=ops.EPI_DGELU, aux_in=u, colsum_out=cs))
t2 = timeit(lambda: ops.gemm_nt_q8(db2, w2t, b, 1, epilogue=ops.EPI_DGELU, aux_in=u, colsum_out=cs, keep_bf16=False))
t3 = timeit(lambda: ops.gemm_nt_q8(db2, w2t, b, 0, epilogue=ops.EPI_DGELU, aux_in=u, colsum_out=cs))
print(f'GELU\' input gradient [T,768] x [768,3072]: bf16 out {t0:7.1f} us | + e5m2 copy {t1:7.1f} ({t1 / t0:.2f}x) | e5m2 copy only {t2:7.1f} ({t2 / t0:.2f}x) | + e4m3 copy {t3:7.1f}')
del db2, cs

x = torch.randn(T, H, device=dev, generator=g).bfloat16()
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
w1 = (torch.randn(I, H, device=dev, generator=g) * 0.02).bfloat16()
bias = torch.zeros(I, device=dev)
w8, sw = ops.quantize_e4m3(w1)
_, x8, rs, _, _ = ops.ln_fwd_q8(x, gamma, beta)
t0 = timeit(lambda: ops.gemm_fp8_nt(x8, None, w8, sw, bias=bias, epilogue=ops.EPI_GELU, aux_out=u, a_row_scale=rs))
b = blk(100.0)
t1 = timeit(lambda: ops.gemm_fp8_nt_q8(x8, None, w8, sw, b, bias=bias, aux_out=u, a_row_scale=rs))
t2 = timeit(lambda: ops.gemm_fp8_nt_q8(x8, None, w8, sw, b, bias=bias, aux_out=u, a_row_scale=rs, keep_bf16=False))
print(f'fc1 + GELU on e4m3 operands:               bf16 out {t0:7.1f} us | + e4m3 copy {t1:7.1f} ({t1 / t0:.2f}x) | e4m3 copy only {t2:7.1f} ({t2 / t0:.2f}x)')
a = torch.randn(T, I, device=dev, generator=g).bfloat16()
w2 = (torch.randn(H, I, device=dev, generator=g) * 0.02).bfloat16()
w28, sw2 = ops.quantize_e4m3(w2)
a8, sa = ops.quantize_f8(a, 0)
t0 = timeit(lambda: ops.gemm_nt(a, w2, bias=gamma, epilogue=ops.EPI_RESIDUAL, aux_in=x, dropout_p=0.1, dropout_seed=3))
t1 = timeit(lambda: ops.gemm_fp8_nt(a8, sa, w28, sw2, bias=gamma, epilogue=ops.EPI_RESIDUAL, aux_in=x, dropout_p=0.1, dropout_seed=3))
print(f'fc2 + residual:                            bf16 {t0:7.1f} us | e4m3 operands {t1:7.1f} ({t1 / t0:.2f}x)')
del a, a8, u

t0 = timeit(lambda: ops.ln_fwd(x, gamma, beta))
t1 = timeit(lambda: ops.ln_fwd_q8(x, gamma, beta))
b = blk(50.0)
t2 = timeit(lambda: ops.ln_fwd_q8t(x, gamma, beta, b, out_bf16=True))
t3 = timeit(lambda: ops.ln_fwd_q8t(x, gamma, beta, b))
print(f'LayerNorm forward: bf16 {t0:7.1f} us | + per-row e4m3 {t1:7.1f} | + per-tensor e4m3 {t2:7.1f} | per-tensor e4m3 only {t3:7.1f}')
dy = (torch.randn(T, H, device=dev, generator=g) * 1e-3).bfloat16()
dres = (torch.randn(T, H, device=dev, generator=g) * 1e-3).bfloat16()
_, _, mean, rstd = ops.ln_fwd(x, gamma, beta)
dg, dbt, bb = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
t0 = timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, gamma, dg, dbt, dres=dres, branch_bias_grad=bb, drop_p=0.1, drop_seed=5))
b = blk(1e6)
t1 = timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, gamma, dg, dbt, dres=dres, branch_bias_grad=bb, drop_p=0.1, drop_seed=5, db8_block=b, db8_fmt=1))
print(f'LayerNorm backward (+ branch gradient): {t0:7.1f} us | + e5m2 copy {t1:7.1f} ({t1 / t0:.2f}x)')
