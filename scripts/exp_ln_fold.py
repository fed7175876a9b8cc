"""Round 6: the LayerNorm fold, same box -- residual GEMM + stand-alone LayerNorm launch against merlot_gemm_bf16_nt_ln (one launch), at the three
row counts of the bench step (ViT 405 504, joint 167 936, text-only 65 536) for proj (K = 768) and fc2 (K = 3 072), dropout 0.1.  Mirrored order."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def bench(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


torch.manual_seed(0)
for T in (405504, 167936, 65536):
    for name, K in (('proj', 768), ('fc2', 3072)):
        a = torch.randn(T, K, device='cuda').to(BF16)
        w = (torch.randn(768, K, device='cuda') * 0.02).to(BF16)
        bias = torch.randn(768, device='cuda') * 0.1
        res = torch.randn(T, 768, device='cuda').to(BF16)
        g, b = torch.ones(768, device='cuda'), torch.zeros(768, device='cuda')
        two = lambda: ops.ln_fwd(ops.gemm_nt(a, w, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1), g, b)   # noqa: E731
        gemm_only = lambda: ops.gemm_nt(a, w, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1)                # noqa: E731
        one = lambda: ops.gemm_nt_ln(a, w, g, b, bias=bias, aux_in=res, dropout_p=0.1, dropout_seed=1)                                        # noqa: E731
        t = [bench(two), bench(one), bench(one), bench(two)]
        tg = bench(gemm_only)
        print(f'{name:5s} T={T:7d} K={K:4d}  GEMM + ln_fwd {t[0]:7.1f} / {t[3]:7.1f} us (GEMM alone {tg:7.1f})   fused {t[1]:7.1f} / {t[2]:7.1f} us   '
              f'saved {0.5 * (t[0] + t[3]) - 0.5 * (t[1] + t[2]):6.1f} us per layer-half', flush=True)
