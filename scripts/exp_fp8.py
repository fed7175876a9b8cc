"""fp8 vs bf16 NT GEMM (the same ping-pong kernel, e4m3 vs bf16 operands) at the transformer shapes of configs #2 / #5."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)
ROWS = [int(r) for r in os.environ.get('ROWS', '101376,147968').split(',')]      # config #2 ViT rows; config #5: 16 x 16 x 578
SHAPES = [('qkv', 2304, 768, ops.EPI_NONE), ('proj', 768, 768, ops.EPI_RESIDUAL), ('fc1', 3072, 768, ops.EPI_GELU),
          ('fc2', 768, 3072, ops.EPI_RESIDUAL)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in ROWS:
    for name, N, K, epi in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == ops.EPI_RESIDUAL else None
        aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == ops.EPI_GELU else None
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        a8, sa = ops.quantize_e4m3(a)
        b8, sb = ops.quantize_e4m3(b)
        t16 = timeit(lambda: ops.gemm_nt(a, b, bias=bias, epilogue=epi, aux_in=res, aux_out=aux, out=out))
        t8 = timeit(lambda: ops.gemm_fp8_nt(a8, sa, b8, sb, bias=bias, epilogue=epi, aux_in=res, aux_out=aux, out=out))
        tq = timeit(lambda: ops.quantize_e4m3(a, out=a8))
        fl = 2.0 * M * N * K
        print(f"{name:5s} [{M} x {N} x {K}]  bf16 {t16:7.1f} us {fl / t16 * 1e-6:6.0f} TF | fp8 {t8:7.1f} us {fl / t8 * 1e-6:6.0f} TF | "
              f"quantize A {tq:6.1f} us ({M * K * 5 / tq * 1e-3:5.0f} GB/s) | fp8 incl. quantize {fl / (t8 + tq) * 1e-6:6.0f} TF", flush=True)
