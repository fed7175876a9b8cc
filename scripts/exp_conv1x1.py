"""1x1 convolutions of the ResNet-hybrid stem on the GEMM kernels: achieved bytes per second (they are HBM-bound: 26-205 flop per byte) and the kernel the
plan picks, forward / input gradient / weight gradient, 1 024 frames."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402
from merlot_amd.lib import LIB  # noqa: E402
from ab_lib_tn import bench  # noqa: E402

BF16 = torch.bfloat16
N = int(os.environ.get('FRAMES', 1024))
for (H, K, Co) in ((56, 64, 64), (56, 64, 256), (56, 256, 64), (56, 256, 128), (28, 128, 512), (28, 512, 128), (28, 512, 256), (14, 256, 1024), (14, 1024, 256)):
    M = N * H * H
    a = torch.randn(M, K, device='cuda').to(BF16)
    w = (torch.randn(Co, K, device='cuda') / K ** 0.5).to(BF16)
    wT = w.t().contiguous()
    dy = torch.randn(M, Co, device='cuda').to(BF16)
    dw = torch.zeros(Co, K, device='cuda')
    t_f = bench(lambda: ops.gemm_nt(a, w), 10)
    t_b = bench(lambda: ops.gemm_nt(dy, wT), 10)
    t_w = bench(lambda: ops.gemm_tn(dy, a, dw, accumulate=False), 10)
    gb = (M * K + M * Co) * 2 / 1e9
    plan_f = LIB.query('merlot_gemm_bf16_nt_plan', M, Co, K)
    plan_b = LIB.query('merlot_gemm_bf16_nt_plan', M, K, Co)
    print(f'[{N} x {H}^2] {K:4d} -> {Co:4d}: {gb:5.2f} GB | forward {t_f:7.1f} us ({gb / t_f * 1e3:4.1f} TB/s, plan {plan_f}) | input gradient {t_b:7.1f} us ({gb / t_b * 1e3:4.1f} TB/s, plan {plan_b}) | '
          f'weight gradient {t_w:7.1f} us ({gb / t_w * 1e3:4.1f} TB/s)', flush=True)
    del a, dy
    torch.cuda.empty_cache()
