"""Launches for the TCC passes of scripts/gpu_traffic_conv.sh: the implicit 3x3 convolution kernels (forward, weight gradient) and the explicit path they
replace (im2col3x3 + GEMM), one stem shape each, 1 024 frames.  No timing here -- rocprofv3 --pmc serialises the kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402

BF16 = torch.bfloat16
N = 1024
for (H, C, Co) in ((56, 64, 64), (28, 256, 256)):
    x = torch.randn(N, H, H, C, device='cuda').to(BF16)
    Kp = (9 * C + 63) // 64 * 64
    w = torch.zeros(Co, Kp, device='cuda', dtype=BF16)
    w[:, :9 * C] = (torch.randn(Co, 9 * C, device='cuda') / (3 * C ** 0.5)).to(BF16)
    dy = torch.randn(N * H * H, Co, device='cuda').to(BF16)
    dw = torch.zeros(Co, Kp, device='cuda')
    for _ in range(2):
        ops.conv3x3(x, w, Co)
        ops.conv3x3_wgrad(dy, x, dw)
        a = ops.im2col3x3(x)
        ops.gemm_nt(a, w)
        ops.gemm_tn(dy, a, dw, accumulate=False)
        del a
    torch.cuda.synchronize()
