"""GroupNorm backward with the ReLU mask read from the stored output (y) or recomputed from x (y = None, beta given): us per call."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402
from ab_lib_tn import bench  # noqa: E402

for (N, H, C) in ((1024, 56, 64), (1024, 56, 256), (1024, 28, 128), (1024, 14, 256), (1024, 112, 32)):
    x = torch.randn(N, H, H, C, device='cuda').bfloat16()
    g_, b_ = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    y, st = ops.groupnorm_fwd(x, g_, b_, relu=True)
    dy = torch.randn_like(x)
    gg, gb = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    t_y = bench(lambda: ops.groupnorm_bwd(dy, y, x, st, g_, gg, gb, beta=b_, relu=True), 10)
    t_x = bench(lambda: ops.groupnorm_bwd(dy, None, x, st, g_, gg, gb, beta=b_, relu=True), 10)
    gbytes = x.numel() * 2 / 1e9
    print(f'[{N} x {H}^2 x {C}] {gbytes:5.2f} GB per tensor: mask from y {t_y:7.1f} us ({7 * gbytes / t_y * 1e3:4.1f} TB/s)   mask from x {t_x:7.1f} us ({5 * gbytes / t_x * 1e3:4.1f} TB/s)', flush=True)
