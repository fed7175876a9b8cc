"""im2col3x3 / col2im3x3 / groupnorm forward: achieved bytes per second at the stem's shapes (1 024 frames)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402
from ab_lib_tn import bench  # noqa: E402

for (N, H, C, s) in ((1024, 56, 64, 1), (1024, 28, 128, 1), (1024, 14, 256, 1), (1024, 112, 32, 1), (256, 224, 3, 2)):
    x = torch.randn(N, H, H, C, device='cuda').bfloat16()
    a = ops.im2col3x3(x, s, 0.0)
    t_i = bench(lambda: ops.im2col3x3(x, s, 0.0), 10)
    gb_i = (x.numel() + a.numel()) * 2 / 1e9
    line = f'[{N} x {H}^2 x {C}] stride {s}: im2col {t_i:7.1f} us ({gb_i / t_i * 1e3:4.1f} TB/s of {gb_i:5.2f} GB)'
    if C % 8 == 0:
        dp = torch.randn_like(a)
        t_c = bench(lambda: ops.col2im3x3(dp, N, H, H, C, s), 10)
        line += f' | col2im {t_c:7.1f} us ({gb_i / t_c * 1e3:4.1f} TB/s)'
        g_, b_ = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
        t_g = bench(lambda: ops.groupnorm_fwd(x, g_, b_, relu=True), 10)
        line += f' | groupnorm fwd {t_g:7.1f} us ({3 * x.numel() * 2 / 1e9 / t_g * 1e3:4.1f} TB/s)'
    print(line, flush=True)
