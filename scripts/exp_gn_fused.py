"""Round 6: GroupNorm in one launch per direction (merlot_groupnorm_*_fused, ABI v10) against the two-launch entries, at the GroupNorm
shapes of the as-shipped hybrid stem (192 x 352 frames, resnet_layers [3, 4, 9], 896 frames per step = bench.py --native-yaml).  Same box, alternating.
    python scripts/exp_gn_fused.py > gpurun_out/r06_x_gn_fused.txt"""
import sys
import torch

sys.path.insert(0, '.')
from merlot_amd import ops  # noqa: E402

BF16 = torch.bfloat16
N = 896
# (H, W, C, relu, res, layers of this shape in the stem)
SHAPES = [(96, 176, 32, True, False, 2), (96, 176, 64, True, False, 1), (48, 88, 64, True, False, 6), (48, 88, 256, False, False, 1),
          (48, 88, 256, True, True, 3), (48, 88, 128, True, False, 2), (24, 44, 512, False, False, 1), (24, 44, 512, True, True, 4),
          (24, 44, 128, True, False, 6), (24, 44, 256, True, False, 2), (12, 22, 1024, False, False, 1), (12, 22, 1024, True, True, 9),
          (12, 22, 256, True, False, 16)]


if len(sys.argv) > 1:                                      # one shape per process: scripts/gpu_r6_z.sh runs each under its own timeout
    SHAPES = [SHAPES[int(sys.argv[1])]]


def timed(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tot = {('fwd', False): 0.0, ('fwd', True): 0.0, ('bwd', False): 0.0, ('bwd', True): 0.0}
print(f'N = {N} frames; us per call, two launches | one launch; GB/s = algorithmic bytes (forward x + y [+ res]; backward x + dy + dx [+ y + dres]) / time', flush=True)
for H, W, C, relu, res, cnt in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(0)
    x = (torch.randn(N, H, W, C, generator=g, device='cuda') * 1.5 + 0.2).to(BF16)
    r = torch.randn(N, H, W, C, generator=g, device='cuda').to(BF16) if res else None
    dy = torch.randn(N, H, W, C, generator=g, device='cuda').to(BF16)
    gamma = 1 + 0.1 * torch.randn(C, generator=g, device='cuda')
    beta = 0.1 * torch.randn(C, generator=g, device='cuda')
    dga, dbe = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    tb = x.numel() * 2 / 1e9
    row = []
    for rep in range(2):
        for mode in (False, True):
            ops.GN_FUSED = mode
            y, stats = ops.groupnorm_fwd(x, gamma, beta, res=r, relu=relu)
            tf = timed(lambda: ops.groupnorm_fwd(x, gamma, beta, res=r, relu=relu))
            yy = y if (relu and res) else None
            tbw = timed(lambda: ops.groupnorm_bwd(dy, yy, x, stats, gamma, dga, dbe, beta=beta, relu=relu, want_dres=res))
            row.append((tf, tbw))
            if rep == 1:
                tot[('fwd', mode)] += tf * cnt
                tot[('bwd', mode)] += tbw * cnt
    fb, bb = (3 if res else 2) * tb, (5 if res else 3) * tb
    print(f'{H:3d}x{W:3d}x{C:4d} relu {int(relu)} res {int(res)} x{cnt:2d}: fwd {row[0][0]:7.1f} {row[2][0]:7.1f} | {row[1][0]:7.1f} {row[3][0]:7.1f} us '
          f'({fb / row[2][0] * 1e6:5.0f} -> {fb / row[3][0] * 1e6:5.0f} GB/s)   bwd {row[0][1]:7.1f} {row[2][1]:7.1f} | {row[1][1]:7.1f} {row[3][1]:7.1f} us '
          f'({bb / row[2][1] * 1e6:5.0f} -> {bb / row[3][1] * 1e6:5.0f} GB/s)', flush=True)
    del x, r, dy
if len(sys.argv) == 1:
    print(f"per step (54 layers): forward {tot[('fwd', False)] / 1e3:.2f} -> {tot[('fwd', True)] / 1e3:.2f} ms, backward {tot[('bwd', False)] / 1e3:.2f} -> {tot[('bwd', True)] / 1e3:.2f} ms")
