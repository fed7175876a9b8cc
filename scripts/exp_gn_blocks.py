"""Round 6, call 45: blocks per launch of the two-launch GroupNorm BACKWARD (csrc/conv.hip gn_split: >= 2 048 blocks = 3 slices per sample at 896 frames -- 2 688 blocks on a chip that
holds 2 048 of them: a second, 31 %-full round?).  Experiments build, MERLOT_GN_BLOCKS = target blocks per launch; every as-shipped shape, each target twice (mirrored), best time."""
import _exp_lib  # noqa: F401
import os
import sys
import torch
from merlot_amd import ops

BF16 = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 896
# default: 1, 2, 3 (rounds 3 - 6), 4, 6, 10 slices per sample at 896 frames (capped by the sample's size)
TARGETS = tuple(int(t) for t in sys.argv[2].split(',')) if len(sys.argv) > 2 else (896, 1792, 2048, 3584, 5376, 8192)
APPLY = len(sys.argv) > 3 and sys.argv[3] == 'apply'      # the targets drive the APPLY pass alone; the sums pass keeps the product's rule
SHAPES = [(96, 176, 32, True, False, 2), (96, 176, 64, True, False, 1), (48, 88, 64, True, False, 6), (48, 88, 256, False, False, 1),
          (48, 88, 256, True, True, 3), (48, 88, 128, True, False, 2), (24, 44, 512, False, False, 1), (24, 44, 512, True, True, 4),
          (24, 44, 128, True, False, 6), (24, 44, 256, True, False, 2), (12, 22, 1024, False, False, 1), (12, 22, 1024, True, True, 9),
          (12, 22, 256, True, False, 16)]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


ops.GN_FUSED = False
tot = {t: 0.0 for t in TARGETS}
print(f'N = {N} frames; two-launch backward, us per call (best of two mirrored runs) at {"MERLOT_GN_BLOCKS_APPLY" if APPLY else "MERLOT_GN_BLOCKS"} = {TARGETS}', flush=True)
for H, W, C, relu, res, cnt in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(0)
    x = (torch.randn(N, H, W, C, generator=g, device='cuda') * 1.5 + 0.2).to(BF16)
    r = torch.randn(N, H, W, C, generator=g, device='cuda').to(BF16) if res else None
    dy = torch.randn(N, H, W, C, generator=g, device='cuda').to(BF16)
    gamma = 1 + 0.1 * torch.randn(C, generator=g, device='cuda')
    beta = 0.1 * torch.randn(C, generator=g, device='cuda')
    dga, dbe = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    y, stats = ops.groupnorm_fwd(x, gamma, beta, res=r, relu=relu)
    yy = y if (relu and res) else None
    best = {t: 1e30 for t in TARGETS}
    for order in (TARGETS, TARGETS[::-1]):
        for t in order:
            os.environ['MERLOT_GN_BLOCKS_APPLY' if APPLY else 'MERLOT_GN_BLOCKS'] = str(t)
            best[t] = min(best[t], timed(lambda: ops.groupnorm_bwd(dy, yy, x, stats, gamma, dga, dbe, beta=beta, relu=relu, want_dres=res)))
    for t in TARGETS:
        tot[t] += best[t] * cnt
    print(f'{H:3d}x{W:3d}x{C:4d} relu {int(relu)} res {int(res)} x{cnt:2d}: ' + ' '.join(f'{best[t]:8.1f}' for t in TARGETS), flush=True)
    del x, r, dy
print('per step (54 layers), ms: ' + ' '.join(f'{tot[t] / 1e3:8.2f}' for t in TARGETS))
