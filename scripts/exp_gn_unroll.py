"""Round 6, call 59: positions per loop iteration of the two-launch GroupNorm backward (U = 1: rounds 3 - 6; 2, 4: every load of the iteration issued before the first use).  With one block
per sample (the new block rule) a CU holds 14 waves and each keeps ONE position's 2 - 3 loads in flight.  Experiments build, MERLOT_GN_UNROLL; as-shipped shapes, argv[1] frames."""
import _exp_lib  # noqa: F401
import os
import sys
import torch
from merlot_amd import ops

BF16 = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 896
US = (1, 2, 4)
SHAPES = [(96, 176, 32, True, False, 2), (96, 176, 64, True, False, 1), (48, 88, 64, True, False, 6), (48, 88, 256, False, False, 1),
          (48, 88, 256, True, True, 3), (48, 88, 128, True, False, 2), (24, 44, 512, False, False, 1), (24, 44, 512, True, True, 4),
          (24, 44, 128, True, False, 6), (24, 44, 256, True, False, 2), (12, 22, 1024, False, False, 1), (12, 22, 1024, True, True, 9),
          (12, 22, 256, True, False, 16)]


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


ops.GN_FUSED = False
tot = {u: 0.0 for u in US}
print(f'N = {N} frames; two-launch backward, us per call (best of two mirrored runs) at MERLOT_GN_UNROLL = {US}', flush=True)
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
    best, ref = {u: 1e30 for u in US}, None
    for order in (US, US[::-1]):
        for u in order:
            os.environ['MERLOT_GN_UNROLL'] = str(u)
            dx, _ = ops.groupnorm_bwd(dy, yy, x, stats, gamma, dga, dbe, beta=beta, relu=relu, want_dres=res)
            if ref is None:
                ref = dx.clone()
            err = float((dx.float() - ref.float()).norm() / ref.float().norm())      # (the sums pass adds with atomics: last bits move between any two runs)
            assert err < 1e-3, ('dx differs', u, err)
            best[u] = min(best[u], timed(lambda: ops.groupnorm_bwd(dy, yy, x, stats, gamma, dga, dbe, beta=beta, relu=relu, want_dres=res)))
    for u in US:
        tot[u] += best[u] * cnt
    print(f'{H:3d}x{W:3d}x{C:4d} relu {int(relu)} res {int(res)} x{cnt:2d}: ' + ' '.join(f'{best[u]:8.1f}' for u in US), flush=True)
    del x, r, dy
print('per step (54 layers), ms: ' + ' '.join(f'{tot[u] / 1e3:8.2f}' for u in US))
