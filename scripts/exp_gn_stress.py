"""Round 6, call 42: does the one-launch GroupNorm FORWARD (the product default) ever take a call of seconds, as its backward did once at 66 slices per sample
(profiles/r06_z5_gn_fused_bwd.txt)?  Every as-shipped shape at 896 frames, CALLS calls each timed on its own (events), median / max / calls above 3x the median.
argv[1] = 'bwd <shape>': the same for the one-launch backward of one shape."""
import os
import sys

sys.path.insert(0, '.')
if os.environ.get('MERLOT_GN_STATIC'):                    # the experiments build reads it (item = workgroup id instead of a claim)
    sys.path.insert(0, 'scripts')
    import _exp_lib  # noqa: F401
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402

BF16 = torch.bfloat16
SHAPES = [(96, 176, 32, True, False), (96, 176, 64, True, False), (48, 88, 64, True, False), (48, 88, 256, False, False), (48, 88, 256, True, True),
          (48, 88, 128, True, False), (24, 44, 512, False, False), (24, 44, 512, True, True), (24, 44, 128, True, False), (24, 44, 256, True, False),
          (12, 22, 1024, False, False), (12, 22, 1024, True, True), (12, 22, 256, True, False)]
N = 896
bwd = len(sys.argv) > 1 and sys.argv[1] == 'bwd'
CALLS = 60 if bwd else 300
todo = [int(sys.argv[2])] if bwd else range(len(SHAPES))
for idx in todo:
    H, W, C, relu, res = SHAPES[idx]
    g = torch.Generator(device='cuda').manual_seed(0)
    x = (torch.randn(N, H, W, C, generator=g, device='cuda') * 1.5 + 0.2).to(BF16)
    r = torch.randn(N, H, W, C, generator=g, device='cuda').to(BF16) if res else None
    gamma = 1 + 0.1 * torch.randn(C, generator=g, device='cuda')
    beta = 0.1 * torch.randn(C, generator=g, device='cuda')
    ops.GN_FUSED = 'fwd'
    y, stats = ops.groupnorm_fwd(x, gamma, beta, res=r, relu=relu)
    if bwd:
        dy = torch.randn(N, H, W, C, generator=g, device='cuda').to(BF16)
        dga, dbe = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        yy = y if (relu and res) else None
        ops.GN_FUSED = True
        fn = lambda: ops.groupnorm_bwd(dy, yy, x, stats, gamma, dga, dbe, beta=beta, relu=relu, want_dres=res)
    else:
        fn = lambda: ops.groupnorm_fwd(x, gamma, beta, res=r, relu=relu)
    fn()
    torch.cuda.synchronize()
    ts = []
    for k in range(CALLS):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
        if ts[-1] > 1e5:
            print(f'   shape {idx} call {k}: {ts[-1] / 1e6:.2f} s', flush=True)
    st = sorted(ts)
    med = st[len(st) // 2]
    print(f'shape {idx:2d} {H:3d}x{W:3d}x{C:4d} relu {int(relu)} res {int(res)} {"backward" if bwd else "forward"} one launch, {CALLS} calls: median {med:8.1f} us  max {st[-1]:10.1f} us  '
          f'calls above 3x the median: {sum(t > 3 * med for t in ts)}', flush=True)
    del x, r, y
