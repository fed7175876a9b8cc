"""Round 6, call 48: blocks per launch of the LayerNorm backward (csrc/layernorm.hip: <= 2 048 blocks, each ending in 2 - 3 atomics per column): the GroupNorm backward gained 14 % from
fewer blocks (profiles/r06_z8_gn_blocks.txt) -- does this one?  Experiments build, MERLOT_LN_BWD_BLOCKS; the step's three row counts, with and without the branch gradient's column sums / dropout."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops

BF16 = torch.bfloat16
CAPS = (128, 256, 384, 512, 768, 2048)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print(f'LayerNorm backward, 768 columns, us per call (best of two mirrored runs) at MERLOT_LN_BWD_BLOCKS = {CAPS}', flush=True)
for name, rows in (('ViT (2 048 x 198)', 2048 * 198), ('joint (512 x 328)', 512 * 328), ('text-only (128 x 512)', 128 * 512)):
    for branch in (False, True):
        g = torch.Generator(device='cuda').manual_seed(0)
        x = torch.randn(rows, 768, generator=g, device='cuda').to(BF16)
        dy = torch.randn(rows, 768, generator=g, device='cuda').to(BF16)
        dres = torch.randn(rows, 768, generator=g, device='cuda').to(BF16)
        gamma = 1 + 0.1 * torch.randn(768, generator=g, device='cuda')
        mean, rstd = torch.zeros(rows, device='cuda'), torch.ones(rows, device='cuda')
        dga, dbe, bb = torch.zeros(768, device='cuda'), torch.zeros(768, device='cuda'), torch.zeros(768, device='cuda')
        kw = dict(dres=dres, branch_bias_grad=bb, drop_p=0.1, drop_seed=7) if branch else dict(dres=dres)
        best = {c: 1e30 for c in CAPS}
        for order in (CAPS, CAPS[::-1]):
            for c in order:
                os.environ['MERLOT_LN_BWD_BLOCKS'] = str(c)
                best[c] = min(best[c], timed(lambda: ops.ln_bwd(dy, x, mean, rstd, gamma, dga, dbe, **kw)))
        what = '+ branch gradient, dropout, column sums' if branch else 'dx = LN-backward(dy) + dres            '
        print(f'{name:22s} {what}: ' + ' '.join(f'{best[c]:7.1f}' for c in CAPS), flush=True)
