"""Round 6: the LayerNorm kernels with a grid-stride loop and the next row's operands requested ahead, against the library built before the change
(AB_LIB=<old .so>, same box, alternating processes): forward and backward at the three row counts of the bench step, bf16, H = 768, with the backward's
residual / dropout / bias-gradient tail as the transformer layers use it.  Prints us per launch and TB/s (algorithmic bytes: forward 2 tensors, backward 5)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_amd import lib  # noqa: E402
if os.environ.get('AB_LIB'):
    lib.LIB.path = os.path.abspath(os.environ['AB_LIB'])
    lib.LIB.check_abi = False
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def bench(fn, iters=30):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


torch.manual_seed(0)
row = []
for T in (405504, 167936, 65536):
    h = torch.randn(T, 768, device='cuda').to(BF16)
    dy = torch.randn(T, 768, device='cuda').to(BF16)
    dres = torch.randn(T, 768, device='cuda').to(BF16)
    g, b = torch.ones(768, device='cuda'), torch.zeros(768, device='cuda')
    dg, db, bg = torch.zeros(768, device='cuda'), torch.zeros(768, device='cuda'), torch.zeros(768, device='cuda')
    y, _, mean, rstd = ops.ln_fwd(h, g, b)
    tf = bench(lambda: ops.ln_fwd(h, g, b))
    tb = bench(lambda: ops.ln_bwd(dy, h, mean, rstd, g, dg, db, dres=dres, branch_bias_grad=bg, drop_p=0.1, drop_seed=7))
    row.append(f'T={T}: fwd {tf:6.1f} us {2 * T * 1536 / tf / 1e6:5.2f} TB/s | bwd {tb:6.1f} us {5 * T * 1536 / tb / 1e6:5.2f} TB/s')
    if T == 65536:                                          # same results from either build: checksums
        dx, dd = ops.ln_bwd(dy, h, mean, rstd, g, dg, db, dres=dres, branch_bias_grad=bg, drop_p=0.1, drop_seed=7)
        row.append(f'checksums y {float(y.float().sum()):.6f} {float(y.float().abs().sum()):.4f} dx {float(dx.float().sum()):.6f} {float(dx.float().abs().sum()):.4f} '
                   f'dbranch {float(dd.float().abs().sum()):.4f} mean {float(mean.sum()):.6f}')
print(f'{os.path.basename(lib.LIB.path or "libmerlot_hip.so"):24s} ' + ' || '.join(row), flush=True)
