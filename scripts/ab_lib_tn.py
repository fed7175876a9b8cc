"""Same-box A/B of two builds of the PRODUCT library on the step's weight-gradient (TN) shapes: AB_LIB=<path to .so> python scripts/ab_lib_tn.py"""
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

T = int(os.environ.get('T', 101376))
torch.manual_seed(0)


def bench(fn, iters=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


if __name__ == '__main__':
    row, tot = [], 0.0
    for (M, N, name) in [(768, 768, 'dWproj'), (3072, 768, 'dW1'), (768, 3072, 'dW2'), (2304, 768, 'dWqkv')]:
        a = torch.randn(T, M, device='cuda').bfloat16()
        b = torch.randn(T, N, device='cuda').bfloat16()
        out = torch.zeros((M, N), device='cuda')
        t = bench(lambda: ops.gemm_tn(a, b, out, accumulate=True))
        tot += t
        row.append(f'{name} {t:6.1f} us {2.0 * T * M * N / t / 1e6:5.0f} TF')
    print(f'{os.path.basename(lib.LIB.path or "libmerlot_hip.so"):24s} sum {tot:7.1f} us | ' + ' | '.join(row), flush=True)
