"""Same-box A/B of two builds of the PRODUCT library on the step's NT GEMM shapes: AB_LIB=<path to .so> python scripts/ab_lib.py
(boxes of the pool differ by +-1.5 %, so a 1-3 % kernel change only shows inside one call)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_amd import lib  # noqa: E402
if os.environ.get('AB_LIB'):
    lib.LIB.path = os.path.abspath(os.environ['AB_LIB'])
    lib.LIB.check_abi = False             # an older build: only entry points whose signature did not change are called here
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402

dev = 'cuda'
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


tot = 0.0
row = []
for name, N, K, epi in [('qkv', 2304, 768, 'none'), ('proj', 768, 768, 'residual'), ('fc1', 3072, 768, 'gelu'), ('fc2', 768, 3072, 'residual'),
                        ('dgrad_fc1', 768, 3072, 'none'), ('dgrad_fc2', 3072, 768, 'dgelu'), ('dgrad_proj', 768, 768, 'none'), ('dgrad_qkv', 768, 2304, 'none')]:
    a = torch.randn(T, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(T, N, device=dev).bfloat16()
    fn = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
          'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
          'residual': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1),
          'dgelu': lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res)}[epi]
    t = bench(fn)
    tot += t
    row.append(f'{name} {t:6.1f}')
print(f'{os.path.basename(lib.LIB.path or "libmerlot_hip.so"):24s} sum {tot:7.1f} us | ' + ' | '.join(row), flush=True)
