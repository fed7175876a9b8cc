import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlot_amd import ops
from bench_kernels import timeit
BF16 = torch.bfloat16
T = 50688
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: (torch.randn(s, device='cuda', generator=g) * 0.5).to(BF16)
for T in [101376, 50688, 20992]:
  for (M, N, name) in [(768, 768, 'proj'), (3072, 768, 'fc1'), (768, 3072, 'fc2'), (2304, 768, 'qkv')]:
    a, b = rnd(T, M), rnd(T, N)
    out = torch.zeros((M, N), device='cuda')
    line = f"T={T} {name}:"
    for cfg in [0, 1]:
        os.environ['MERLOT_TN_CFG'] = str(cfg)
        t = timeit(lambda: ops.gemm_tn(a, b, out, accumulate=True), iters=10)
        line += f"  cfg{cfg}: {t*1e6:8.1f} us {2*T*M*N/t/1e12:6.0f} TF"
    print(line)
