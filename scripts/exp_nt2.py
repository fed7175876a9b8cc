import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlot_amd import ops
from bench_kernels import timeit
BF16 = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: (torch.randn(s, device='cuda', generator=g) * 0.5).to(BF16)
T, N = 50688, 768
for c in [int(x) for x in sys.argv[1].split(',')]:
    os.environ['MERLOT_NT_CFG_DYN'] = str(c)
    for K in [768, 3072]:
        a, bt = rnd(T, K), rnd(N, K)
        line = f"cfg{c} T={T} N={N} K={K:5d}:"
        for dbg in [0, 1, 5]:
            os.environ['MERLOT_DBG'] = str(dbg)
            t = timeit(lambda: ops.gemm_nt(a, bt), iters=10)
            line += f"  dbg{dbg}: {t*1e6:7.1f}us"
        print(line, flush=True)
