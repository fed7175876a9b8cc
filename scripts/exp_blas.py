# calibration only (NOT used by the product): what the vendor GEMM reaches on the same shapes / data
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_kernels import timeit
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: (torch.randn(s, device='cuda', generator=g) * 0.5).to(torch.bfloat16)
for T in [101376, 50688, 8192]:
    for (N, K, name) in [(2304, 768, 'qkv'), (768, 768, 'proj'), (3072, 768, 'fc1'), (768, 3072, 'fc2')]:
        a, bt = rnd(T, K), rnd(N, K)
        t = timeit(lambda: torch.mm(a, bt.t()), iters=10)
        # wgrad form
        b2 = rnd(T, N)
        t2 = timeit(lambda: torch.mm(a.t(), b2), iters=10)
        print(f"T={T:6d} {name:5s}: NT {2*T*N*K/t/1e12:6.0f} TF   TN(wgrad [K,N]) {2*T*N*K/t2/1e12:6.0f} TF")
