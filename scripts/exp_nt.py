import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlot_amd import ops
from bench_kernels import timeit
BF16 = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: (torch.randn(s, device='cuda', generator=g) * 0.5).to(BF16)
cfgs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,4").split(',')]
for T in [50688, 20992, 8192]:
    for (N, K, name) in [(2304, 768, 'qkv'), (768, 768, 'proj'), (3072, 768, 'fc1'), (768, 3072, 'fc2')]:
        a, bt = rnd(T, K), rnd(N, K)
        bias = torch.randn(N, device='cuda')
        ref = None
        line = f"T={T:6d} {name:5s} N={N:5d} K={K:5d}:"
        for c in cfgs:
            os.environ['MERLOT_NT_CFG_DYN'] = str(c)
            out = ops.gemm_nt(a, bt, bias=bias).float()
            if ref is None:
                ref = (a[:2048].float() @ bt.float().t() + bias)
            err = float((out[:2048] - ref).norm() / ref.norm())
            t = timeit(lambda: ops.gemm_nt(a, bt, bias=bias), iters=10)
            line += f"  c{c}: {2*T*N*K/t/1e12:6.0f}TF{'' if err < 6e-3 else ' ERR%.3g' % err}"
        print(line, flush=True)
# epilogue + edge correctness on the ring configs
for c in cfgs:
    os.environ['MERLOT_NT_CFG_DYN'] = str(c)
    a, bt = rnd(1000, 768), rnd(770, 768)
    res = rnd(1000, 770 + 6)[:, :770]
    u = torch.empty((1000, 776), device='cuda', dtype=BF16)[:, :770]
    o1 = ops.gemm_nt(a, bt, epilogue=ops.EPI_GELU, aux_out=u).float()
    pre = a.float() @ bt.float().t()
    e1 = float((o1 - torch.nn.functional.gelu(pre)).norm() / pre.norm())
    e1u = float((u.float() - pre).norm() / pre.norm())
    o2 = ops.gemm_nt(a, bt, epilogue=ops.EPI_RESIDUAL, aux_in=res).float()
    e2 = float((o2 - (pre + res.float())).norm() / pre.norm())
    o3 = ops.gemm_nt(a, bt, out_dtype=torch.float32, alpha=0.5)
    e3 = float((o3 - 0.5 * pre).norm() / pre.norm())
    print(f"cfg {c}: gelu {e1:.2e} u {e1u:.2e} residual {e2:.2e} f32 {e3:.2e}")
