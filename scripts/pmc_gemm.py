import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlot_amd import ops
BF16 = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: (torch.randn(s, device='cuda', generator=g) * 0.5).to(BF16)
T = 101376
a, bt = rnd(T, 768), rnd(3072, 768)
a2, b2 = rnd(T, 3072), rnd(T, 768)
out = torch.zeros((3072, 768), device='cuda')
bias = torch.zeros(3072, device='cuda')
for c in [21, 11]:                       # production kernels: persistent dynamic-claims 256x256, ring 128x256
    os.environ['MERLOT_NT_CFG_DYN'] = str(c)
    for _ in range(3):
        ops.gemm_nt(a, bt, bias=bias)
for _ in range(3):
    ops.gemm_tn(a2, b2, out)
qkv = rnd(512 * 198, 2304)
o, lse = ops.attention_fwd(qkv, 512, 198, 12)
do = rnd(512 * 198, 768)
ops.attention_bwd(qkv, o, do, lse, 512, 198, 12)
torch.cuda.synchronize()
