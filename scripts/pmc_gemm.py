"""Launches for the PMC passes of scripts/gpu_traffic.sh: the PRODUCT library (libmerlot_hip.so, the binary bench.py runs),
one representative shape per dominant kernel.  No timing here -- rocprofv3 --pmc serialises the kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402

BF16 = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: (torch.randn(s, device='cuda', generator=g) * 0.5).to(BF16)  # noqa: E731
T = 405504                                    # ViT rows of one bench step (128 examples x 16 frames x 198 tokens)
a, w = rnd(T, 768), rnd(3072, 768)
bias = torch.zeros(3072, device='cuda')
for _ in range(3):
    ops.gemm_nt(a, w, bias=bias)              # gemm_nt_p8_kernel<0, false, false, true>: the launch bench.py's roofline.traffic quotes
u = torch.empty(T, 3072, device='cuda', dtype=BF16)
for _ in range(3):
    ops.gemm_nt(a, w, bias=bias, epilogue=ops.EPI_GELU, aux_out=u)          # <1, false, false>: fc1
a8, sa = ops.quantize_e4m3(a)
w8, sw = ops.quantize_e4m3(w)
for _ in range(3):
    ops.gemm_fp8_nt(a8, sa, w8, sw, bias=bias)                              # <0, false, true>
dy, x = rnd(T, 3072), rnd(T, 768)
gw = torch.zeros((3072, 768), device='cuda')
for _ in range(3):
    ops.gemm_tn(dy, x, gw)                                                  # gemm_tn_p1_kernel + tn_reduce_kernel
qkv = rnd(2048 * 198, 2304)
for _ in range(2):
    o, lse = ops.attention_fwd(qkv, 2048, 198, 12)
    do = rnd(2048 * 198, 768)
    ops.attention_bwd(qkv, o, do, lse, 2048, 198, 12)
torch.cuda.synchronize()
