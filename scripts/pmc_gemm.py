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
del qkv, o, do
# round 6: the kernels round 5's file had no rows for -- the joint encoder's masked attention (512 sequences of 328: training forward
# attn_fwd_ppm_kernel, backward attn_bwd_fused_kernel<512, true, true> with the attention log), the text-only pass (128 sequences of 512: forward with
# the column sums attn_fwd_res_kernel<true, 16>, backward attn_bwd_fused_kernel<512, true, false>) and the two LayerNorm kernels at the ViT's row count
for (B, S, side) in ((512, 328, 'log'), (128, 512, 'colsum')):
    qkv = rnd(B * S, 2304)
    valid = torch.ones(B, S, device='cuda', dtype=torch.uint8)
    valid[:, S - 7:] = 0                                                    # a few padded keys per sequence, as the captions have
    for _ in range(2):
        if side == 'colsum':
            cs = torch.zeros(B, S, device='cuda')
            o, lse = ops.attention_fwd(qkv, B, S, 12, valid=valid, colsum_lo=cs, valid_q_only=True)
            do = rnd(B * S, 768)
            ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid=valid)
        else:
            o, lse = ops.attention_fwd(qkv, B, S, 12, valid=valid)
            do = rnd(B * S, 768)
            lo, hi = torch.zeros(B, S, device='cuda'), torch.zeros(B, S, device='cuda')
            ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid=valid, log_lo=lo, log_hi=hi, log_split=200, log_weight=1.0 / 12)
    del qkv, o, do
h = rnd(T, 768)
gam, bet = torch.ones(768, device='cuda'), torch.zeros(768, device='cuda')
dgam, dbet = torch.zeros(768, device='cuda'), torch.zeros(768, device='cuda')
for _ in range(2):
    y16, _, mean, rstd = ops.ln_fwd(h, gam, bet)
    ops.ln_bwd(y16, h, mean, rstd, gam, dgam, dbet, dres=h)
torch.cuda.synchronize()
