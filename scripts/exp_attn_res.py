"""Round 3: the resident forward (attention_res.inc) after its loads moved to LDS-DMA and its stores to 16 B: (a) as the plain
forward of the ViT pass against the tiled kernel (MERLOT_ATTN_RES=1, experiments build), (b) with the side outputs at the joint /
text-only shapes (compare with the averages of profiles/r03_z_kernel_stats.csv: 754 us over both)."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from exp_attn_time import timeit

SC = int(os.environ.get('SCALE', 4))
for B, S, masked in ((512 * SC, 198, False), (64, 100, False), (64, 256, False), (128 * SC, 328, True)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = None
    if masked:
        valid = (torch.rand(B, S, device='cuda') > 0.2).to(torch.uint8)
        valid[:, 0] = 1
    outs = {}
    for k in ('0', '1'):
        os.environ['MERLOT_ATTN_RES'] = k
        outs[k] = ops.attention_fwd(qkv, B, S, 12, valid)
    d = float((outs['0'][0].float() - outs['1'][0].float()).abs().max())
    dl = float((outs['0'][1] - outs['1'][1]).abs().max())
    row = []
    for k in ('0', '1', '1', '0'):
        os.environ['MERLOT_ATTN_RES'] = k
        t = timeit(lambda: ops.attention_fwd(qkv, B, S, 12, valid))
        row.append(f'{ {"0": "tiled", "1": "resident"}[k] } {t:7.1f} us')
    print(f'fwd B {B:5d} S {S:4d} masked {masked!s:5s}: max|dO| {d:.2e} max|dlse| {dl:.2e} | ' + ' | '.join(row), flush=True)
os.environ['MERLOT_ATTN_RES'] = '0'
# (b) side outputs from the forward launch
for B, S, joint in ((128 * SC, 328, True), (32 * SC, 512, False)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = (torch.rand(B, S, device='cuda') > 0.1).to(torch.uint8)
    valid[:, 0] = 1
    lo = torch.zeros(B, S, device='cuda')
    hi = torch.zeros(B, S, device='cuda')
    if joint:
        f = lambda: ops.attention_fwd(qkv, B, S, 12, valid, colsum_lo=lo, colsum_hi=hi, qsplit=200, valid_q_only=True, weight=1.0 / 12)
    else:
        f = lambda: ops.attention_fwd(qkv, B, S, 12, valid, colsum_lo=lo, valid_q_only=False, weight=1.0 / 12)
    t = timeit(f)
    print(f'fwd + side outputs B {B:5d} S {S:4d}: {t:7.1f} us', flush=True)

# (c) where the side-output launches spend their time (MERLOT_ATTN_DBG: 8 = no side-output pass, 16 = no main pass, 24 = neither)
for B, S, joint in ((128 * SC, 328, True), (32 * SC, 512, False)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = (torch.rand(B, S, device='cuda') > 0.1).to(torch.uint8)
    valid[:, 0] = 1
    lo = torch.zeros(B, S, device='cuda')
    hi = torch.zeros(B, S, device='cuda')
    if joint:
        f = lambda: ops.attention_fwd(qkv, B, S, 12, valid, colsum_lo=lo, colsum_hi=hi, qsplit=200, valid_q_only=True, weight=1.0 / 12)
    else:
        f = lambda: ops.attention_fwd(qkv, B, S, 12, valid, colsum_lo=lo, valid_q_only=False, weight=1.0 / 12)
    row = []
    for dbg in ('0', '8', '16', '24', '0'):
        os.environ['MERLOT_ATTN_DBG'] = dbg
        row.append(f'dbg {dbg}: {timeit(f):7.1f} us')
    os.environ['MERLOT_ATTN_DBG'] = '0'
    print(f'side-output forward B {B:5d} S {S:4d}: ' + ' | '.join(row), flush=True)
