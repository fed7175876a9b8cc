"""Round 3: the persistent streaming attention kernels (MERLOT_ATTN_PS=1, experiments build) against the one-shot kernels (=0) at
the step's shapes: outputs compared (same arithmetic, different chunking of the online softmax: rounding-level differences), then
timing in mirrored order."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from exp_attn_time import timeit

SC = int(os.environ.get('SCALE', 4))                     # 4: the bench batch (2048 frames); 1: 512 frames
for B, S, masked in (() if os.environ.get('SKIP_FWD') else ((512 * SC, 198, False), (128 * SC, 328, True), (32 * SC, 512, True), (512 * SC, 198, True), (64, 130, True))):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = None
    if masked:
        valid = (torch.rand(B, S, device='cuda') > 0.2).to(torch.uint8)
        valid[:, 0] = 1
        valid[0, S // 2:] = 0
    outs = {}
    for k in ('0', '1', '2'):
        os.environ['MERLOT_ATTN_PS'] = k
        outs[k] = ops.attention_fwd(qkv, B, S, 12, valid)
    d = max(float((outs['0'][0].float() - outs[k][0].float()).abs().max()) for k in ('1', '2'))
    dl = max(float((outs['0'][1] - outs[k][1]).abs().max()) for k in ('1', '2'))
    row = []
    for k in ('0', '1', '2', '2', '1', '0'):
        os.environ['MERLOT_ATTN_PS'] = k
        t = timeit(lambda: ops.attention_fwd(qkv, B, S, 12, valid))
        row.append(f'{ {"0": "one-shot", "1": "ps 128 rows", "2": "ps 256 rows"}[k] } {t:7.1f} us')
    gb = B * S * 768 * 2 * 4 / 1e9
    print(f'fwd B {B:5d} S {S:4d} masked {masked!s:5s}: max|dO| {d:.2e} max|dlse| {dl:.2e} | ' + ' | '.join(row) + f' | {gb:.2f} GB algorithmic', flush=True)

# ---- backward: MERLOT_ATTN_PS_BWD bit 0 = persistent streaming dQ kernel, bit 1 = dK / dV kernel
for B, S, masked in ((512 * SC, 198, False), (128 * SC, 328, True), (32 * SC, 512, True), (64, 130, True)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = None
    if masked:
        valid = (torch.rand(B, S, device='cuda') > 0.2).to(torch.uint8)
        valid[:, 0] = 1
        valid[0, S // 2:] = 0
    os.environ['MERLOT_ATTN_PS'] = '0'
    o, lse = ops.attention_fwd(qkv, B, S, 12, valid)
    do = torch.randn_like(o)
    if masked:
        do = do * valid.reshape(B * S, 1).to(do.dtype)
    outs = {}
    modes = os.environ.get('BWD_MODES', '0,1').split(',')
    for k in modes:
        os.environ['MERLOT_ATTN_PS_BWD'] = k
        outs[k] = ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid).float()
    errs = []
    for k in modes[1:]:
        for name, sl in (('dq', slice(0, 768)), ('dk', slice(768, 1536)), ('dv', slice(1536, 2304))):
            a, r = outs[k][:, sl], outs['0'][:, sl]
            errs.append(f'{k}:{name} {float((a - r).norm() / r.norm()):.1e}')
    row = []
    for k in modes + modes[::-1]:
        os.environ['MERLOT_ATTN_PS_BWD'] = k
        t = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid))
        row.append(f'mode {k} {t:7.1f} us')
    print(f'bwd B {B:5d} S {S:4d} masked {masked!s:5s}: rel-L2 vs one-shot [{" ".join(errs)}] | ' + ' | '.join(row), flush=True)
