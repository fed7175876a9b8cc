"""Round 3: how long do the persistent streaming attention kernels take when they only MOVE the data (MERLOT_ATTN_DBG=1: LDS-DMA stream,
stationary steps, barriers and stores as usual, no tile arithmetic)?  Separates 'waiting for data' from 'computing'."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from exp_attn_time import timeit

SC = int(os.environ.get('SCALE', 4))
for B, S, masked in ((512 * SC, 198, False), (128 * SC, 328, True)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = (torch.rand(B, S, device='cuda') > 0.2).to(torch.uint8) if masked else None
    os.environ['MERLOT_ATTN_DBG'] = '0'
    os.environ['MERLOT_ATTN_PS'] = '0'
    os.environ['MERLOT_ATTN_PS_BWD'] = '0'
    o, lse = ops.attention_fwd(qkv, B, S, 12, valid)
    do = torch.randn_like(o)
    row = []
    for ps in ('1', '2'):
        for dbg in ('0', '1'):
            os.environ['MERLOT_ATTN_PS'], os.environ['MERLOT_ATTN_DBG'] = ps, dbg
            row.append(f'fwd {"128-row items" if ps == "1" else "256-row items"} {"data only" if dbg == "1" else "full"} {timeit(lambda: ops.attention_fwd(qkv, B, S, 12, valid)):7.1f}')
    os.environ['MERLOT_ATTN_PS'] = '0'
    for dbg in ('0', '1'):
        os.environ['MERLOT_ATTN_PS_BWD'], os.environ['MERLOT_ATTN_DBG'] = '1', dbg
        row.append(f'bwd (ps dq + one-shot dkdv) {"data only" if dbg == "1" else "full"} {timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid)):7.1f}')
    print(f'B {B} S {S} masked {masked}: ' + ' | '.join(row) + ' us', flush=True)
