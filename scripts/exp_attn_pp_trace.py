"""Round 5: per-wave phase timeline of the persistent attention backward (experiments build, MERLOT_ATTN_DBG=256): workgroup 0,
first 16 items; stamps: 0 top | 1 after barrier A | 2 after A' (delta) | 3 pass 2 done | 4 stores + lifts done | 5 after barrier B |
6 the own K / V row loads issued | 7 pass 1 done (the dQ stores end at the next 0)."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from merlot_amd.lib import LIB

LIB.protos['merlot_probe_attn_trace'] = ('int', [('void*', 'dst'), ('int64_t', 'bytes'), ('merlot_stream_t', 'stream')])
B, S = 2048, 198
qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
os.environ['MERLOT_ATTN_PP'] = '1'
o, lse = ops.attention_fwd(qkv, B, S, 12, None)
do = torch.randn_like(o)
for dbg in ('256', '288'):
    os.environ['MERLOT_ATTN_DBG'] = dbg
    for _ in range(3):
        ops.attention_bwd(qkv, o, do, lse, B, S, 12, None)
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 16 * 8, dtype=torch.int64, device='cuda')
    LIB.call('merlot_probe_attn_trace', buf.data_ptr(), buf.numel() * 8, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = buf.cpu().view(8, 16, 8)
    print(f'--- dbg {dbg} ({ {"256": "full", "768": "full, setprio 1 for waves 4-6", "288": "data only"}[dbg] }): cycles per phase (s_memtime ticks = 100 MHz? see ratio), items 4..11 averaged, per wave')
    # rotated loop: slot k holds [6, 7] = own-row loads of item k issued / end of pass 1 of item k - 1 (top of the iteration), then 0 .. 5 of item k
    names = ['wait A+bar', 'delta+bar', 'pass 2', 'store+lift', 'wait B+bar', 'own loads issue', 'pass 1', 'dQ store', 'item total']
    for w in range(7):
        a = t[w, 4:12].double()
        nxt = t[w, 5:13].double()
        d = [a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 4] - a[:, 3], a[:, 5] - a[:, 4], nxt[:, 6] - a[:, 5],
             nxt[:, 7] - nxt[:, 6], nxt[:, 0] - nxt[:, 7], nxt[:, 0] - a[:, 0]]
        print(f'wave {w}: ' + ' | '.join(f'{n} {float(x.mean()):8.0f}' for n, x in zip(names, d)))
os.environ['MERLOT_ATTN_DBG'] = '0'
