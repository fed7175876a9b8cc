"""Round 3: the fused short-sequence attention backward (attention_fb.inc, MERLOT_ATTN_FB=1, experiments build) against the
dQ + dK/dV kernel pair (=0): outputs compared bit for bit (dqkv and the published delta), then timing in mirrored order."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from merlot_amd.lib import call
from exp_attn_time import timeit

_p = ops._p


def bwd(qkv, o, do, lse, B, S, valid):
    dqkv = torch.full_like(qkv, float('nan'))
    delta = torch.full((B, 12, S), float('nan'), device='cuda')
    call('merlot_attention_bwd', _p(qkv), qkv.stride(0), _p(o), o.stride(0), _p(do), do.stride(0), _p(lse), _p(valid), None, _p(dqkv),
         dqkv.stride(0), _p(delta), B, S, 12, 0.125, None, None, S, 1.0, *ops._attn_ws(), ops._stream())
    return dqkv, delta


SC = int(os.environ.get('SCALE', 4))
CASES = [(64, 65, False), (64, 100, False), (64, 198, False), (64, 225, False), (64, 256, False), (64, 257, False), (32, 400, False), (32, 512, False),
         (64, 70, True), (64, 148, True), (64, 198, True), (64, 256, True), (32, 300, True), (32, 328, True), (32, 512, True),
         (512 * SC, 198, False), (128 * SC, 328, True), (32 * SC, 512, True)]
for B, S, masked in CASES:
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = None
    if masked:
        valid = (torch.rand(B, S, device='cuda') > 0.2).to(torch.uint8)
        valid[:, 0] = 1
        valid[0, S // 2:] = 0
    os.environ['MERLOT_ATTN_FB'] = '0'
    o, lse = ops.attention_fwd(qkv, B, S, 12, valid)
    do = torch.randn_like(o)
    if masked and B > 40:                                  # the model's padded rows carry zero upstream gradients; keep some cases without
        do = do * valid.reshape(B * S, 1).to(do.dtype)
    outs = {}
    for k in ('0', '1'):
        os.environ['MERLOT_ATTN_FB'] = k
        outs[k] = bwd(qkv, o, do, lse, B, S, valid)
    torch.cuda.synchronize()
    same = torch.equal(outs['0'][0].view(torch.int16), outs['1'][0].view(torch.int16))
    same_d = torch.equal(outs['0'][1].view(torch.int32), outs['1'][1].view(torch.int32))
    nan = bool(torch.isnan(outs['1'][0].float()).any())
    err = []
    if not same:
        for name, sl in (('dq', slice(0, 768)), ('dk', slice(768, 1536)), ('dv', slice(1536, 2304))):
            a, r = outs['1'][0][:, sl].float(), outs['0'][0][:, sl].float()
            err.append(f'{name} {float((a - r).norm() / r.norm()):.1e} ({int((a != r).sum())} differ)')
    row = []
    for k in ('0', '1', '1', '0'):
        os.environ['MERLOT_ATTN_FB'] = k
        t = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid))
        row.append(f'{ {"0": "dQ + dK/dV", "1": "fused"}[k] } {t:7.1f} us')
    print(f'bwd B {B:5d} S {S:4d} masked {masked!s:5s}: dqkv bit-identical {same} delta bit-identical {same_d} nan {nan} {" ".join(err)} | ' + ' | '.join(row), flush=True)

# ---- where the time goes (MERLOT_ATTN_DBG: bit 0 = no tile arithmetic, bit 2 = no Q | dO refill)
B, S = 512 * SC, 198
qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
o, lse = ops.attention_fwd(qkv, B, S, 12)
do = torch.randn_like(o)
os.environ['MERLOT_ATTN_FB'] = '1'
for dbg in ('0', '1', '4', '5', '0'):
    os.environ['MERLOT_ATTN_DBG'] = dbg
    t = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12))
    print(f'fused B {B} S {S} dbg {dbg}: {t:7.1f} us', flush=True)
os.environ['MERLOT_ATTN_DBG'] = '0'
