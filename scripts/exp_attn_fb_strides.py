"""Round 3: the fused attention backward and the resident forward with NON-default leading dimensions (the LDS-DMA source addresses are
built from them) and odd head counts: fused (MERLOT_ATTN_FB=1) against the dQ + dK/dV pair (=0), bit for bit; resident forward against
the tiled one, to rounding."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from merlot_amd.lib import call

torch.manual_seed(1)
bad = 0
for B, S, heads, masked, pads in ((3, 198, 12, False, (64, 8, 16, 8)), (2, 130, 5, True, (8, 24, 8, 16)), (2, 328, 3, True, (128, 8, 8, 8)),
                                  (1, 512, 2, True, (8, 8, 8, 8)), (5, 77, 7, False, (16, 40, 8, 24)), (2, 256, 1, True, (0, 0, 0, 0)),
                                  (3, 257, 4, False, (8, 8, 8, 8)), (2, 65, 12, True, (8, 8, 8, 8))):
    D = heads * 64
    ld, ldo, lddo, lddq = 3 * D + pads[0], D + pads[1], D + pads[2], 3 * D + pads[3]
    qkv_full = (torch.randn(B * S, ld, device='cuda') * 0.7).bfloat16()
    qkv = qkv_full[:, :3 * D] if pads[0] else qkv_full
    valid = None
    if masked:
        valid = (torch.rand(B, S, device='cuda') > 0.25).to(torch.uint8)
        valid[:, 0] = 1
        valid[0, S // 3:] = 0
    vp = valid.data_ptr() if masked else None
    outs = {}
    for res in ('0', '1'):
        os.environ['MERLOT_ATTN_RES'] = res
        os.environ['MERLOT_ATTN_RESFWD'] = res
        o_full = torch.full((B * S, ldo), float('nan'), device='cuda', dtype=torch.bfloat16)
        lse = torch.empty(B, heads, S, device='cuda')
        call('merlot_attention_fwd', qkv_full.data_ptr(), ld, o_full.data_ptr(), ldo, lse.data_ptr(), vp, None, B, S, heads, 0.125,
             None, None, S, 0, 1.0, *ops._attn_ws(), ops._stream())
        outs[res] = (o_full[:, :D].float().clone(), lse.clone(), o_full)
    d_o = float((outs['0'][0] - outs['1'][0]).abs().max())
    d_l = float((outs['0'][1] - outs['1'][1]).abs().max())
    pad_clean = bool(torch.isnan(outs['1'][2][:, D:].float()).all()) if pads[1] else True
    o_full, lse = outs['0'][2], outs['0'][1]
    do_full = torch.randn(B * S, lddo, device='cuda').bfloat16()
    if masked:
        do_full = do_full * valid.reshape(B * S, 1).to(do_full.dtype)
    res_b = {}
    for k in ('0', '1'):
        os.environ['MERLOT_ATTN_FB'] = k
        dqkv = torch.full((B * S, lddq), float('nan'), device='cuda', dtype=torch.bfloat16)
        delta = torch.full((B, heads, S), float('nan'), device='cuda')
        call('merlot_attention_bwd', qkv_full.data_ptr(), ld, o_full.data_ptr(), ldo, do_full.data_ptr(), lddo, lse.data_ptr(), vp, None,
             dqkv.data_ptr(), lddq, delta.data_ptr(), B, S, heads, 0.125, None, None, S, 1.0, *ops._attn_ws(), ops._stream())
        res_b[k] = (dqkv, delta)
    torch.cuda.synchronize()
    same = torch.equal(res_b['0'][0][:, :3 * D].view(torch.int16), res_b['1'][0][:, :3 * D].view(torch.int16))
    same_d = torch.equal(res_b['0'][1].view(torch.int32), res_b['1'][1].view(torch.int32))
    pad_b = bool(torch.isnan(res_b['1'][0][:, 3 * D:].float()).all()) if pads[3] else True
    ok = same and same_d and pad_b and pad_clean and d_o < 4e-3 and d_l < 1e-4
    bad += 0 if ok else 1
    print(f'B {B} S {S:3d} heads {heads:2d} masked {masked!s:5s} ld {ld} ldo {ldo} lddo {lddo} lddqkv {lddq}: backward bit-identical {same} delta {same_d} '
          f'pad columns untouched {pad_b and pad_clean} | forward resident vs tiled max|dO| {d_o:.1e} max|dlse| {d_l:.1e} | {"ok" if ok else "MISMATCH"}', flush=True)
print('ALL OK' if bad == 0 else f'{bad} MISMATCHES')
