"""Round 5: would hiding the GELU / GELU' arithmetic of the fc1 / GELU'-dgrad epilogues under the next tile's main loop pay?
Upper bound, measured: the same launches with the polynomial switched OFF (MERLOT_DBG=128: identical loads and stores, no GELU arithmetic),
with the stores off (8: arithmetic + LDS staging only) and with the epilogue off (1: main loop only).  Experiments build."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from exp_epi import bench

for T in (101376, 405504):
    a = torch.randn(T, 768, device='cuda').bfloat16()
    w1 = (torch.randn(3072, 768, device='cuda') * 0.02).bfloat16()
    bias1 = torch.zeros(3072, device='cuda')
    u = torch.empty(T, 3072, device='cuda', dtype=torch.bfloat16)
    da = torch.randn(T, 768, device='cuda').bfloat16()          # dgrad of fc2: dA = dY W2 -> [T, 3072], times GELU'(u)
    w2t = (torch.randn(3072, 768, device='cuda') * 0.02).bfloat16()
    colsum = torch.zeros(3072, device='cuda')
    uu = torch.randn(T, 3072, device='cuda').bfloat16()
    cases = {'fc1 + GELU (writes u and a)': lambda: ops.gemm_nt(a, w1, bias=bias1, epilogue=ops.EPI_GELU, aux_out=u),
             "GELU' dgrad (reads u, writes dU)": lambda: ops.gemm_nt(da, w2t, epilogue=ops.EPI_DGELU, aux_in=uu, colsum_out=colsum)}
    flops = 2.0 * T * 3072 * 768
    for name, fn in cases.items():
        row = []
        for dbg, what in (('0', 'full'), ('128', 'no polynomial'), ('8', 'no stores'), ('1', 'main loop only'), ('0', 'full'), ('128', 'no polynomial')):
            os.environ['MERLOT_DBG'] = dbg
            t = bench(fn, 20)
            row.append(f'{what} {t:7.1f} us ({flops / t * 1e-6 / 2500:.3f})')
        os.environ['MERLOT_DBG'] = '0'
        print(f'T {T:6d} {name}: ' + ' | '.join(row), flush=True)
