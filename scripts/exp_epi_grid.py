"""Is the NT epilogue bound per CU or by a shared resource (HBM / fabric)?  The round-1 persistent kernel (id 21, same
epilogue code as the ping-pong kernel) with the main loop skipped (MERLOT_DBG=2) on 256 / 128 / 64 / 32 workgroups
(MERLOT_NT_GRID): per-workgroup time per tile.  If a tile's epilogue gets faster when fewer CUs write at the same time,
the bound is shared, and workgroups that are in their epilogues at DIFFERENT times would all be faster."""
import _exp_lib  # noqa: F401
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from exp_epi import bench

dev = 'cuda'
T = 101376
torch.manual_seed(0)
os.environ['MERLOT_NT_CFG_DYN'] = '21'
for name, N, K, epi in [('qkv', 2304, 768, 'none'), ('fc1', 3072, 768, 'gelu'), ('fc2', 768, 3072, 'residual'), ('dgrad_fc2', 3072, 768, 'dgelu')]:
    a = torch.randn(T, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(T, N, device=dev).bfloat16()
    fn = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
          'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
          'residual': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1),
          'dgelu': lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res)}[epi]
    tiles = ((T + 255) // 256) * ((N + 255) // 256)
    row = []
    for dbg, label in (('2', 'epilogue only'), ('10', 'epilogue only, no memory ops')):
        os.environ['MERLOT_DBG'] = dbg
        for grid in (256, 128, 64, 32):
            os.environ['MERLOT_NT_GRID'] = str(grid)
            t = bench(fn, 10)
            row.append(f'{label} grid {grid:3d}: {t:7.1f} us = {t * grid / tiles:5.2f} us/tile/WG')
    os.environ['MERLOT_DBG'] = '0'
    os.environ['MERLOT_NT_GRID'] = '256'
    print(f'{name:10s} [{T} x {N} x {K}] {epi:8s}\n    ' + '\n    '.join(row), flush=True)
