"""Epilogue time of ONE tile per CU with the rest of the chip idle (12 or 256 workgroups, one tile each)."""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from exp_epi import bench

dev = 'cuda'
os.environ['MERLOT_NT_CFG_DYN'] = '21'
for M, N, K in [(256, 3072, 768), (256 * 8, 3072, 768), (256 * 21, 3072, 768), (256 * 64, 3072, 768), (256, 3072, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.zeros(N, device=dev)
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).bfloat16()
    cases = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
             'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
             'residual': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1)}
    tiles = (M // 256) * (N // 256)
    for k, fn in cases.items():
        os.environ['MERLOT_DBG'] = '1'
        tl = bench(fn, 50)
        os.environ['MERLOT_DBG'] = '8'
        t8 = bench(fn, 50)
        os.environ['MERLOT_DBG'] = '0'
        t = bench(fn, 50)
        print(f'M={M:6d} N={N} K={K} tiles={tiles:4d} {k:9s}: loop {tl:6.1f} us, +epilogue without stores {t8:6.1f}, full {t:6.1f} '
              f'(stores +{t - t8:5.1f} us)', flush=True)
