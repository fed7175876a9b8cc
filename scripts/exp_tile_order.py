"""A/B of the persistent NT kernel's tile enumeration: row-major (MERLOT_NT_TILE_CG_DYN=0) vs column groups of 6 tile
columns (=6), mirrored order after a warm-up, results must be bit-identical (same tiles, different order)."""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from exp_epi import bench

dev = 'cuda'
T = 101376
os.environ['MERLOT_NT_CFG_DYN'] = '21'
for name, N, K, epi in [('qkv', 2304, 768, 'none'), ('fc1', 3072, 768, 'gelu'), ('dgrad_fc2', 3072, 768, 'dgelu'), ('ragged', 2304 + 64, 768, 'none')]:
    a = torch.randn(T if name != 'ragged' else 5000, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    aux = torch.empty(a.shape[0], N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(a.shape[0], N, device=dev).bfloat16()
    fn = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
          'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
          'dgelu': lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res)}[epi]
    os.environ['MERLOT_NT_TILE_CG_DYN'] = '0'
    bench(fn, 40)
    out, row = {}, []
    for cg in ('0', '6', '6', '0'):
        os.environ['MERLOT_NT_TILE_CG_DYN'] = cg
        out[cg] = fn().clone()
        row.append(f'cg {cg}: {bench(fn, 30):7.1f} us')
    print(f'{name:10s} [{a.shape[0]} x {N} x {K}] {epi:6s} ' + '  '.join(row) + f'  identical={torch.equal(out["0"], out["6"])}', flush=True)
    assert torch.equal(out['0'], out['6'])
