"""Does de-phasing the persistent workgroups (so that the tile epilogues of different CUs do not hit HBM at the same instant)
shorten the store-bound epilogues?  MERLOT_P8_DEPHASE = delay unit in shader cycles: workgroup i of an XCD starts
(i & 7) * unit late (experiments build only)."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from exp_epi import bench

T = int(os.environ.get('T', 101376))
torch.manual_seed(0)
os.environ['MERLOT_NT_CFG_DYN'] = '22'
for name, N, K, epi in [('qkv', 2304, 768, 'none'), ('proj', 768, 768, 'residual'), ('fc1', 3072, 768, 'gelu'), ('fc2', 768, 3072, 'residual'), ('dgrad_fc2', 3072, 768, 'dgelu')]:
    a = torch.randn(T, K, device='cuda').bfloat16()
    b = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
    bias = torch.randn(N, device='cuda') * 0.1
    aux = torch.empty(T, N, device='cuda', dtype=torch.bfloat16)
    res = torch.randn(T, N, device='cuda').bfloat16()
    fn = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
          'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
          'residual': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1),
          'dgelu': lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res)}[epi]
    os.environ['MERLOT_P8_DEPHASE'] = '0'
    bench(fn, 40)
    row = []
    for unit in (0, 2000, 4000, 8000, 16000, 0):
        os.environ['MERLOT_P8_DEPHASE'] = str(unit)
        t = bench(fn, 30)
        row.append(f'{unit}: {t:6.1f} us')
    os.environ['MERLOT_P8_DEPHASE'] = '0'
    print(f'{name:10s} [{T} x {N} x {K}] {epi:8s} ' + ' | '.join(row), flush=True)
