"""Tile enumeration of the ping-pong NT kernel: row-major (cg 0) vs groups of cg tile columns (MERLOT_P8_CG, experiments
build).  profiles/r02_traffic.txt shows the row-major order fetching ~6x the algorithmic operand bytes through the fabric
(the 12-column weight panel, 4.7 MB, does not stay in a 4 MB L2); does keeping an XCD inside a column group pay in TIME?
Mirrored order, outputs must be bit-identical."""
import _exp_lib  # noqa: F401
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from exp_epi import bench

dev = 'cuda'
T = int(os.environ.get('T', 101376))
CGS = os.environ.get('CGS', '0,2,3,4,6,0').split(',')
torch.manual_seed(0)
os.environ['MERLOT_NT_CFG_DYN'] = '22'
for name, N, K, epi in [('qkv', 2304, 768, 'none'), ('proj', 768, 768, 'residual'), ('fc1', 3072, 768, 'gelu'), ('fc2', 768, 3072, 'residual'),
                        ('dgrad_fc1', 768, 3072, 'none'), ('dgrad_fc2', 3072, 768, 'dgelu')]:
    a = torch.randn(T, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(T, N, device=dev).bfloat16()
    fn = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
          'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
          'residual': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1),
          'dgelu': lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res)}[epi]
    os.environ['MERLOT_P8_CG'] = '0'
    os.environ['MERLOT_DBG'] = '0'
    bench(fn, 60)
    ref, row = None, []
    for cg in CGS:
        os.environ['MERLOT_P8_CG'] = cg
        o = fn().clone()
        ref = o if ref is None else ref
        assert torch.equal(o, ref), (name, cg)
        t = bench(fn, 30)
        os.environ['MERLOT_DBG'] = '1'
        tl = bench(fn, 30)
        os.environ['MERLOT_DBG'] = '0'
        row.append(f'cg {cg}: {t:6.1f} / loop {tl:6.1f}')
    print(f'{name:10s} [{T} x {N} x {K}] {epi:8s} us  ' + '   '.join(row), flush=True)
