"""Ablation of the ping-pong NT kernel's main loop (MERLOT_DBG bits, experiments build only; results are garbage, only the
time counts): 1 = no epilogue, +4 = no LDS-DMA, +16 = no fragment reads, +64 = no MFMAs."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from exp_epi import bench

T = int(os.environ.get('T', 101376))
torch.manual_seed(0)
for name, N, K in [('qkv', 2304, 768), ('fc2', 768, 3072)]:
    a = torch.randn(T, K, device='cuda').bfloat16()
    b = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
    fn = lambda: ops.gemm_nt(a, b)
    fl = 2.0 * T * N * K
    os.environ['MERLOT_NT_CFG_DYN'] = '22'
    os.environ['MERLOT_DBG'] = '1'
    bench(fn, 60)
    row = []
    for rep in range(2):
        for dbg, lab in [(1, 'loop'), (5, 'no DMA'), (17, 'no reads'), (21, 'no DMA, no reads'), (65, 'no MFMA'), (85, 'barriers only')]:
            os.environ['MERLOT_DBG'] = str(dbg)
            t = bench(fn, 30)
            row.append(f'{lab}: {t:6.1f} us ({fl / t / 1e6:5.0f} TF-equiv)')
    os.environ['MERLOT_DBG'] = '0'
    print(f'{name} [{T} x {N} x {K}]  ' + ' | '.join(row), flush=True)
