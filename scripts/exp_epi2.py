"""NT GEMM configs (21 persistent-dynamic 256x256, 11 = 128x256 2 WG/CU, 3 = 256x256 1 WG/CU) per epilogue type."""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from exp_epi import bench  # noqa

dev = 'cuda'
T = 101376
cfgs = [int(c) for c in os.environ.get('CFGS', '21,11,3').split(',')]
for N, K, name in [(768, 768, 'proj'), (768, 3072, 'fc2'), (3072, 768, 'fc1/dfc2'), (2304, 768, 'qkv')]:
    a = torch.randn(T, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.zeros(N, device=dev)
    res = torch.randn(T, N, device=dev).bfloat16()
    aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    flops = 2.0 * T * N * K
    cases = {
        'none': lambda: ops.gemm_nt(a, b, bias=bias),
        'gelu(+preact out)': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
        'residual p=0.1': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1,
                                              dropout_seed=123),
        'dgelu': lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res),
    }
    print(f'{name:9s} N={N} K={K}', flush=True)
    for k, fn in cases.items():
        row = []
        for c in cfgs:
            os.environ['MERLOT_NT_CFG_DYN'] = str(c)
            os.environ['MERLOT_DBG'] = '1'
            tl = bench(fn)
            os.environ['MERLOT_DBG'] = os.environ.get('BASE_DBG', '0')
            t = bench(fn)
            extra = ''
            if os.environ.get('DECOMP'):
                ts = {}
                for d in (8, 128, 136):
                    os.environ['MERLOT_DBG'] = str(d)
                    ts[d] = bench(fn)
                os.environ['MERLOT_DBG'] = '0'
                extra = f' no-store {ts[8]:6.1f} no-math {ts[128]:6.1f} neither {ts[136]:6.1f}'
            row.append(f'cfg{c}: {t:7.1f} us {flops / t / 1e6:5.0f} TF (loop {tl:6.1f}){extra}')
        print(f'    {k:20s} ' + ' | '.join(row), flush=True)
