"""Per-workgroup timeline of the persistent NT GEMM (MERLOT_DBG bit 512): when do the epilogues happen?"""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from merlot_amd.lib import LIB

dev = 'cuda'
T, N, K = 101376, 3072, 768
a = torch.randn(T, K, device=dev).bfloat16()
b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
bias = torch.zeros(N, device=dev)
aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
os.environ['MERLOT_NT_CFG_DYN'] = '21'
TR = 32
for name, dbg in (('timeline', 512),):
    for epi in ('none', 'gelu'):
        fn = (lambda: ops.gemm_nt(a, b, bias=bias)) if epi == 'none' else \
            (lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux))
        os.environ['MERLOT_DBG'] = '0'
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        os.environ['MERLOT_DBG'] = str(dbg)
        fn()
        torch.cuda.synchronize()
        buf = torch.zeros(256 * TR * 4, dtype=torch.int64, device=dev)
        LIB.call('merlot_probe_persist_trace', buf.data_ptr(), buf.numel() * 8, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        tr = buf.cpu().numpy().reshape(256, TR, 4)
        t0 = tr[:, 0, 0].min()
        ntile = (tr[:, :, 2] > 0).sum(1)
        loop = (tr[:, :, 1] - tr[:, :, 0])[tr[:, :, 2] > 0]
        epi_t = (tr[:, :, 2] - tr[:, :, 1])[tr[:, :, 2] > 0]
        end = tr[:, :, 2].max() - t0
        print(f'{name:10s} {epi:5s}: tiles/WG min {ntile.min()} max {ntile.max()} | loop cycles mean {loop.mean():8.0f} p10 {np.percentile(loop, 10):8.0f} '
              f'p90 {np.percentile(loop, 90):8.0f} | epilogue cycles mean {epi_t.mean():8.0f} p10 {np.percentile(epi_t, 10):8.0f} p90 '
              f'{np.percentile(epi_t, 90):8.0f} | span {end} cycles')
        # epilogue concurrency: how many WGs are inside an epilogue at each sampled instant
        ts = np.linspace(0, end, 2000)
        starts = (tr[:, :, 1] - t0)[tr[:, :, 2] > 0]
        ends = (tr[:, :, 2] - t0)[tr[:, :, 2] > 0]
        conc = np.array([((starts <= t) & (ends > t)).sum() for t in ts])
        print(f'             WGs in epilogue simultaneously: mean {conc.mean():6.1f} p50 {np.percentile(conc, 50):5.0f} p90 '
              f'{np.percentile(conc, 90):5.0f} max {conc.max()}  (first-tile start spread {np.ptp(tr[:, 0, 0])} cycles)')
        np.save(f'/tmp/trace_{name}_{epi}.npy', tr)
