"""Per-tile timeline of the ping-pong NT kernel (MERLOT_DBG=512, experiments build): thread 0 of every workgroup stamps
s_memtime at tile start / K-loop end / after the un-stagger barrier / after the epilogue issued its stores / after the
vmcnt(0) that waits for them / after the closing barrier.  Where do the 6-15 us per tile between two main loops go, and
how many workgroups of an XCD are inside their epilogue at the same time?"""
import _exp_lib  # noqa: F401
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from merlot_amd.lib import LIB

dev = 'cuda'
T = int(os.environ.get('T', 101376))
torch.manual_seed(0)
os.environ['MERLOT_NT_CFG_DYN'] = '22'
TT = 32
for name, N, K, epi in [('qkv', 2304, 768, 'none'), ('fc1', 3072, 768, 'gelu'), ('fc2', 768, 3072, 'residual')]:
    a = torch.randn(T, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(T, N, device=dev).bfloat16()
    fn = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
          'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
          'residual': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1)}[epi]
    for extra in [int(x) for x in os.environ.get('EXTRA_DBG', '0').split(',')]:
        dephase = str(extra)
        os.environ['MERLOT_DBG'] = '0'
        for _ in range(5):
            fn()
        os.environ['MERLOT_DBG'] = str(512 + extra)
        fn()
        os.environ['MERLOT_DBG'] = '0'
        buf = torch.zeros(256 * TT * 8, device=dev, dtype=torch.int64)
        LIB.call('merlot_probe_persist_trace', buf.data_ptr(), buf.numel() * 8, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        tr = buf.cpu().numpy().reshape(256, TT, 8)
        ntile = ((T + 255) // 256) * ((N + 255) // 256)
        per = ntile // 256
        use = tr[:, 1:per - 1]                               # interior tiles of every workgroup
        seg = {'loop': use[:, :, 1] - use[:, :, 0], 'unstagger': use[:, :, 2] - use[:, :, 1], 'epilogue issue': use[:, :, 3] - use[:, :, 2],
               'store wait': use[:, :, 4] - use[:, :, 3], 'barrier': use[:, :, 5] - use[:, :, 4]}
        gap = tr[:, 2:per - 1, 0] - tr[:, 1:per - 2, 5]
        line = ' | '.join(f'{k} {v.mean():7.0f} (p10 {np.percentile(v, 10):6.0f} p90 {np.percentile(v, 90):6.0f})' for k, v in seg.items())
        # concurrency inside XCD 0: at the moment a workgroup enters its epilogue, how many others of the XCD are inside theirs
        x = tr[0::8, 1:per - 1]
        starts, ends = x[:, :, 2].reshape(-1), x[:, :, 4].reshape(-1)
        conc = [(int(((starts <= s) & (ends > s)).sum())) for s in starts]
        print(f'{name:5s} dbg+{dephase:>4s}: cycles per tile: {line} | next-tile gap {gap.mean():5.0f} | '
              f'XCD0 workgroups in epilogue when one enters: mean {np.mean(conc):4.1f} of 32 (p10 {np.percentile(conc, 10):.0f}, p90 {np.percentile(conc, 90):.0f})', flush=True)
os.environ['MERLOT_P8_DEPHASE'] = '0'
