"""Round 6: per-tile timeline of the LayerNorm-fold launch (experiments build, MERLOT_DBG=512): thread 0 of every workgroup stamps s_memtime at tile start (0),
K-loop end (1), after the un-stagger barrier (2), after the epilogue issued its stores (3), after their wait (4), after the arrival count came back (6), at tile end (5).
4 -> 5 = the closing barrier and, fused, the look at the previous tile's arrival count (issued at the epilogue's start) + the row block's LayerNorm pass where this
workgroup's tile was the last of the three."""
import _exp_lib  # noqa: F401
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from merlot_amd.lib import LIB

T = int(os.environ.get('T', 405504))
TT = 32
torch.manual_seed(0)
for name, K in (('proj', 768), ('fc2', 3072)):
    a = torch.randn(T, K, device='cuda').bfloat16()
    w = (torch.randn(768, K, device='cuda') * 0.02).bfloat16()
    bias = torch.randn(768, device='cuda') * 0.1
    res = torch.randn(T, 768, device='cuda').bfloat16()
    g, b = torch.ones(768, device='cuda'), torch.zeros(768, device='cuda')
    for fused in (0, 1):
        fn = (lambda: ops.gemm_nt_ln(a, w, g, b, bias=bias, aux_in=res, dropout_p=0.1, dropout_seed=1)) if fused else \
             (lambda: ops.gemm_nt(a, w, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1))
        os.environ['MERLOT_DBG'] = '0'
        for _ in range(3):
            fn()
        os.environ['MERLOT_DBG'] = '512'
        fn()
        os.environ['MERLOT_DBG'] = '0'
        buf = torch.zeros(256 * TT * 8, device='cuda', dtype=torch.int64)
        LIB.call('merlot_probe_persist_trace', buf.data_ptr(), buf.numel() * 8, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        tr = buf.cpu().numpy().reshape(256, TT, 8)
        per = min(TT, (T // 256) * 3 // 256)
        use = tr[:, 1:per - 1]
        seg = {'loop': use[..., 1] - use[..., 0], 'unstagger': use[..., 2] - use[..., 1], 'epilogue': use[..., 3] - use[..., 2], 'store wait': use[..., 4] - use[..., 3]}
        tail = (use[..., 5] - use[..., 4]).reshape(-1)      # closing barrier (+ fused: the previous tile's arrival looked at, its row block normalised if this workgroup closed it)
        if fused:
            big = tail > 12000
            seg['tail, no LayerNorm pass'] = tail[~big]
            seg['tail, with a LayerNorm pass'] = tail[big] if big.any() else np.zeros(1)
            frac = big.mean()
        else:
            seg['closing barrier'] = tail
            frac = 0.0
        gap = tr[:, 2:per - 1, 0] - tr[:, 1:per - 2, 5]
        print(f'{name} K={K} fused={fused}: ' + ' | '.join(f'{k} {np.mean(v):7.0f}' for k, v in seg.items()) + f' | next-tile gap {gap.mean():5.0f} | tiles that ran the LayerNorm pass {frac:.2f}',
              flush=True)
