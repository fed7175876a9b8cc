"""Per-K-TILE timeline of the ping-pong NT kernel (MERLOT_DBG=1024, experiments build): thread 0 of every workgroup stamps s_memtime at
the start of every K-tile of its first 16 tiles, at the end of the K loop and at the end of the tile.  Question: a K = 768 tile spends
~3 040 cycles per K-tile in its main loop, a K = 3 072 tile ~2 490 -- which K-tiles of a short tile are the slow ones?"""
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
for name, N, K, epi in [('qkv', 2304, 768, 'none'), ('fc1', 3072, 768, 'gelu'), ('dgrad_proj', 768, 768, 'none')]:
    a = torch.randn(T, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    fn = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
          'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux)}[epi]
    os.environ['MERLOT_DBG'] = '0'
    for _ in range(5):
        fn()
    os.environ['MERLOT_DBG'] = '1024'
    fn()
    os.environ['MERLOT_DBG'] = '0'
    buf = torch.zeros(256 * 32 * 8, device=dev, dtype=torch.int64)
    LIB.call('merlot_probe_persist_trace', buf.data_ptr(), buf.numel() * 8, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    tr = buf.cpu().numpy().reshape(256, 16, 16)
    nk = K // 64
    ntile = ((T + 255) // 256) * ((N + 255) // 256)
    per = min(ntile // 256, 16)
    use = tr[:, 1:per - 1]                                   # interior tiles
    kt = np.concatenate([use[:, :, 1:nk] - use[:, :, 0:nk - 1], (use[:, :, 14] - use[:, :, nk - 1])[:, :, None]], axis=2)   # duration of K-tile i
    epi_t = use[:, :, 15] - use[:, :, 14]
    gap = tr[:, 2:per - 1, 0] - tr[:, 1:per - 2, 15]
    print(f'{name:10s} [T x {N} x {K}] cycles per K-tile (mean over interior tiles of 256 workgroups): ' + ' '.join(f'{v:5.0f}' for v in kt.mean(axis=(0, 1))) +
          f' | loop {kt.sum(axis=2).mean():6.0f} | epilogue + barriers {epi_t.mean():6.0f} | gap to the next tile {gap.mean():4.0f}', flush=True)
    print(f'{"":10s} p10: ' + ' '.join(f'{v:5.0f}' for v in np.percentile(kt, 10, axis=(0, 1))) + ' | p90: ' + ' '.join(f'{v:5.0f}' for v in np.percentile(kt, 90, axis=(0, 1))), flush=True)
