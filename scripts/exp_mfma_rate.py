"""Matrix-pipe ceiling under the 256x256 GEMM's K-step instruction mix (merlot_probe_mfma_rate): what pure MFMA issue
reaches on this box, and what the fragment reads / the per-K-step barrier cost on top -- no global memory traffic."""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd.lib import LIB

dev = torch.device('cuda', 0)
sink = torch.zeros(4, device=dev)
iters = 4000
for blocks in (256, 512):
    for mode, name in [(0, 'MFMA only'), (1, '+ 12 ds_read_b128 / K-step (results unused)'), (5, '+ reads feeding the MFMAs'),
                       (2, '+ s_barrier / K-step'), (7, '+ reads feeding the MFMAs + s_barrier')]:
        out = torch.zeros((blocks, 4), dtype=torch.int64, device=dev)
        for _ in range(2):
            _exp_lib.PROBE.call('merlot_probe_mfma_rate', blocks, iters, mode, out.data_ptr(), sink.data_ptr(), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _exp_lib.PROBE.call('merlot_probe_mfma_rate', blocks, iters, mode, out.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        o = out.cpu().double()
        flops = blocks * 8 * iters * 16 * 2.0 * 32 * 32 * 16
        clk, wall = o[:, 0].mean().item(), o[:, 1].mean().item()
        print(f'blocks {blocks:4d} {name:48s}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.0f} TFLOP/s   '
              f's_memtime/s_memrealtime = {clk / wall:6.2f}  per-block wall {wall / 100:8.1f} us  '
              f'MFMA-busy if 32 clk each @2.4 GHz: {iters * 16 * 2 * 32 / 2400 / (wall / 100):5.2f}', flush=True)
