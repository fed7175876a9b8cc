"""What bounds the GEMM epilogue's stores?  merlot_probe_store writes 256 x 256 bf16 tiles with the epilogue's own pattern
from 256 / 128 / 64 / 32 workgroups (one per CU), with and without the per-tile vmcnt(0) + barrier, and with longer
contiguous row segments per instruction."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from probe_lib import PROBE  # noqa: E402

dev = torch.device('cuda', 0)
M, N = 101376, 3072
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
tiles_m, tiles_n = M // 256, N // 256
ntiles = tiles_m * tiles_n
clk = torch.zeros(256, device=dev, dtype=torch.int64)
s = torch.cuda.current_stream().cuda_stream


def run(blocks, rpi, mode):
    tiles = ntiles // blocks
    for _ in range(2):
        PROBE.call('merlot_probe_store', out.data_ptr(), N, tiles_m, blocks, tiles, rpi, mode, clk.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    PROBE.call('merlot_probe_store', out.data_ptr(), N, tiles_m, blocks, tiles, rpi, mode, clk.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    nbytes = blocks * tiles * 256 * 256 * 2
    c = clk[:blocks].float().mean().item()
    return us, nbytes / us * 1e-3, us / tiles, nbytes / blocks / c


print(f'[{M} x {N}] bf16 = {M * N * 2 / 1e6:.0f} MB per pass, {ntiles} tiles')
for mode, label in ((3, 'vmcnt(0)+barrier per tile, 1 WG/CU'), (2, 'no per-tile wait, 1 WG/CU')):
    for rpi in (8, 4, 2, 1):
        row = []
        for blocks in (256, 128, 64, 32):
            us, gbs, us_tile, bpc = run(blocks, rpi, mode)
            row.append(f'{blocks:3d} WGs: {gbs:6.0f} GB/s {us_tile:5.2f} us/tile {bpc:5.1f} B/clk/CU')
        print(f'{label}; {1024 // rpi:4d} B per row and instruction | ' + ' | '.join(row), flush=True)
