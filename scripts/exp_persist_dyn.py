"""Static vs dynamic tile claims of the persistent NT GEMM, alone and beside a kernel that keeps N CUs' LDS busy
(stand-in for an RCCL all-reduce overlapping the backward).  gpurun: python scripts/exp_persist_dyn.py"""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os
import sys
import ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from merlot_amd.lib import LIB

dev = 'cuda'
torch.manual_seed(0)
shapes = [(101376, 3072, 768, 'fc1 fwd (GELU)'), (101376, 768, 3072, 'fc2 fwd'), (101376, 2304, 768, 'qkv fwd'),
          (41984, 3072, 768, 'joint fc1')]
side = torch.cuda.Stream()
sink = torch.zeros(4, dtype=torch.int32, device=dev)


def run(a, b, bias, cfg, hog_blocks, iters=20):
    M, K = a.shape
    N = b.shape[0]
    os.environ['MERLOT_NT_CFG_DYN'] = str(cfg)
    for _ in range(3):
        ops.gemm_nt(a, b, bias=bias)
    torch.cuda.synchronize()
    if hog_blocks:
        # ~ (iters * 0.6 ms) of hogging at ~2 GHz
        with torch.cuda.stream(side):
            _exp_lib.PROBE.call('merlot_probe_cu_hog', hog_blocks, 96 * 1024, ctypes.c_int64(int(iters * 3e6)),
                     sink.data_ptr(), side.cuda_stream)
        torch.cuda._sleep(200000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = ops.gemm_nt(a, b, bias=bias)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6, c


for M, N, K, name in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16() * 0.02
    bias = torch.zeros(N, device=dev)
    ref = None
    for hog in (0, 16, 32, 64):
        row = []
        for cfg in (20, 21):
            us, tf, c = run(a, b, bias, cfg, hog)
            if ref is None:
                ref = c
            else:
                assert torch.equal(c, ref), 'static and dynamic results differ'
            row.append(f'cfg{cfg}: {us:8.1f} us {tf:7.1f} TF')
        print(f'{name:16s} hog={hog:3d} CUs | ' + ' | '.join(row), flush=True)
