"""Round 4: what does VALU work placed in the MFMA segments of the ping-pong NT kernel cost?  Experiments builds with MERLOT_EXP_FILLER = 0 / 2 / 4
independent v_fma_f32 behind every MFMA of the main loop (EXP_LIB=<library>): whole launch and main loop alone (MERLOT_DBG=1), us."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops, lib
from ab_lib_tn import bench

T = 101376
torch.manual_seed(0)
os.environ['MERLOT_NT_CFG_DYN'] = '22'
row = []
for name, N, K in [('qkv', 2304, 768), ('fc2', 768, 3072)]:
    a = torch.randn(T, K, device='cuda').bfloat16()
    b = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
    bias = torch.randn(N, device='cuda') * 0.1
    fn = lambda: ops.gemm_nt(a, b, bias=bias)
    os.environ['MERLOT_DBG'] = '0'
    t = bench(fn, 20)
    os.environ['MERLOT_DBG'] = '1'
    tl = bench(fn, 20)
    os.environ['MERLOT_DBG'] = '0'
    row.append(f'{name} [{T} x {N} x {K}]: launch {t:6.1f} us, main loop alone {tl:6.1f} us ({2.0 * T * N * K / tl / 1e6:5.0f} TF)')
print(f'{os.path.basename(lib.LIB.path):26s} ' + ' | '.join(row), flush=True)
