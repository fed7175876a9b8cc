"""Where the persistent NT GEMM's main loop loses time to its global->LDS stream (loop only, MERLOT_DBG bit 1):
normal | no loads (4) | L1-resident source (1024) | whole-cache-line pieces, each line fetched once (2048).
The three experiment switches lived in `stage_next` of gemm_nt_persist_dyn_kernel for the measurement of
profiles/r01_i_gemm_ceiling.txt section 2 and were REMOVED afterwards (their address arithmetic cost the production
loop 3-6 %); re-apply them from the commit "GEMM: half-step skewed main loop..." to re-run this script."""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from exp_epi import bench

dev = 'cuda'
T = 101376
for name, N, K in [('qkv', 2304, 768), ('fc1', 3072, 768), ('fc2', 768, 3072)]:
    a = torch.randn(T, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    fn = lambda: ops.gemm_nt(a, b)
    for cfg in ('21', '22'):
        os.environ['MERLOT_NT_CFG_DYN'] = cfg
        row = []
        for dbg, label in [(1, 'normal'), (1 | 4, 'no loads'), (1 | 1024, 'L1-resident'), (1 | 2048, 'whole lines')]:
            os.environ['MERLOT_DBG'] = str(dbg)
            t = bench(fn, 30)
            row.append(f'{label} {t:6.1f} us {2.0 * T * N * K / t / 1e6:5.0f} TF')
        os.environ['MERLOT_DBG'] = '0'
        print(f'{name:4s} id {cfg}: ' + ' | '.join(row), flush=True)
