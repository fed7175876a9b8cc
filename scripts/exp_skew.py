"""A/B harness for NT GEMM configurations (CFGS=21,11,11,21 ...): full kernel and main loop only (MERLOT_DBG=1) on the
step's big shapes, mirrored order after a warm-up; results must agree bit for bit (same MFMA order per accumulator).
Used for the id 21 vs id 22 (half-step skewed loop, since removed) and 256x256 vs 128x256 comparisons of
profiles/r01_i_gemm_ceiling.txt."""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from exp_epi import bench

dev = 'cuda'
T = int(os.environ.get('T', 101376))
torch.manual_seed(0)
ROWS = os.environ.get('ROWS')
for name, N, K, epi in [r for r in [('qkv', 2304, 768, 'none'), ('proj', 768, 768, 'residual'), ('fc1', 3072, 768, 'gelu'), ('fc2', 768, 3072, 'residual'),
                        ('dgrad_fc1', 768, 3072, 'none'), ('dgrad_fc2', 3072, 768, 'dgelu'), ('dgrad_proj', 768, 768, 'none')] if not ROWS or r[0] in ROWS.split(',')]:
    a = torch.randn(T, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(T, N, device=dev).bfloat16()
    fn = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
          'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
          'residual': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1),
          'dgelu': lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res)}[epi]
    out, row = {}, []
    cfgs = os.environ.get('CFGS', '21,11,11,21').split(',')       # ABBA: the first measurement after new allocations
    os.environ['MERLOT_NT_CFG_DYN'] = cfgs[0]                      # runs ~10 % slow (clock ramp), so warm up first and
    os.environ['MERLOT_DBG'] = '0'                                 # measure every config twice in mirrored order
    bench(fn, 60)
    for cfg in cfgs:
        os.environ['MERLOT_NT_CFG_DYN'] = cfg
        os.environ['MERLOT_DBG'] = '0'
        out[cfg] = fn().clone()
        t = bench(fn, 30)
        os.environ['MERLOT_DBG'] = '1'
        tl = bench(fn, 30)
        os.environ['MERLOT_DBG'] = '0'
        fl = 2.0 * T * N * K
        row.append(f'id {cfg}: {t:7.1f} us {fl / t / 1e6:6.0f} TF | loop only {tl:7.1f} us {fl / tl / 1e6:6.0f} TF')
    ks = list(out)
    same = all(torch.equal(out[ks[0]], out[k]) for k in ks[1:])
    print(f'{name:10s} [{T} x {N} x {K}] {epi:8s} ' + '   '.join(row) + f'   identical={same}', flush=True)
    assert same
