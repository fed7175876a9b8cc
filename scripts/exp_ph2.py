"""Round 3: two phases per K-tile (MERLOT_P8_PH2 / MERLOT_TN_PH2 = 1, experiments build) against four (= 0) in the ping-pong NT and
TN kernels: outputs must be bit-identical (the accumulation order per accumulator is unchanged), then timing in mirrored order
on the step's shapes, whole launch and main loop only (MERLOT_DBG=1)."""
import _exp_lib  # noqa: F401
import os
import sys
import torch
from merlot_amd import ops
from exp_epi import bench

dev = 'cuda'
T = int(os.environ.get('T', 101376))
torch.manual_seed(0)
os.environ['MERLOT_NT_CFG_DYN'] = '22'
NT_SHAPES = [] if os.environ.get('SKIP_NT') else [('qkv', 2304, 768, 'none'), ('proj', 768, 768, 'residual'), ('fc1', 3072, 768, 'gelu'), ('fc2', 768, 3072, 'residual'),
                        ('dgrad_fc1', 768, 3072, 'none'), ('dgrad_fc2', 3072, 768, 'dgelu'), ('dgrad_qkv', 768, 2304, 'none')]
for name, N, K, epi in NT_SHAPES:
    a = torch.randn(T, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(T, N, device=dev).bfloat16()
    fn = {'none': lambda: ops.gemm_nt(a, b, bias=bias),
          'gelu': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
          'residual': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1),
          'dgelu': lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res)}[epi]
    os.environ['MERLOT_P8_PH2'] = '0'
    os.environ['MERLOT_DBG'] = '0'
    bench(fn, 40)
    ref, row = None, []
    for ph in ('0', '1', '1', '0'):
        os.environ['MERLOT_P8_PH2'] = ph
        o = fn().clone()
        ref = o if ref is None else ref
        same = torch.equal(o, ref)
        t = bench(fn, 30)
        os.environ['MERLOT_DBG'] = '1'
        tl = bench(fn, 30)
        os.environ['MERLOT_DBG'] = '0'
        row.append(f'ph{4 if ph == "0" else 2}: {t:6.1f} ({2.0 * T * N * K / t / 1e6:4.0f} TF) loop {tl:6.1f}{"" if same else " MISMATCH"}')
    print(f'{name:10s} [{T} x {N} x {K}] {epi:8s} us  ' + ' | '.join(row), flush=True)

for (M, N, name) in [(768, 768, 'dWproj'), (3072, 768, 'dW1'), (768, 3072, 'dW2'), (2304, 768, 'dWqkv')]:
    a = torch.randn(T, M, device=dev).bfloat16()
    b = torch.randn(T, N, device=dev).bfloat16()
    fn = lambda out: ops.gemm_tn(a, b, out, accumulate=False)
    os.environ['MERLOT_TN_PH2'] = '0'
    out = torch.zeros((M, N), device=dev)
    bench(lambda: fn(out), 20)
    row, ref = [], None
    for ph in os.environ.get('TN_MODES', '0,1,3,3,1,0').split(','):
        os.environ['MERLOT_TN_PH2'] = ph
        o = torch.zeros((M, N), device=dev)
        fn(o)
        ref = o.clone() if ref is None else ref
        same = torch.equal(o, ref)
        t = bench(lambda: fn(out), 20)
        row.append(f'ph{ {"0": 4, "1": 2, "3": 1}[ph] }: {t:7.1f} us {2.0 * T * M * N / t / 1e6:5.0f} TF{"" if same else " MISMATCH"}')
    print(f'T={T} {name:6s} [{M} x {N}]  ' + ' | '.join(row), flush=True)
