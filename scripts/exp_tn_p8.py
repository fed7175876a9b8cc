"""Round 2: TN (wgrad) ping-pong kernel (MERLOT_TN_P8=1) against the 128x256 ring kernel (=0): correctness vs the torch fp32
product (and vs the ring kernel), then timing in mirrored order on the wgrad shapes of the step."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from exp_epi import bench

torch.manual_seed(0)
dev = 'cuda'
bad = 0
for (R, M, N) in [(4096, 768, 768), (8192, 3072, 768), (20000, 2304, 768), (16384, 768, 3072), (5000, 1000, 770), (9000, 256, 128)]:
    a = torch.randn(R, M, device=dev).bfloat16()
    b = torch.randn(R, N, device=dev).bfloat16()
    ref = a.float().t() @ b.float()
    outs = {}
    for k in (0, 1):
        os.environ['MERLOT_TN_P8'] = str(k)
        o = torch.full((M, N), 3.0, device=dev)
        ops.gemm_tn(a, b, o, accumulate=False)
        ops.gemm_tn(a, b, o, accumulate=True, alpha=0.5)
        outs[k] = o
        rel = float((o - 1.5 * ref).norm() / (1.5 * ref).norm())
        if rel > 2e-3:
            bad += 1
        print(f'R={R} M={M} N={N} kernel {k}: rel-L2 vs fp32 {rel:.2e}', flush=True)
print('tn p8 correctness:', 'OK' if bad == 0 else f'{bad} BAD')
for T in (101376, 41984, 16384):
    for (M, N, name) in [(768, 768, 'dWproj'), (3072, 768, 'dW1'), (768, 3072, 'dW2'), (2304, 768, 'dWqkv')]:
        a = torch.randn(T, M, device=dev).bfloat16()
        b = torch.randn(T, N, device=dev).bfloat16()
        out = torch.zeros((M, N), device=dev)
        fn = lambda: ops.gemm_tn(a, b, out, accumulate=True)
        os.environ['MERLOT_TN_P8'] = '0'
        bench(fn, 30)
        row = []
        for k in (0, 1, 1, 0):
            os.environ['MERLOT_TN_P8'] = str(k)
            t = bench(fn, 20)
            row.append(f'{"ring" if k == 0 else "p8"}: {t:7.1f} us {2.0 * T * M * N / t / 1e6:5.0f} TF')
        print(f'T={T} {name:6s} [{M} x {N}]  ' + ' | '.join(row), flush=True)
