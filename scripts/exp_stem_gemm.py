"""NT GEMM shapes of the ResNet-hybrid stem at 512 frames: which tile config serves narrow outputs."""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops
from exp_epi import bench
for M, N, K, name in [(512 * 3136, 64, 64, 'g1 1x1 64->64'), (512 * 3136, 256, 64, 'g1 1x1 64->256'), (512 * 3136, 64, 256, 'g1 1x1 256->64'),
                      (512 * 3136, 64, 576, 'g1 3x3 64'), (512 * 12544, 32, 320, 'stem 3x3 32'), (512 * 3136, 128, 256, 'g2 1x1 256->128'),
                      (512 * 3136, 128, 1152, 'g2 3x3 128 @56'), (512 * 196, 256, 1024, 'g3 1x1 1024->256'), (512 * 196, 1024, 256, 'g3 1x1 256->1024')]:
    a = torch.randn(M, K, device='cuda').bfloat16()
    b = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    row = []
    for c in (-1, 21, 11, 14, 15):
        if c == -1:
            os.environ.pop('MERLOT_NT_CFG_DYN', None)
        else:
            os.environ['MERLOT_NT_CFG_DYN'] = str(c)
        try:
            t = bench(lambda: ops.gemm_nt(a, b), 10)
            row.append(f'{"auto" if c < 0 else c}: {t:7.1f}us')
        except Exception as e:
            row.append(f'{c}: err')
    mb = (M * K + M * N) * 2 / 1e6
    print(f'{name:20s} M={M} N={N} K={K} ({mb:6.0f} MB) ' + ' '.join(row), flush=True)
