"""3x3 convolution of the ResNet-hybrid stem: implicit GEMM (merlot_conv3x3_bf16) against the explicit path (im2col3x3 + gemm_nt; input
gradient: gemm_nt to [T, 9 C] + col2im3x3) at the stem's shapes, 1 024 frames; us per layer and direction."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402
from ab_lib_tn import bench  # noqa: E402

BF16 = torch.bfloat16
N = int(os.environ.get('FRAMES', 1024))
for (H, C, Co) in ((112, 32, 32), (112, 32, 64), (56, 64, 64), (56, 128, 128), (28, 128, 128), (28, 256, 256), (14, 256, 256)):
    x = torch.randn(N, H, H, C, device='cuda').to(BF16)
    Kp = (9 * C + 63) // 64 * 64
    w = torch.zeros(Co, Kp, device='cuda', dtype=BF16)
    w[:, :9 * C] = (torch.randn(Co, 9 * C, device='cuda') / (3 * C ** 0.5)).to(BF16)
    wT = torch.zeros(Kp, (Co + 63) // 64 * 64, device='cuda', dtype=BF16)
    wT[:9 * C, :Co] = w[:, :9 * C].t()
    wdg = w[:, :9 * C].reshape(Co, 3, 3, C).flip(1, 2).permute(3, 1, 2, 0).reshape(C, 9 * Co).contiguous()
    dy = torch.randn(N, H, H, Co, device='cuda').to(BF16)
    dyp = dy.reshape(-1, Co)
    if wT.shape[1] != Co:
        dyp = torch.zeros(N * H * H, wT.shape[1], device='cuda', dtype=BF16)
        dyp[:, :Co] = dy.reshape(-1, Co)
    same = torch.equal(ops.conv3x3(x, w, Co), ops.gemm_nt(ops.im2col3x3(x), w).view(N, H, H, Co))
    t_fi = bench(lambda: ops.conv3x3(x, w, Co), 10)
    t_fe = bench(lambda: ops.gemm_nt(ops.im2col3x3(x), w), 10)
    t_bi = bench(lambda: ops.conv3x3(dy, wdg, C), 10)
    t_be = bench(lambda: ops.col2im3x3(ops.gemm_nt(dyp, wT), N, H, H, C), 10)
    dw = torch.zeros(Co, Kp, device='cuda')
    dy2 = dy.reshape(-1, Co)
    t_wi = bench(lambda: ops.conv3x3_wgrad(dy2, x, dw), 10)
    t_we = bench(lambda: ops.gemm_tn(dy2, ops.im2col3x3(x), dw, accumulate=False), 10)
    gf = 2.0 * N * H * H * 9 * C * Co / 1e9
    print(f'[{N} x {H}^2] {C:3d} -> {Co:3d}: forward implicit {t_fi:7.1f} us ({gf / t_fi * 1e3:5.0f} TFLOP/s) explicit {t_fe:7.1f} us | '
          f'input gradient implicit {t_bi:7.1f} us explicit {t_be:7.1f} us | weight gradient implicit {t_wi:7.1f} us ({gf / t_wi * 1e3:5.0f} TFLOP/s) explicit (im2col + gemm_tn) {t_we:7.1f} us | forward bits equal: {same}', flush=True)
    del x, dy, dyp
    torch.cuda.empty_cache()
