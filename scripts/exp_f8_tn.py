"""Round 6: the 8-bit weight-gradient kernel (merlot_gemm_f8_tn, csrc/gemm_q8.inc).
  1. what ds_read_b64_tr_b8 does (the lane map gemm_q8.inc assumes), from libmerlot_probe.so;
  2. the kernel against a matmul of the DEQUANTISED operands (exact products, fp32 sums), e4m3 x e4m3 and e5m2 x e4m3;
  3. A/B against merlot_gemm_bf16_tn at the weight-gradient shapes of config #5 (M rows = 48 x 16 x 578) and of the headline (405 504 rows):
     kernel alone, and kernel + the quantising passes (current: amax + convert; delayed: one pass).
python scripts/exp_f8_tn.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_amd import ops  # noqa: E402

dev = 'cuda'


def probe_tr8():
    lib = ctypes.CDLL(os.path.join(ROOT, 'merlot_amd', 'libmerlot_probe.so'))
    lib.merlot_probe_tr8.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    # byte (lane i, b) of the LDS image carries a 9-bit id split over two runs: run 0 the lane, run 1 the byte index
    src_lane = torch.arange(64, dtype=torch.uint8).repeat_interleave(8).to(dev)
    src_byte = torch.arange(8, dtype=torch.uint8).repeat(64).to(dev)
    o_lane = torch.zeros(512, dtype=torch.uint8, device=dev)
    o_byte = torch.zeros(512, dtype=torch.uint8, device=dev)
    assert lib.merlot_probe_tr8(src_lane.data_ptr(), o_lane.data_ptr(), None) == 0
    assert lib.merlot_probe_tr8(src_byte.data_ptr(), o_byte.data_ptr(), None) == 0
    torch.cuda.synchronize()
    ol = o_lane.cpu().numpy().reshape(64, 8)
    ob = o_byte.cpu().numpy().reshape(64, 8)
    print('ds_read_b64_tr_b8: lane t, result byte j  <-  (source lane, source byte)')
    for t in (0, 1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 48, 63):
        print(f'  lane {t:2d}: ' + ' '.join(f'({ol[t, j]:2d},{ob[t, j]})' for j in range(8)))
    # the assumed map: within a 16-lane group, result (t, j) = source (lane 2 j + (t >> 3) of the group, byte t & 7)
    ok = True
    for t in range(64):
        g, i = t & ~15, t & 15
        for j in range(8):
            ok &= (ol[t, j] == g + 2 * j + (i >> 3)) and (ob[t, j] == (i & 7))
    print('assumed lane map (result (t, j) = source lane 2 j + (t >> 3) of the 16-lane group, byte t & 7):', 'HOLDS' if ok else 'DOES NOT HOLD')
    return ok


def deq(y8, scale):
    return y8.float() * scale[1]


def check(R, M, N, fa, fb, seed=0, accumulate=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    a = (torch.randn(R, M, device=dev, generator=g) * torch.rand(1, M, device=dev, generator=g) * 3).bfloat16()
    b = (torch.randn(R, N, device=dev, generator=g) * 0.7).bfloat16()
    pad16 = lambda v: (v + 15) // 16 * 16
    Rp = (R + 127) // 128 * 128
    a8 = torch.empty(Rp, pad16(M), device=dev, dtype=ops._F8_DTYPES[fa])[:, :M]      # leading dimensions: multiples of 16 (the kernel's contract)
    b8 = torch.empty(Rp, pad16(N), device=dev, dtype=ops._F8_DTYPES[fb])[:, :N]
    a8, sa = ops.quantize_f8(a, fa, out=a8)
    b8, sb = ops.quantize_f8(b, fb, out=b8)
    out = torch.full((M, N), 0.5 if accumulate else float('nan'), device=dev)
    ops.gemm_f8_tn(a8, sa, b8, sb, out, accumulate=accumulate, alpha=0.75)
    ref = 0.75 * (deq(a8, sa)[:R].double().T @ deq(b8, sb)[:R].double()) + (0.5 if accumulate else 0.0)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    full = 0.75 * (a.double().T @ b.double()) + (0.5 if accumulate else 0.0)
    qerr = float((out.double() - full).norm() / full.norm())
    print(f'  R {R:7d} M {M:5d} N {N:5d} fmt ({fa},{fb}) acc {int(accumulate)}: max|C - deq matmul| / max|C| = {err:.2e}   rel-L2 against the bf16 operands {qerr:.2e}')
    return err


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def ab(R, M, N, label):
    a = (torch.randn(R, M, device=dev) * 0.02).bfloat16()
    b = torch.randn(R, N, device=dev).bfloat16()
    out = torch.zeros(M, N, device=dev)
    t_bf = timeit(lambda: ops.gemm_tn(a, b, out, accumulate=True))
    a8, sa = ops.quantize_f8(a, ops.F8_E5M2)
    b8, sb = ops.quantize_f8(b, ops.F8_E4M3)
    t_f8 = timeit(lambda: ops.gemm_f8_tn(a8, sa, b8, sb, out, accumulate=True))
    a8e, sae = ops.quantize_f8(a, ops.F8_E4M3)
    t_f8e = timeit(lambda: ops.gemm_f8_tn(a8e, sae, b8, sb, out, accumulate=True))
    t_qc = timeit(lambda: (ops.quantize_f8(a, ops.F8_E5M2, out=a8), ops.quantize_f8(b, ops.F8_E4M3, out=b8)))
    t_qd = timeit(lambda: (ops.quantize_f8(a, ops.F8_E5M2, out=a8, scale=sa), ops.quantize_f8(b, ops.F8_E4M3, out=b8, scale=sb)))
    fl = 2.0 * R * M * N
    print(f'{label:28s} R {R:7d} M {M:5d} N {N:5d}: bf16 {t_bf:7.1f} us ({fl / t_bf / 1e6:6.0f} TFLOP/s) | f8 e5m2 x e4m3 {t_f8:7.1f} us ({fl / t_f8 / 1e6:6.0f}) '
          f'e4m3 x e4m3 {t_f8e:7.1f} | x{t_bf / t_f8:.2f} | quantise both: current {t_qc:7.1f} us, delayed {t_qd:7.1f} us '
          f'({(R * (M + N) * 3) / t_qd / 1e6:.2f} TB/s) | f8 + delayed {t_f8 + t_qd:7.1f} us x{t_bf / (t_f8 + t_qd):.2f}')


if __name__ == '__main__':
    print(torch.cuda.get_device_name(0))
    probe_tr8()
    print('correctness (tolerance: the MFMA\'s own fp32 summation order, <= 3e-5 of the maximum)')
    worst = 0.0
    for (R, M, N, fa, fb, acc) in [(2048, 256, 256, 0, 0, False), (4096, 768, 768, 0, 0, False), (4096, 768, 768, 1, 0, True), (8192, 2304, 768, 1, 0, False),
                                   (6144, 200, 328, 0, 1, False), (2048 + 128 * 5 - 37, 3072, 768, 1, 1, True), (24576, 768, 3072, 1, 0, False)]:
        worst = max(worst, check(R, M, N, fa, fb, accumulate=acc))
    print('worst', worst, 'OK' if worst < 3e-5 else 'FAIL')
    if '--no-ab' not in sys.argv:
        for R, tag in ((48 * 16 * 578, 'config #5 ViT'), (48 * 2832, 'config #5 joint'), (405504, 'headline ViT'), (128 * 328, 'headline joint')):
            for (M, N, nm) in ((2304, 768, 'dWqkv'), (768, 768, 'dWproj'), (3072, 768, 'dW1'), (768, 3072, 'dW2')):
                if R * max(M, N) >= 2 ** 32 - 128 * 3072:
                    continue
                ab(R, M, N, f'{tag} {nm}')
