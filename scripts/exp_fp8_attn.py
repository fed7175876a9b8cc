"""fp8 vs bf16 attention forward (tiled kernels) at the config-#5 shapes: ViT S = 578 (192 frames), joint S = 2832 (12 groups)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, S in ((192, 578), (12, 2832), (512, 198)):
    qkv = (torch.randn(B * S, 2304, device=dev) * 0.7).to(torch.bfloat16)
    valid = torch.ones(B, S, dtype=torch.uint8, device=dev)
    fl = 4.0 * S * S * 768 * B
    t16 = timeit(lambda: ops.attention_fwd(qkv, B, S, 12, valid))
    ta = timeit(lambda: ops.amax_groups(qkv, 3))
    am = ops.amax_groups(qkv, 3)
    t8 = timeit(lambda: ops.attention_fwd_fp8(qkv, B, S, 12, valid, amax3=am))
    print(f'B {B:4d} S {S:5d}: bf16 {t16:8.1f} us {fl / t16 * 1e-6:5.0f} TF | fp8 {t8:8.1f} us {fl / t8 * 1e-6:5.0f} TF | amax pass {ta:6.1f} us ({B * S * 2304 * 2 / ta * 1e-3:5.0f} GB/s)', flush=True)
