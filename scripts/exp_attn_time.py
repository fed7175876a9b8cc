"""attention forward / backward timing at the step's shapes (product library)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, S, masked in ((512, 198, False), (128, 328, True), (192, 578, False), (12, 2832, True), (512, 32, True)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = torch.ones(B, S, dtype=torch.uint8, device='cuda') if masked else None
    o, lse = ops.attention_fwd(qkv, B, S, 12, valid)
    do = torch.randn_like(o)
    tf = timeit(lambda: ops.attention_fwd(qkv, B, S, 12, valid))
    tb = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid))
    fl = 4.0 * S * S * 768 * B
    print(f'B {B:4d} S {S:5d} masked {masked!s:5s}: fwd {tf:7.1f} us {fl / tf * 1e-6:5.0f} TF | bwd {tb:7.1f} us {2.5 * fl / tb * 1e-6:5.0f} TF', flush=True)
