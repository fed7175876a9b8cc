"""Round 6: what dqkv's 8-bit copy costs the tiled attention backward (merlot_attention_bwd_q8) against the quantising pass it replaces, at config #5's shapes.
python scripts/exp_attn_bwd_q8.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops  # noqa: E402


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


heads = 12
for (B, S, masked, tag) in ((768, 578, False, 'config #5 ViT'), (48, 2832, True, 'config #5 joint')):
    g = torch.Generator(device='cuda').manual_seed(0)
    qkv = torch.randn(B * S, 3 * heads * 64, device='cuda', generator=g).bfloat16()
    valid = torch.ones(B, S, dtype=torch.uint8, device='cuda') if masked else None
    out, lse = ops.attention_fwd(qkv, B, S, heads, valid)
    dout = (torch.randn(B * S, heads * 64, device='cuda', generator=g) * 1e-2).bfloat16()
    blk = torch.tensor([1e4, 1e-4, 1.0, 0.0], device='cuda')
    t0 = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, S, heads, valid))
    t1 = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, S, heads, valid, q8_block=blk, q8_fmt=1))
    dqkv = ops.attention_bwd(qkv, out, dout, lse, B, S, heads, valid)
    d8, sb = ops.quantize_f8(dqkv, 1)
    t2 = timeit(lambda: ops.quantize_f8(dqkv, 1, out=d8, scale=sb))
    print(f'{tag:16s} B {B} S {S}: attention backward {t0:8.1f} us | with the e5m2 copy {t1:8.1f} us (+{t1 - t0:6.1f}) | the quantising pass it replaces {t2:7.1f} us')
