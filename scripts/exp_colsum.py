import os, sys, torch
sys.path.insert(0, '/root/repo')
from merlot_amd import ops
def bench(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for B, S, split in ((32, 512, None), (128, 328, 200)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.5).bfloat16()
    valid = (torch.rand(B, S, device='cuda') > 0.2).to(torch.uint8)
    valid[:, 0] = 1
    o, lse = ops.attention_fwd(qkv, B, S, 12, valid)
    cs = torch.zeros(B, S, device='cuda'); cs2 = torch.zeros(B, S, device='cuda')
    if split is None:
        t = bench(lambda: ops.attention_colsum(qkv, lse, B, S, 12, cs, valid=valid, valid_q_only=False, weight=1/12))
    else:
        t = bench(lambda: ops.attention_colsum(qkv, lse, B, S, 12, cs, cs2, qsplit=split, valid=valid, valid_q_only=True, weight=1/12))
    tf = bench(lambda: ops.attention_fwd(qkv, B, S, 12, valid))
    print(f'B={B} S={S}: colsum {t:.1f} us, attention fwd {tf:.1f} us')
