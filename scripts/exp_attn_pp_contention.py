"""Round 5: the persistent attention kernels (one workgroup per CU, static item lists) when a communication kernel holds CUs -- RCCL keeps one CU per
channel while a collective runs (NCCL_MAX_NCHANNELS=16 in bench.py) and the backward of the transformer stacks is what the gradient all-reduce overlaps.
`merlot_probe_cu_hog` pins H whole CUs (160 KiB of LDS each) for the duration; old = the one-shot kernels (24 576 workgroups), pp = the persistent ones."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from probe_lib import cu_hog

B, S = 2048, 198
qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
os.environ['MERLOT_ATTN_PP'] = '0'
o, lse = ops.attention_fwd(qkv, B, S, 12, None)
do = torch.randn_like(o)
Bm, Sm = 512, 328
qkvm = (torch.randn(Bm * Sm, 2304, device='cuda') * 0.7).bfloat16()
valid = (torch.rand(Bm, Sm, device='cuda') > 0.1).to(torch.uint8)
valid[:, 0] = 1
side = torch.cuda.Stream()
sink = torch.zeros(4, device='cuda', dtype=torch.int32)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
cu_hog(1, 160 * 1024, 20_000_000, sink)
e1.record()
torch.cuda.synchronize()
hz = 20_000_000 / (e0.elapsed_time(e1) * 1e-3)


def timed(fn, hog, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if hog:
        with torch.cuda.stream(side):
            cu_hog(hog, 160 * 1024, int(0.2 * hz), sink)          # 200 ms: longer than the timed region
        torch.cuda.current_stream().wait_stream(torch.cuda.current_stream())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    time.sleep(0.005)                                              # let the hog take its CUs
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


cases = {'ViT forward': lambda: ops.attention_fwd(qkv, B, S, 12, None),
         'ViT backward': lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, None),
         'joint training forward': lambda: ops.attention_fwd(qkvm, Bm, Sm, 12, valid)}
for name, fn in cases.items():
    for pp in ('0', '1'):
        os.environ['MERLOT_ATTN_PP'] = pp
        row = []
        base = None
        for hog in (0, 8, 16, 32):
            t = timed(fn, hog)
            base = base or t
            row.append(f'{hog:2d} CUs held {t:7.1f} us ({100 * (t / base - 1):+5.1f} %)')
        print(f'{name:24s} {"pp " if pp == "1" else "old"}: ' + ' | '.join(row), flush=True)
