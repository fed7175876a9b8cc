"""Round 6: the fused attention backward of 257 .. 512 tokens with its K | V and Q | dO images arriving in groups of 128 rows UNDER the passes that
read them (attn_bwd_fused_kernel<512, .., CH = true>) against the drain-and-barrier form of rounds 3 - 5.  Experiments build, MERLOT_ATTN_FB_CH = 1 | 0, mirrored
order; the two forms do the same arithmetic in the same order: dqkv must agree bit for bit, the log sums (atomics) to the last bits."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops


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


torch.manual_seed(0)
for name, B, S, nfix, log in (('joint + log (512 x 328)', 512, 328, 200, True), ('joint (512 x 328)', 512, 328, 200, False), ('text-only (1536 x 512, masked)', 1536, 512, 64, False),
                              ('as-shipped ViT (896 x 266, no mask)', 896, 266, 266, False), ('as-shipped joint + log (224 x 396)', 224, 396, 268, True),
                              ('sort_story joint (320 x 410, masked)', 320, 410, 250, False), ('258 tokens, no mask', 512, 258, 258, False), ('385 tokens + log', 256, 385, 200, True),
                              ('266 tokens, masked', 512, 266, 200, False), ('300 tokens, masked', 512, 300, 200, False), ('320 tokens, masked', 512, 320, 200, False),
                              ('300 tokens, no mask', 512, 300, 300, False), ('328 tokens, no mask', 512, 328, 328, False), ('400 tokens, no mask', 384, 400, 400, False)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = None
    if nfix < S:
        n = torch.randint(8, S - nfix + 1, (B, 1), device='cuda')
        valid = ((torch.arange(S, device='cuda')[None, :] < nfix + n)).to(torch.uint8).contiguous()
    o, lse = ops.attention_fwd(qkv, B, S, 12, valid)
    do = torch.randn_like(o)
    out, ts = {}, []
    for mode in ('0', '1', '1', '0'):
        os.environ['MERLOT_ATTN_FB_CH'] = mode
        lo, hi = torch.zeros(B, S, device='cuda'), torch.zeros(B, S, device='cuda')
        kw = dict(log_lo=lo, log_hi=hi, log_split=200, log_weight=1 / 12) if log else {}
        d = ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid, **kw)
        out[mode] = (d.clone(), lo.clone(), hi.clone())
        ts.append((mode, timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid, **kw))))
    same = torch.equal(out['0'][0], out['1'][0])
    logs = float((out['0'][1] - out['1'][1]).abs().max()), float((out['0'][2] - out['1'][2]).abs().max())
    print(f'{name:44s} ' + '  '.join(f'{"chunked" if m == "1" else "drained"} {t:7.1f} us' for m, t in ts) + f'   dqkv identical={same}  log sums max diff {logs[0]:.1e} {logs[1]:.1e}', flush=True)
