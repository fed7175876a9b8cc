"""Round 5: the fused attention backward at the step's masked shapes (joint encoder: 512 sequences of 328 tokens with the attention log; text-only:
128 sequences of 512) -- run once per library build (EXP_LIB=...) on the same box; scripts/gpu_r5_l.sh."""
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


os.environ['MERLOT_ATTN_PP'] = '1'
torch.manual_seed(0)
row = []
for name, B, S, nfix, log in (('joint + log', 512, 328, 200, True), ('joint', 512, 328, 200, False), ('text-only', 128, 512, 0, False), ('ViT, no workspace kernels', 2048, 198, 198, False)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = None
    if nfix < S:                                         # the first nfix tokens valid (vision), the others a caption of random length
        n = torch.randint(8, S - nfix + 1, (B, 1), device='cuda')
        valid = ((torch.arange(S, device='cuda')[None, :] < nfix + n)).to(torch.uint8).contiguous()
    o, lse = ops.attention_fwd(qkv, B, S, 12, valid)
    do = torch.randn_like(o)
    lo, hi = torch.zeros(B, S, device='cuda'), torch.zeros(B, S, device='cuda')
    if name.startswith('ViT'):
        os.environ['MERLOT_ATTN_PP'] = '0'
    kw = dict(log_lo=lo, log_hi=hi, log_split=200, log_weight=1 / 12) if log else {}
    t = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid, **kw))
    d = ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid, **kw).float()
    row.append(f'{name} {t:7.1f} us (|d| {float(d.abs().sum()):.6e})')
print(os.path.basename(os.environ.get('EXP_LIB', 'libmerlot_hip_exp.so')) + ': ' + ' | '.join(row), flush=True)
