"""Round 4: what the attention LOG costs inside the fused backward (joint pass of the bench: 512 sequences x 328 tokens, split 200; the
text-only shape for comparison), and what it saves in the forward.  AB_LIB=<other build> runs the same on another library."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_amd import lib  # noqa: E402
if os.environ.get('AB_LIB'):
    lib.LIB.path = os.path.abspath(os.environ['AB_LIB'])
    lib.LIB.check_abi = False
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402
from exp_attn_fallbacks import timeit  # noqa: E402

for B, S, split in ((512, 328, 200), (128, 512, 256)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = torch.ones(B, S, dtype=torch.uint8, device='cuda')
    valid[:, S - 9:] = 0
    o, lse = ops.attention_fwd(qkv, B, S, 12, valid)
    do = torch.randn_like(o)
    lo, hi = torch.zeros(B, S, device='cuda'), torch.zeros(B, S, device='cuda')
    t_f = timeit(lambda: ops.attention_fwd(qkv, B, S, 12, valid))
    t_fl = timeit(lambda: ops.attention_fwd(qkv, B, S, 12, valid, colsum_lo=lo, colsum_hi=hi, qsplit=split, valid_q_only=True, weight=1 / 12))
    t_b = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid))
    t_bl = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid, log_lo=lo, log_hi=hi, log_split=split, log_weight=1 / 12))
    print(f'{os.path.basename(lib.LIB.path or "libmerlot_hip.so"):22s} B {B} S {S}: fwd {t_f:7.1f} us, fwd + log {t_fl:7.1f} | bwd {t_b:7.1f} us, bwd + log {t_bl:7.1f} | '
          f'log in fwd costs {t_fl - t_f:6.1f}, in bwd {t_bl - t_b:6.1f}', flush=True)
