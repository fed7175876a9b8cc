"""Round 5 debugging aid: fused / persistent backward launches in a row on one shape, each against fp32."""
import _exp_lib  # noqa: F401
import os
import sys
import torch
from merlot_amd import ops

B, S = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
os.environ['MERLOT_ATTN_PP'] = '0'
o, lse = ops.attention_fwd(qkv, B, S, 12, None)
do = torch.randn_like(o)
x = qkv.float().requires_grad_(True)
q_, k_, v_ = [t.reshape(B, S, 12, 64).permute(0, 2, 1, 3) for t in x.split(768, dim=1)]
oo = (torch.softmax(q_ @ k_.transpose(-1, -2) * 0.125, -1) @ v_).permute(0, 2, 1, 3).reshape(B * S, 768)
oo.backward(do.float())
g = x.grad
torch.cuda.synchronize()
for k in sys.argv[3]:
    os.environ['MERLOT_ATTN_PP'] = k
    r = ops.attention_bwd(qkv, o, do, lse, B, S, 12, None)
    torch.cuda.synchronize()
    r = r.float()
    e = torch.nan_to_num((r - g).abs(), nan=1e9).view(B, S, 3, 12, 64).amax(dim=(1, 4))
    print('pp' if k == '1' else 'fused', 'nan', int(torch.isnan(r).sum()), 'bad (batch, third, head):', [tuple(int(v) for v in i) for i in torch.nonzero(e > 0.01)][:12])
