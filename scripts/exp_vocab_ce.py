"""Round 3: the MLM head's logits GEMM + softmax cross-entropy at the bench batch (12 800 masked rows x 50 370 classes) through
merlot_vocab_ce_fwd with different fp32 scratch sizes: the whole logits tensor (2.6 GB, one chunk) against cache-sized row chunks."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402
from merlot_amd.lib import call  # noqa: E402
from exp_attn_time import timeit  # noqa: E402

T, H, V, VP = 12800, 768, 50370, 50432
torch.manual_seed(0)
hb = (torch.randn(T, H, device='cuda') * 0.5).bfloat16()
wb = (torch.randn(VP, H, device='cuda') * 0.02).bfloat16()
bias = torch.zeros(V, device='cuda')
targets = torch.randint(100, V, (T,), device='cuda', dtype=torch.int32)
roww = torch.full((T,), 1.0 / T, device='cuda')
loss = torch.empty(T, device='cuda')
am = torch.empty(T, device='cuda', dtype=torch.int32)
dl = torch.empty(T, VP, device='cuda', dtype=torch.bfloat16)
ref = None
for mb in (2600, 160, 96, 160, 2600):
    scratch = torch.empty((mb << 20) // 4, device='cuda')

    def fn():
        call('merlot_vocab_ce_fwd', hb.data_ptr(), H, wb.data_ptr(), H, bias.data_ptr(), targets.data_ptr(), roww.data_ptr(), loss.data_ptr(),
             am.data_ptr(), dl.data_ptr(), VP, T, V, H, scratch.data_ptr(), scratch.numel() * 4, *ops._nt_ws(), ops._stream())
    fn()
    cur = (loss.clone(), am.clone(), dl.clone())
    ref = cur if ref is None else ref
    same = all(torch.equal(a, b) for a, b in zip(cur, ref))
    print(f'scratch {mb:5d} MiB ({(mb << 20) // (VP * 4):6d} rows per chunk): {timeit(fn, 10):8.1f} us   identical={same}', flush=True)
    del scratch
