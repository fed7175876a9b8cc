"""Timing rows of the attention kernels BESIDE the step's main path: what BASELINE config #5's sequence lengths
and the `disable_pairwise_lang_attn` segment mask cost on the tiled kernels (forward: attn_fwd_kernel; backward: attn_bwd_dq_kernel +
attn_bwd_dkdv_kernel -- everything the fused / resident kernels do not take: S > 512, a segment mask, S <= 64), product library."""
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


if __name__ == '__main__':
    rows = [  # B, S, masked, segment mask (P vision tokens, Lc tokens per caption), what
        (768, 578, False, None, 'config #5 ViT pass (384^2: 578 tokens/frame), 48 examples x 16 frames'),
        (48, 2832, True, None, 'config #5 joint pass (16 x (1 + 144 + 32) tokens), 48 groups'),
        (48, 512, True, None, 'config #5 text-only pass (fused / resident kernels: the main path, for comparison)'),
        (512, 328, True, (200, 32), 'config #2 joint pass under disable_pairwise_lang_attn (segment mask -> tiled kernels)'),
        (512, 328, True, None, 'config #2 joint pass (fused / resident kernels: the main path, for comparison)'),
        (8192, 32, True, None, 'S = 32 (<= 64: tiled kernels)'),
    ]
    for B, S, masked, seg, what in rows:
        qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
        valid = torch.ones(B, S, dtype=torch.uint8, device='cuda') if masked else None
        sg = None
        if seg is not None:
            P, Lc = seg
            sg = torch.cat([torch.zeros(P, dtype=torch.int32), 1 + torch.arange(S - P, dtype=torch.int32) // Lc]).cuda()
        o, lse = ops.attention_fwd(qkv, B, S, 12, valid, seg=sg)
        do = torch.randn_like(o)
        tf = timeit(lambda: ops.attention_fwd(qkv, B, S, 12, valid, seg=sg))
        tb = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, valid, seg=sg))
        fl = 4.0 * S * S * 768 * B
        gb = B * S * 768 * 2 * 4 / 1e9                        # forward: Q, K, V in, O out
        print(f'B {B:5d} S {S:5d} masked {masked!s:5s} seg {seg is not None!s:5s}: fwd {tf:8.1f} us {fl / tf * 1e-6:5.0f} TF {gb / tf * 1e3:5.2f} TB/s | '
              f'bwd {tb:8.1f} us {2.5 * fl / tb * 1e-6:5.0f} TF {2 * gb / tb * 1e3:5.2f} TB/s | {what}', flush=True)
