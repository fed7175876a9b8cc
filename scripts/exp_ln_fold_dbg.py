"""debug: which rows of the fused LayerNorm differ from the two-launch composition, and how (round 6)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
from merlot_amd import ops  # noqa: E402
from test_gemm_ln_gpu import rnd  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
M, K, N = int(os.environ.get('M', 65536)), int(os.environ.get('K', 768)), 768
P = float(os.environ.get('P', 0.0))
a, w = rnd((M, K), 11), rnd((N, K), 12, 0.03)
bias = rnd((N,), 13, 0.1, F32)
res = rnd((M, N), 14)
gamma, beta = torch.ones(N, device='cuda'), torch.zeros(N, device='cuda')
h0 = ops.gemm_nt(a, w, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=P, dropout_seed=77)
y0, _, m0, r0 = ops.ln_fwd(h0, gamma, beta)
for it in range(2):
    h1, y1, m1, r1 = ops.gemm_nt_ln(a, w, gamma, beta, bias=bias, aux_in=res, dropout_p=P, dropout_seed=77)
    torch.cuda.synchronize()
    print('M', M, 'K', K, 'p', P, 'iteration', it, 'h equal', torch.equal(h0, h1), ' y: non-finite', int((~torch.isfinite(y1.float())).sum()), ' max |y1 - y0|', float((y1.float() - y0.float()).abs().nan_to_num(1e9).max()),
          ' elements that differ', int((y1 != y0).sum()), 'of', y1.numel())
    badm = ((m1 - m0).abs() > 1e-4) | ~torch.isfinite(m1)
    badr = ((r1 - r0).abs() / r0 > 1e-3) | ~torch.isfinite(r1)
    bady = (~torch.isfinite(y1.float())).any(1) | ((y1.float() - y0.float()).abs().amax(1) > 0.1)
    print('rows with bad mean', int(badm.sum()), 'bad rstd', int(badr.sum()), 'bad y', int(bady.sum()), 'of', M)
    for name, bad in (('mean', badm), ('rstd', badr), ('y', bady)):
        idx = bad.nonzero().flatten()
        if idx.numel():
            blocks = torch.unique(idx // 256)
            print(f'  {name}: row blocks affected {blocks.numel()} of {M // 256}; first blocks {blocks[:12].tolist()}; rows within the first bad block: '
                  f'{(idx[idx // 256 == blocks[0]] % 256)[:40].tolist()}')
            r = int(idx[0])
            print(f'    row {r}: mean {float(m1[r]):.5f} vs {float(m0[r]):.5f}   rstd {float(r1[r]):.5f} vs {float(r0[r]):.5f}   y[:4] {y1[r, :4].float().tolist()} vs {y0[r, :4].float().tolist()}')
    # the partial statistics the tiles left (workspace layout: counters, then [12][Mpad] (sum, M2))
    ws = [v for k, v in ops._LN_WS.items() if k[2] == M][0]
    nblk = M // 256
    ctr_words = ((nblk * 4 + 255) // 256 * 256) // 4
    part = ws[ctr_words:ctr_words + 12 * M * 2].view(torch.float32).view(12, M, 2)
    hs = h1.float().view(M, 12, 64)
    s_ref = hs.sum(2).t()
    q_ref = ((hs - hs.mean(2, keepdim=True)) ** 2).sum(2).t()
    ds = (part[:, :, 0] - s_ref).abs().amax(0)
    dq = ((part[:, :, 1] - q_ref).abs() / (q_ref + 1e-3)).amax(0)
    print('  partials: rows whose segment sums are off', int((ds > 1e-2).sum()), ' M2 off', int((dq > 1e-3).sum()), ' counters non-zero', int(ws[:nblk].abs().sum()))
    bp = (ds > 1e-2).nonzero().flatten()
    if bp.numel():
        r = int(bp[0])
        print('    first row with bad partial sums', r, 'segments', ((part[:, r, 0] - s_ref[:, r]).abs() > 1e-2).nonzero().flatten().tolist(), part[:, r, 0].tolist()[:4], s_ref[:, r].tolist()[:4])
