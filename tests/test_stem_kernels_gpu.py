"""GPU: the HIP kernels of the ResNet-hybrid stem against their torch emulation (tests/emu_ops.py), which
the CPU host tests in turn check against the oracle.  bf16 outputs: rel-L2 <= 6e-3; im2col / col2im / pooling exact
up to one bf16 rounding of a sum."""
import pytest
import torch

import emu_ops as emu
from common import rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope='module')
def ops():
    from merlot_amd import ops as real
    return real


@pytest.mark.parametrize('N,H,W,C,stride', [(2, 8, 8, 32, 1), (3, 16, 12, 64, 2), (2, 64, 64, 3, 2), (1, 14, 14, 256, 1)])
def test_im2col_and_col2im(ops, N, H, W, C, stride):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, W, C, generator=g).to(BF16)
    shift = -0.5 if C == 3 else 0.0
    got = ops.im2col3x3(x.cuda(), stride, shift).cpu()
    ref = emu.im2col3x3(x, stride, shift)
    assert got.shape == ref.shape and torch.equal(got, ref)
    if C % 8 == 0:
        dp = torch.randn(ref.shape, generator=g).to(BF16)
        dx = ops.col2im3x3(dp.cuda(), N, H, W, C, stride).cpu()
        assert rel_l2(dx, emu.col2im3x3(dp, N, H, W, C, stride)) < 3e-3


@pytest.mark.parametrize('N,H,W,C,Co', [(2, 8, 8, 32, 32), (3, 12, 10, 64, 64), (1, 14, 14, 256, 256), (2, 9, 7, 128, 128),
                                        (1, 56, 56, 64, 64), (2, 5, 33, 32, 64), (1, 3, 3, 64, 128), (2, 1, 1, 32, 32),
                                        (1, 28, 28, 128, 256), (5, 7, 7, 96, 40), (1, 6, 6, 64, 512), (2, 4, 4, 512, 64)])
def test_conv3x3_implicit_gemm_is_the_explicit_path_bit_for_bit(ops, N, H, W, C, Co):
    """merlot_conv3x3_bf16 against merlot_im2col3x3 + merlot_gemm_bf16_nt (same K order, same MFMA sequence: equal bits) and
    against the emulation; then the layer's input gradient -- the same kernel on dY with flipped taps -- against the explicit
    dgrad (GEMM to [T, 9 C] in bf16, col2im sum): equal up to the bf16 rounding of the nine partial products."""
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + C)
    x = torch.randn(N, H, W, C, generator=g).to(BF16)
    Kp = (9 * C + 63) // 64 * 64
    w = torch.zeros(Co, Kp, dtype=BF16)
    w[:, :9 * C] = (torch.randn(Co, 9 * C, generator=g) / (3 * C ** 0.5)).to(BF16)
    y = ops.conv3x3(x.cuda(), w.cuda(), Co)
    y_explicit = ops.gemm_nt(ops.im2col3x3(x.cuda()), w.cuda()).view(N, H, W, Co)
    assert torch.equal(y, y_explicit)
    assert rel_l2(y.cpu(), emu.conv3x3(x, w, Co)) < 6e-3
    if Co % 32 == 0:
        dy = torch.randn(N, H, W, Co, generator=g).to(BF16)
        wdg = w[:, :9 * C].reshape(Co, 3, 3, C).flip(1, 2).permute(3, 1, 2, 0).reshape(C, 9 * Co).contiguous()
        dx = ops.conv3x3(dy.cuda(), wdg.cuda(), C).cpu()
        wT = torch.zeros(Kp, (Co + 63) // 64 * 64, dtype=BF16)
        wT[:9 * C, :Co] = w[:, :9 * C].t()
        dyp = torch.zeros(N * H * W, wT.shape[1], dtype=BF16)
        dyp[:, :Co] = dy.reshape(-1, Co)
        dx_explicit = emu.col2im3x3(emu.gemm_nt(dyp, wT), N, H, W, C)
        assert rel_l2(dx, dx_explicit) < 4e-3
        ref = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2),
                                                   w[:, :9 * C].float().reshape(Co, 3, 3, C).permute(0, 3, 1, 2), padding=1)
        assert rel_l2(dx, ref.permute(0, 2, 3, 1)) < 3e-3


@pytest.mark.parametrize('N,H,W,C,Co', [(2, 8, 8, 32, 32), (3, 12, 10, 64, 64), (1, 14, 14, 256, 256), (2, 9, 7, 128, 136),
                                        (4, 56, 56, 64, 64), (2, 5, 33, 32, 64), (1, 3, 3, 64, 128), (2, 1, 1, 32, 32),
                                        (16, 28, 28, 128, 256), (5, 7, 7, 96, 40), (1, 6, 6, 64, 512), (2, 4, 4, 512, 64)])
def test_conv3x3_implicit_weight_gradient(ops, N, H, W, C, Co):
    """merlot_conv3x3_wgrad_bf16 (no patch matrix; pixel counts that are not multiples of 32, images narrower than a K-step, ragged
    filter counts, many pixel ranges) against dy^T @ im2col(x) in fp32 and against the explicit HIP path; accumulate on top."""
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + C)
    x = torch.randn(N, H, W, C, generator=g).to(BF16)
    dy = torch.randn(N * H * W, Co, generator=g).to(BF16)
    Kp = (9 * C + 63) // 64 * 64
    ref = dy.float().t() @ emu.im2col3x3(x)[:, :9 * C].float()
    dw = torch.full((Co, Kp), 7.0, device='cuda')
    ops.conv3x3_wgrad(dy.cuda(), x.cuda(), dw)
    assert rel_l2(dw[:, :9 * C].cpu(), ref) < 2e-6
    assert torch.all(dw[:, 9 * C:] == 7.0)                       # padding columns untouched
    ex = torch.zeros((Co + Co % 2, Kp), device='cuda')
    ops.gemm_tn(dy.cuda(), ops.im2col3x3(x.cuda()), ex, accumulate=False, m=Co + Co % 2) if Co % 2 == 0 else None
    assert rel_l2(dw[:, :9 * C], ex[:Co, :9 * C]) < 2e-6
    ops.conv3x3_wgrad(dy.cuda(), x.cuda(), dw, accumulate=True)
    assert rel_l2(dw[:, :9 * C].cpu(), 2 * ref) < 2e-6


def test_conv3x3_rejects_what_the_kernel_cannot_take(ops):
    from merlot_amd.lib import MerlotHipError
    x = torch.zeros(1, 4, 4, 24, dtype=BF16).cuda()
    with pytest.raises(MerlotHipError, match='multiple of 32'):
        ops.conv3x3(x, torch.zeros(32, 9 * 24, dtype=BF16).cuda(), 32)


@pytest.fixture(params=[True, False], ids=['one-launch', 'two-launch'])
def gn_mode(ops, request):
    """both GroupNorm implementations in both directions: merlot_groupnorm_*_fused (ABI v10) and the two-launch entries; the product default (ops.GN_FUSED = 'fwd') is the
    one-launch forward with the two-launch backward"""
    was = ops.GN_FUSED
    ops.GN_FUSED = request.param
    yield request.param
    ops.GN_FUSED = was


# (the last four: several slices per sample with a ragged last one -- 1 760 positions of 64 channels = 4 forward / 7 backward slices --, a
# channel count whose 16-byte chunks do not divide 256 threads and whose groups of 3 channels straddle a thread's 8, one slice wider than the
# sample, the as-shipped geometry's largest per-sample tensor, and its largest layer with a residual -- the one-launch forward's late residual read: 33 slices of 16 positions
# per thread where holding the residual across the wait would make 66)
@pytest.mark.parametrize('N,H,W,C,relu,res', [(3, 8, 8, 32, True, False), (2, 14, 14, 256, False, False),
                                               (2, 7, 9, 1024, True, True), (4, 16, 16, 64, True, False),
                                               (3, 40, 44, 64, True, True), (2, 10, 11, 96, True, False), (5, 3, 3, 256, False, True),
                                               (2, 96, 176, 64, True, False), (2, 48, 88, 256, True, True)])
def test_groupnorm_forward_backward(ops, gn_mode, N, H, W, C, relu, res):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(N, H, W, C, generator=g) * 2 + 0.3).to(BF16)
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    r = torch.randn(N, H, W, C, generator=g).to(BF16) if res else None
    y, stats = ops.groupnorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), res=r.cuda() if res else None, relu=relu)
    y_ref, stats_ref = emu.groupnorm_fwd(x, gamma, beta, res=r, relu=relu)
    assert rel_l2(stats.cpu(), stats_ref) < 1e-5
    assert rel_l2(y.cpu(), y_ref) < 6e-3
    dy = torch.randn(N, H, W, C, generator=g).to(BF16)
    dga, dbe = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    dx, dres = ops.groupnorm_bwd(dy.cuda(), y, x.cuda(), stats, gamma.cuda(), dga, dbe, relu=relu, want_dres=res)
    dga_r, dbe_r = torch.zeros(C), torch.zeros(C)
    dx_ref, dres_ref = emu.groupnorm_bwd(dy, y.cpu(), x, stats.cpu(), gamma, dga_r, dbe_r, relu=relu, want_dres=res)
    assert rel_l2(dx.cpu(), dx_ref) < 8e-3
    assert rel_l2(dga.cpu(), dga_r) < 2e-3 and rel_l2(dbe.cpu(), dbe_r) < 2e-3
    if res:
        assert torch.equal(dres.cpu(), dres_ref)
    if relu and not res:
        # the product's path for such layers: no y, the ReLU mask recomputed from x with the forward's own expression -- the SAME
        # gradients as with the stored output (a pre-activation within rounding of 0 could flip: none does on these inputs)
        dga2, dbe2 = torch.zeros(C).cuda(), torch.zeros(C).cuda()
        dx2, _ = ops.groupnorm_bwd(dy.cuda(), None, x.cuda(), stats, gamma.cuda(), dga2, dbe2, beta=beta.cuda(), relu=True)
        assert rel_l2(dx2, dx) < 1e-4 and rel_l2(dga2, dga) < 1e-5 and rel_l2(dbe2, dbe) < 1e-5      # (sums by atomics: last bits)
        with pytest.raises(Exception):                      # neither y nor beta: no way to form the mask
            ops.groupnorm_bwd(dy.cuda(), None, x.cuda(), stats, gamma.cuda(), dga2, dbe2, relu=True)


def test_avgpool2(ops):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 12, 8, 64, generator=g).to(BF16)
    assert rel_l2(ops.avgpool2_fwd(x.cuda()).cpu(), emu.avgpool2_fwd(x)) < 3e-3
    dy = torch.randn(3, 6, 4, 64, generator=g).to(BF16)
    assert torch.equal(ops.avgpool2_bwd(dy.cuda()).cpu(), emu.avgpool2_bwd(dy))


@pytest.mark.parametrize('kh,ci,co', [(3, 3, 32), (1, 64, 256), (3, 64, 64), (1, 256, 100)])
def test_weight_standardisation_kernels(kh, ci, co):
    """merlot_weight_std_fwd / _bwd (utils/vision_transformer.py:52-56) against the torch expression they replaced, and the
    backward against autograd of that expression."""
    from merlot_amd import ops
    g = torch.Generator().manual_seed(kh * 100 + co)
    w = (torch.randn(kh, kh, ci, co, generator=g) * 0.3 + 0.1)
    K = kh * kh * ci
    Kp, Cop = (K + 63) // 64 * 64, (co + 63) // 64 * 64
    k2 = w.reshape(K, co).clone().requires_grad_(True)
    mean = k2.mean(0, keepdim=True)
    khat_ref = (k2 - mean) * torch.rsqrt(((k2 - mean) ** 2).mean(0, keepdim=True) + 1e-5)
    khat, rstd, wb, wbT = ops.weight_std_fwd(w.reshape(K, co).cuda().contiguous(), Kp, Cop)
    assert (khat.cpu() - khat_ref.detach()).abs().max().item() < 2e-5
    assert torch.equal(wb[:, :K].float().cpu(), khat.cpu().t().to(torch.bfloat16).float())
    assert torch.equal(wbT[:K, :co].float().cpu(), khat.cpu().to(torch.bfloat16).float())
    assert float(wb[:, K:].float().abs().sum()) == 0.0 and float(wbT[K:].float().abs().sum()) == 0.0
    dkh = torch.randn(K, co, generator=g)
    (khat_ref * dkh).sum().backward()
    dkt = torch.zeros(co + (co % 2), Kp)
    dkt[:co, :K] = dkh.t()
    gk = torch.full((K, co), 0.5).cuda()
    ops.weight_std_bwd(dkt.cuda(), khat, rstd, gk)
    assert (gk.cpu() - 0.5 - k2.grad).abs().max().item() < 1e-4 * max(1.0, float(k2.grad.abs().max()))


def test_batched_weight_standardisation_is_the_per_kernel_launch_bit_for_bit(ops):
    """merlot_weight_std_fwd_batched / _bwd_batched (every kernel of a stem in one launch) against one merlot_weight_std_fwd / _bwd
    launch per kernel: equal bits; the flipped-tap input-gradient operand wdg against torch's flip / permute of wbT."""
    g = torch.Generator().manual_seed(5)
    shapes = [(3, 3, 3, 32), (3, 3, 32, 32), (1, 1, 64, 256), (3, 3, 64, 64), (1, 1, 256, 72), (3, 3, 32, 40)]
    offs, o = [], 0
    for shp in shapes:
        n = shp[0] * shp[1] * shp[2] * shp[3]
        offs.append(o)
        o += (n + 63) // 64 * 64
    master = torch.randn(o, generator=g).cuda()
    fj, bj, meta = [], [], []
    q = dict(khat=0, rstd=0, wb=0, wbT=0, wdg=0, dk=0)
    blocks = 0
    for (kh, kw, ci, co), off in zip(shapes, offs):
        K = kh * kw * ci
        Kp, Cop, co2 = (K + 63) // 64 * 64, (co + 63) // 64 * 64, co + co % 2
        dg = q['wdg'] if kh == 3 and ci % 8 == 0 else -1
        meta.append((K, co, Kp, Cop, ci, co2, dict(q), dg, off))
        fj.append([off, K, co, q['khat'], q['rstd'], q['wb'], Kp, q['wbT'], Cop, dg, ci, blocks])
        bj.append([q['dk'], Kp, q['khat'], q['rstd'], K, co, off, blocks])
        blocks += (co + 15) // 16
        q['khat'] += K * co; q['rstd'] += (co + 7) // 8 * 8; q['wb'] += co * Kp; q['wbT'] += Kp * Cop; q['dk'] += co2 * Kp
        if dg >= 0:
            q['wdg'] += 9 * ci * co
    khat = torch.zeros(q['khat']).cuda(); rstd = torch.zeros(q['rstd']).cuda()
    wb = torch.zeros(q['wb'], dtype=BF16).cuda(); wbT = torch.zeros(q['wbT'], dtype=BF16).cuda(); wdg = torch.zeros(q['wdg'], dtype=BF16).cuda()
    ops.weight_std_fwd_batched(master, torch.tensor(fj).cuda(), blocks, khat, rstd, wb, wbT, wdg)
    dk = torch.randn(q['dk'], generator=g).cuda()
    grad = torch.randn(master.shape, generator=g).cuda()
    grad_ref = grad.clone()
    ops.weight_std_bwd_batched(dk, torch.tensor(bj).cuda(), blocks, khat, rstd, grad)
    for (K, co, Kp, Cop, ci, co2, oo, dg, off) in meta:
        kh1, rs1, wb1, wbT1 = ops.weight_std_fwd(master[off:off + K * co].view(K, co), Kp, Cop)
        assert torch.equal(khat[oo['khat']:oo['khat'] + K * co].view(K, co), kh1) and torch.equal(rstd[oo['rstd']:oo['rstd'] + co], rs1)
        assert torch.equal(wb[oo['wb']:oo['wb'] + co * Kp].view(co, Kp), wb1) and torch.equal(wbT[oo['wbT']:oo['wbT'] + Kp * Cop].view(Kp, Cop), wbT1)
        if dg >= 0:
            ref = wbT1[:K, :co].reshape(3, 3, ci, co).flip(0, 1).permute(2, 0, 1, 3).reshape(ci, 9 * co)
            assert torch.equal(wdg[dg:dg + 9 * ci * co].view(ci, 9 * co), ref)
        gk = grad_ref[off:off + K * co].view(K, co)
        ops.weight_std_bwd(dk[oo['dk']:oo['dk'] + co2 * Kp].view(co2, Kp), kh1, rs1, gk)
        assert torch.equal(grad[off:off + K * co].view(K, co), gk)


def test_groupnorm_one_launch_agrees_with_two_launches_and_repeats(ops):
    """the two implementations evaluate the same expressions; only the order of the moment sums differs (atomics): statistics to 1e-6, outputs equal
    up to a rare last-place flip -- and the one-launch form's workspace protocol leaves nothing behind (five calls in a row, same results)."""
    g = torch.Generator().manual_seed(7)
    N, H, W, C = 6, 48, 88, 256
    x = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.2).to(BF16).cuda()
    r = torch.randn(N, H, W, C, generator=g).to(BF16).cuda()
    dy = torch.randn(N, H, W, C, generator=g).to(BF16).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    was = ops.GN_FUSED
    try:
        out = {}
        for mode in (False, True, True, True, True, True):
            ops.GN_FUSED = mode
            y, stats = ops.groupnorm_fwd(x, gamma, beta, res=r, relu=True)
            dga, dbe = torch.zeros(C).cuda(), torch.zeros(C).cuda()
            dx, dres = ops.groupnorm_bwd(dy, y, x, stats, gamma, dga, dbe, relu=True, want_dres=True)
            cur = (y, stats, dx, dres, dga, dbe)
            if mode not in out:
                out[mode] = cur
            else:
                for a, b in zip(cur, out[mode]):
                    assert rel_l2(a, b) < 1e-4             # (moment sums by atomics: a last-place flip of a bf16 output here and there; measured 1.6e-5)
        y0, st0, dx0, dr0, dg0, db0 = out[False]
        y1, st1, dx1, dr1, dg1, db1 = out[True]
        assert rel_l2(st1, st0) < 1e-6 and rel_l2(dg1, dg0) < 1e-5 and rel_l2(db1, db0) < 1e-5
        assert (y1 != y0).float().mean() < 1e-4 and rel_l2(y1, y0) < 1e-4
        assert torch.equal(dr1, dr0) or (dr1 != dr0).float().mean() < 1e-4
        assert rel_l2(dx1, dx0) < 1e-4
    finally:
        ops.GN_FUSED = was


def test_groupnorm_one_launch_rejects_a_short_workspace(ops):
    from merlot_amd.lib import MerlotHipError, call
    x = torch.zeros(2, 4, 4, 64, dtype=BF16).cuda()
    y, stats = torch.empty_like(x), torch.empty(2, 32, 2).cuda()
    gam = torch.ones(64).cuda()
    ws = torch.zeros(8, dtype=torch.int32).cuda()
    with pytest.raises(MerlotHipError, match='workspace'):
        call('merlot_groupnorm_fwd_fused', x.data_ptr(), gam.data_ptr(), gam.data_ptr(), None, y.data_ptr(), stats.data_ptr(), 2, 4, 4, 64, 32, 1e-4, 1,
             ws.data_ptr(), ws.numel() * 4, None)


def test_groupnorm_one_launch_refuses_more_slices_than_were_measured(ops):
    """csrc/conv.hip GN_FUSED_MAX_SLICES = 40: the one-launch backward at 66 slices per sample had calls that never returned at 896 frames (profiles/r06_z6_gn_stress.txt); both
    entries refuse such a shape, and ops.groupnorm_* takes the two-launch entries for it (same results as ever: test_groupnorm_forward_backward's 96 x 176 x 64 case)."""
    from merlot_amd.lib import LIB, MerlotHipError, call
    N, H, W, C = 1, 96, 176, 64                            # backward: 16 896 positions / (8 x 32 per slice) = 66 slices
    assert ops._gn_fused_slices(H * W, C, False, bwd=True) == 66 and ops._gn_fused_slices(H * W, C, False) == 33
    x = torch.zeros(N, H, W, C, dtype=BF16).cuda()
    dx, stats, gsum = torch.empty_like(x), torch.zeros(N, 32, 2).cuda(), torch.empty(N, 32, 2).cuda()
    gam, dga, dbe = torch.ones(C).cuda(), torch.zeros(C).cuda(), torch.zeros(C).cuda()
    ws = torch.zeros(LIB.query('merlot_groupnorm_bwd_fused_workspace_bytes', N, H, W, C, 32) // 4, dtype=torch.int32).cuda()
    with pytest.raises(MerlotHipError, match='slices per sample'):
        call('merlot_groupnorm_bwd_fused', x.data_ptr(), None, x.data_ptr(), stats.data_ptr(), gam.data_ptr(), gam.data_ptr(), dga.data_ptr(), dbe.data_ptr(), gsum.data_ptr(),
             dx.data_ptr(), None, N, H, W, C, 32, 1e-4, 1, ws.data_ptr(), ws.numel() * 4, None)
    H2 = 4 * H                                             # forward: 67 584 positions / (16 x 32) = 132 slices
    x2 = torch.zeros(N, H2, W, C, dtype=BF16).cuda()
    ws2 = torch.zeros(LIB.query('merlot_groupnorm_fused_workspace_bytes', N, C, 32) // 4, dtype=torch.int32).cuda()
    with pytest.raises(MerlotHipError, match='slices per sample'):
        call('merlot_groupnorm_fwd_fused', x2.data_ptr(), gam.data_ptr(), gam.data_ptr(), None, torch.empty_like(x2).data_ptr(), stats.data_ptr(), N, H2, W, C, 32, 1e-4, 1,
             ws2.data_ptr(), ws2.numel() * 4, None)
    y, st = ops.groupnorm_fwd((torch.randn(N, H2, W, C, device='cuda') * 2).to(BF16), gam, dbe)     # the product's route for it: two launches, no error
    assert torch.isfinite(y.float()).all() and torch.isfinite(st).all()
