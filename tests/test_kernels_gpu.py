"""GPU parity of every HIP kernel (through the C-ABI via merlot_amd.ops) against the plain torch-CPU fp32
restatement of the same op in tests/emu_ops.py, on seeded inputs.  Tolerances (bf16 in/out, fp32 accumulate):
rel-L2 <= 6e-3 for bf16 outputs, <= 2e-3 for fp32 outputs of bf16 inputs; integer outputs exact."""
import math

import numpy as np
import pytest
import torch

import emu_ops as E
import merlot_amd.ops as ops_mod
from common import rel_l2

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from merlot_amd import ops as o
    from merlot_amd.lib import LIB
    LIB.load()          # fail loudly if libmerlot_hip.so is missing
    return o


def rnd(shape, gen, scale=1.0, dtype=BF16):
    return (torch.randn(shape, generator=gen) * scale).to(dtype)


def dev(*ts):
    return [None if t is None else t.cuda() for t in ts]


# ---- hardware layout probes: pin the lane maps the kernels assume ---------------------------------------
def test_probe_mfma32_layout(ops):
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-3, 4, (32, 16), generator=g).float()      # asymmetric small ints: exact in bf16
    Bm = torch.randint(-3, 4, (16, 32), generator=g).float()
    lanes = torch.arange(64)
    a = torch.stack([A[lanes & 31, 8 * (lanes >> 5) + j] for j in range(8)], 1).to(BF16)     # A[l&31][8(l>>5)+j]
    b = torch.stack([Bm[8 * (lanes >> 5) + j, lanes & 31] for j in range(8)], 1).to(BF16)    # B[8(l>>5)+j][l&31]
    import probe_lib
    d = probe_lib.probe_mfma32(a.cuda().contiguous(), b.cuda().contiguous()).cpu()
    ref = A @ Bm
    got = torch.zeros(32, 32)
    for l in range(64):
        for r in range(16):
            got[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = d[l, r]
    assert torch.equal(got, ref), "MFMA 32x32x16 operand/accumulator lane map differs from the one the kernels assume"


def test_probe_tr16_layout(ops):
    tile = torch.arange(256).to(BF16)          # 0..255 exactly representable
    import probe_lib
    out = probe_lib.probe_tr16(tile.cuda()).cpu().float().reshape(64, 4)
    exp = torch.zeros(64, 4)
    for l in range(64):
        gi, i = l >> 4, l & 15
        for j in range(4):
            exp[l, j] = gi * 64 + j * 16 + i   # column i of the 4x16 row-major block of lane group gi
    assert torch.equal(out, exp), f"ds_read_b64_tr_b16 gather differs from the assumed map:\n{out[:20]}"


# ---- GEMM ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (200, 130, 128), (4000, 768, 768), (37, 4, 768),
                                   (1024, 2304, 768), (512, 768, 3072)])
def test_gemm_nt_plain(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a, bt = rnd((M, K), g), rnd((N, K), g, 0.05)
    bias = torch.randn(N, generator=g) * 0.1
    ref = E.gemm_nt(a, bt, bias=bias, out_dtype=F32)
    got = ops.gemm_nt(*dev(a, bt), bias=bias.cuda())
    assert rel_l2(got, ref) < 6e-3
    got32 = ops.gemm_nt(*dev(a, bt), bias=bias.cuda(), out_dtype=F32, alpha=0.5)
    assert rel_l2(got32, E.gemm_nt(a, bt, bias=bias, out_dtype=F32, alpha=0.5)) < 2e-5 * math.sqrt(K) + 1e-4


def test_gemm_nt_epilogues(ops):
    g = torch.Generator().manual_seed(3)
    M, N, K = 300, 768, 768
    a, bt = rnd((M, K), g), rnd((N, K), g, 0.05)
    bias = torch.randn(N, generator=g) * 0.1
    res = rnd((M, N), g)
    # gelu (+ pre-activation side output)
    u_ref = torch.empty((M, N), dtype=BF16)
    ref = E.gemm_nt(a, bt, bias=bias, epilogue=E.EPI_GELU, aux_out=u_ref, out_dtype=F32)
    u = torch.empty((M, N), dtype=BF16).cuda()
    got = ops.gemm_nt(*dev(a, bt), bias=bias.cuda(), epilogue=ops.EPI_GELU, aux_out=u)
    assert rel_l2(got, ref) < 6e-3 and rel_l2(u, u_ref) < 6e-3
    # residual
    ref = E.gemm_nt(a, bt, bias=bias, epilogue=E.EPI_RESIDUAL, aux_in=res, out_dtype=F32)
    got = ops.gemm_nt(*dev(a, bt), bias=bias.cuda(), epilogue=ops.EPI_RESIDUAL, aux_in=res.cuda())
    assert rel_l2(got, ref) < 6e-3
    # dgelu
    ref = E.gemm_nt(a, bt, epilogue=E.EPI_DGELU, aux_in=res, out_dtype=F32)
    got = ops.gemm_nt(*dev(a, bt), epilogue=ops.EPI_DGELU, aux_in=res.cuda())
    assert rel_l2(got, ref) < 6e-3
    # f32 accumulate into a padded-ld output
    out = torch.ones((M, 772), dtype=F32).cuda()
    ops.gemm_nt(*dev(a, bt), out=out, accumulate=True, n=N)
    ref = 1.0 + E.gemm_nt(a, bt, out_dtype=F32)
    assert rel_l2(out[:, :N], ref) < 1e-3 and torch.all(out[:, N:] == 1.0)


def test_gemm_nt_dropout_statistics(ops):
    g = torch.Generator().manual_seed(5)
    M, N, K = 512, 768, 64
    a, bt = rnd((M, K), g), rnd((N, K), g)
    res = torch.zeros((M, N), dtype=BF16)
    full = ops.gemm_nt(*dev(a, bt), epilogue=ops.EPI_RESIDUAL, aux_in=res.cuda()).float()
    d1 = ops.gemm_nt(*dev(a, bt), epilogue=ops.EPI_RESIDUAL, aux_in=res.cuda(), dropout_p=0.1, dropout_seed=77).float()
    d2 = ops.gemm_nt(*dev(a, bt), epilogue=ops.EPI_RESIDUAL, aux_in=res.cuda(), dropout_p=0.1, dropout_seed=77).float()
    assert torch.equal(d1, d2)                                     # counter-based: reproducible
    kept = d1 != 0
    keep_rate = kept.float().mean().item()
    assert abs(keep_rate - 0.9) < 5e-3
    assert rel_l2(d1[kept], full[kept] / 0.9) < 1e-2               # survivors scaled by 1/(1-p)
    # the standalone mask kernel regenerates the same mask (used by the backward)
    y = ops.dropout_apply(torch.ones((M, N), dtype=BF16).cuda(), 0.1, 77).float()
    assert torch.equal(y != 0, kept | (full == 0))


@pytest.mark.parametrize("R,M,N", [(64, 128, 128), (1000, 768, 768), (777, 2304, 768), (300, 50370, 768), (5000, 768, 3072),
                                   (130, 4, 768), (4096, 3072, 768), (50, 768, 768),
                                   # the ping-pong TN kernel (R >= 4096) on ragged tiles, a reduction tail (R % 64 != 0) and N % 4 == 2
                                   (4500, 1000, 770), (8192, 130, 258), (4096, 256, 128), (20000, 2304, 768)])
def test_gemm_tn(ops, R, M, N):
    g = torch.Generator().manual_seed(R + M)
    a, b = rnd((R, M), g), rnd((R, N), g)
    ref = E.gemm_tn(a, b, torch.zeros((M, N)), accumulate=False)
    out = torch.full((M, N), 7.0).cuda()
    ops.gemm_tn(*dev(a, b), out, accumulate=False)
    assert rel_l2(out, ref) < 2e-3
    ops.gemm_tn(*dev(a, b), out, accumulate=True, alpha=2.0)
    assert rel_l2(out, 3 * ref) < 2e-3


def test_gemm_tn_padded_leading_dims(ops):
    """the LM-head shape: A = dlogits [T, V] living in a [T, Vpad] buffer, output rows limited to V."""
    g = torch.Generator().manual_seed(9)
    R, V, Vpad, N = 200, 1002, 1024, 768
    abuf, b = rnd((R, Vpad), g), rnd((R, N), g)
    ref = E.gemm_tn(abuf, b, torch.zeros((V, N)), accumulate=False, m=V)
    out = torch.zeros((V, N)).cuda()
    ops.gemm_tn(abuf.cuda(), b.cuda(), out, accumulate=False, m=V)
    assert rel_l2(out, ref) < 2e-3


def test_patch_embed(ops):
    g = torch.Generator().manual_seed(11)
    for (n, H, W) in [(3, 64, 64), (2, 224, 224), (1, 192, 352)]:
        img = torch.rand((n, H, W, 3), generator=g).to(BF16)
        wt = rnd((768, 768), g, 0.03)
        bias = torch.randn(768, generator=g) * 0.1
        ref, pref = E.patch_embed_fwd(img, wt, bias, 16)
        got, patches = ops.patch_embed_fwd(img.cuda(), wt.cuda(), bias.cuda(), 16)
        assert torch.equal(patches.cpu(), pref)                      # im2col(image - 0.5) is bit-exact
        assert rel_l2(got, ref.float()) < 6e-3
        dy = rnd((ref.shape[0], 768), g)
        dref = torch.zeros((768, 768))
        E.patch_embed_wgrad(pref, dy, dref, accumulate=False)
        dw = torch.zeros((768, 768)).cuda()
        ops.patch_embed_wgrad(patches, dy.cuda(), dw, accumulate=False)
        assert rel_l2(dw, dref) < 2e-3
        ops.patch_embed_wgrad(patches, dy.cuda(), dw, accumulate=True)
        assert rel_l2(dw, 2 * dref) < 2e-3


@pytest.mark.parametrize("P,H,W", [(8, 64, 64), (32, 128, 192), (24, 96, 48)])
def test_patch_embed_other_patch_sizes(ops, P, H, W):
    """`patch_size` other than merlot.yaml's 16 (utils/vision_transformer.py:196-205 takes any): K = 3 P^2 = 192 / 3072 / 1728."""
    g = torch.Generator().manual_seed(P)
    img = torch.rand((3, H, W, 3), generator=g).to(BF16)
    K = 3 * P * P
    wt = rnd((768, K), g, 0.03)
    bias = torch.randn(768, generator=g) * 0.1
    ref, pref = E.patch_embed_fwd(img, wt, bias, P)
    got, patches = ops.patch_embed_fwd(img.cuda(), wt.cuda(), bias.cuda(), P)
    assert torch.equal(patches.cpu(), pref)
    assert rel_l2(got, ref.float()) < 6e-3
    dy = rnd((ref.shape[0], 768), g)
    dref = torch.zeros((768, K))
    E.patch_embed_wgrad(pref, dy, dref, accumulate=False)
    dw = torch.zeros((768, K)).cuda()
    ops.patch_embed_wgrad(patches, dy.cuda(), dw, accumulate=False)
    assert rel_l2(dw, dref) < 2e-3
    with pytest.raises(Exception):                                   # 3 * 12 * 12 = 432 is not a multiple of 64
        ops.patch_embed_fwd(torch.rand(1, 48, 48, 3).to(BF16).cuda(), rnd((768, 432), g, 0.03).cuda(), bias.cuda(), 12)


@pytest.mark.parametrize("rows,H,nid", [(65536, 768, 20000), (5000, 768, 3), (4097, 64, 4097), (70001, 256, 50370)])
def test_scatter_add_rows_sorted_path(ops, rows, H, nid):
    """large scatters (the word-embedding gradient) are sorted by index and reduced per run of equal indices
    (merlot_scatter_add_sorted): runs inside a block's range, runs across block boundaries (nid = 3: every run spans many blocks),
    unique indices, skipped negative indices, a row count that is not a multiple of the block's 32 -- against index_add_ in fp64;
    the table is ACCUMULATED into."""
    g = torch.Generator().manual_seed(rows + nid)
    idx = torch.randint(0, nid, (rows,), generator=g, dtype=torch.int32)
    idx[torch.rand(rows, generator=g) < 0.01] = -1
    src = torch.randn((rows, H), generator=g)
    tab0 = torch.randn((nid, H), generator=g)
    want = tab0.double()
    keep = idx >= 0
    want.index_add_(0, idx[keep].long(), src[keep].double())
    tab = tab0.clone().cuda()
    ops.scatter_add_rows(src.cuda(), idx.cuda(), tab)
    assert float((tab.cpu().double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    tab2 = tab0.clone().cuda()                              # deterministic where no run crosses a block boundary; always close
    ops.scatter_add_rows(src.cuda(), idx.cuda(), tab2)
    assert float((tab2 - tab).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))


# ---- LayerNorm ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,H,xf32", [(1, 768, False), (1000, 768, False), (333, 768, True), (64, 1024, False), (50, 256, True)])
def test_layernorm(ops, rows, H, xf32):
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn((rows, H), generator=g) * 2 + 0.5).to(F32 if xf32 else BF16)
    gamma, beta = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    y16, y32, mean, rstd = E.ln_fwd(x, gamma, beta, out_bf16=True, out_f32=True)
    g16, g32, gm, gr = ops.ln_fwd(x.cuda(), gamma.cuda(), beta.cuda(), out_bf16=True, out_f32=True)
    assert rel_l2(g32, y32) < 1e-5 and rel_l2(g16, y32) < 4e-3 and rel_l2(gm, mean) < 1e-5 and rel_l2(gr, rstd) < 1e-5
    dy = torch.randn((rows, H), generator=g).to(x.dtype)
    dres = torch.randn((rows, H), generator=g).to(x.dtype)
    dg_r, db_r = torch.zeros(H), torch.zeros(H)
    dx_r = E.ln_bwd(dy, x, mean, rstd, gamma, dg_r, db_r, dres=dres, dx_dtype=F32)
    dg, db = torch.zeros(H).cuda(), torch.zeros(H).cuda()
    dx = ops.ln_bwd(dy.cuda(), x.cuda(), gm, gr, gamma.cuda(), dg, db, dres=dres.cuda())
    tol = 1e-4 if xf32 else 5e-3
    assert rel_l2(dx, dx_r) < tol and rel_l2(dg, dg_r) < 1e-4 and rel_l2(db, db_r) < 1e-4


def test_layernorm_bwd_fused_branch_gradient(ops):
    """the fused tail of ln_bwd == separate dropout_apply + colsum passes on its dx output (same counter mask)."""
    g = torch.Generator().manual_seed(21)
    rows, H = 3000, 768
    x, dy, dres = rnd((rows, H), g), rnd((rows, H), g), rnd((rows, H), g)
    gamma, beta = 1 + 0.1 * torch.randn(H, generator=g), torch.zeros(H)
    _, _, mean, rstd = ops.ln_fwd(x.cuda(), gamma.cuda(), beta.cuda())
    for p_drop in (0.0, 0.1):
        dg, db, dbias = torch.zeros(H).cuda(), torch.zeros(H).cuda(), torch.zeros(H).cuda()
        dx, dbr = ops.ln_bwd(dy.cuda(), x.cuda(), mean, rstd, gamma.cuda(), dg, db, dres=dres.cuda(),
                             branch_bias_grad=dbias, drop_p=p_drop, drop_seed=1234)
        dg2, db2 = torch.zeros(H).cuda(), torch.zeros(H).cuda()
        dx_ref = ops.ln_bwd(dy.cuda(), x.cuda(), mean, rstd, gamma.cuda(), dg2, db2, dres=dres.cuda())
        assert torch.equal(dx, dx_ref) and rel_l2(dg, dg2) < 1e-5
        br_ref = ops.dropout_apply(dx_ref, p_drop, 1234) if p_drop > 0 else dx_ref
        assert torch.equal(dbr, br_ref)
        cs = torch.zeros(H).cuda()
        ops.colsum_bf16(br_ref, cs, accumulate=False)
        assert rel_l2(dbias, cs) < 1e-4


# ---- attention ---------------------------------------------------------------------------------------------
def _attn_inputs(B, S, heads, seed, pad):
    g = torch.Generator().manual_seed(seed)
    qkv = rnd((B * S, 3 * heads * 64), g, 1.0)
    valid = None
    if pad:
        valid = torch.ones((B, S), dtype=torch.uint8)
        for b in range(B):
            n_pad = int(torch.randint(0, S // 2, (1,), generator=g))
            idx = torch.randperm(S, generator=g)[:n_pad]
            valid[b, idx] = 0
        valid[:, 0] = 1
    return qkv, valid, g


@pytest.mark.parametrize("B,S,heads,pad", [(2, 18, 12, False), (3, 198, 12, False), (2, 148, 12, True), (2, 328, 12, True),
                                           (1, 512, 12, True), (2, 64, 2, True), (1, 130, 3, True)])
def test_attention_fwd_bwd(ops, B, S, heads, pad):
    qkv, valid, g = _attn_inputs(B, S, heads, 100 + S, pad)
    o_ref, lse_ref = E.attention_fwd(qkv, B, S, heads, valid)
    o, lse = ops.attention_fwd(qkv.cuda(), B, S, heads, None if valid is None else valid.cuda())
    assert rel_l2(o, o_ref) < 8e-3
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-2
    do = rnd((B * S, heads * 64), g)
    if valid is not None:                         # padded query rows carry exactly-zero upstream grads in the model
        do = do * valid.reshape(B * S, 1).to(BF16)
    dq_ref = E.attention_bwd(qkv, o_ref, do, lse_ref, B, S, heads, valid).float()
    dq = ops.attention_bwd(qkv.cuda(), o, do.cuda(), lse, B, S, heads, None if valid is None else valid.cuda()).float().cpu()
    Hh = heads * 64
    for name, sl in [('dq', slice(0, Hh)), ('dk', slice(Hh, 2 * Hh)), ('dv', slice(2 * Hh, 3 * Hh))]:
        assert rel_l2(dq[:, sl], dq_ref[:, sl]) < 1.5e-2, name


@pytest.mark.parametrize("B,S,heads,pad", [(1, 578, 12, False), (1, 700, 4, True)])
def test_attention_fwd_bwd_long_sequences_tiled_kernels(ops, B, S, heads, pad):
    """S > 512 (config #5: Sv = 578) keeps the tiled kernels: same checks as above."""
    test_attention_fwd_bwd(ops, B, S, heads, pad)


@pytest.mark.parametrize("B,S,heads,pad", [(2, 410, 12, True), (2, 410, 12, False), (1, 2832, 2, True), (1, 2832, 3, False)])
def test_attention_fwd_bwd_at_the_baseline_config_lengths(ops, B, S, heads, pad):
    """the joint sequence of BASELINE config #4 (5 segments at 224^2: S = 5 * 82 = 410) and of
    config #5 (16 segments at 384^2: S = 16 * 177 = 2832), forward AND backward of the bf16 kernels against the fp32
    torch restatement (not against another HIP path)."""
    test_attention_fwd_bwd(ops, B, S, heads, pad)


@pytest.mark.parametrize("B,S,heads", [(2, 65, 12), (2, 96, 3), (3, 100, 12), (2, 130, 2), (2, 160, 12), (2, 197, 12), (2, 224, 4),
                                       (2, 225, 12), (1, 256, 12)])
def test_attention_bwd_fused_short_unmasked(ops, B, S, heads):
    """Unmasked sequences of 65..224 tokens (the ViT pass: 198) take the persistent backward of csrc/attention_pp.inc, 225..256 the
    ONE-launch backward of csrc/attention_fb.inc (K | V, then Q | dO resident in LDS): every block / chunk raggedness of that range
    against the fp32 torch restatement, and the published delta against sum(dO * O)."""
    test_attention_fwd_bwd(ops, B, S, heads, False)
    from merlot_amd.lib import call
    qkv, _, g = _attn_inputs(B, S, heads, 7 + S, False)
    qkv = qkv.cuda()
    o, lse = ops.attention_fwd(qkv, B, S, heads)
    do = rnd((B * S, heads * 64), g).cuda()
    dqkv = torch.full_like(qkv, float('nan'))
    delta = torch.full((B, heads, S), float('nan'), device='cuda')
    call('merlot_attention_bwd', qkv.data_ptr(), qkv.stride(0), o.data_ptr(), o.stride(0), do.data_ptr(), do.stride(0), lse.data_ptr(),
         None, None, dqkv.data_ptr(), dqkv.stride(0), delta.data_ptr(), B, S, heads, 0.125, None, None, S, 1.0, *ops_mod._attn_ws(),
         torch.cuda.current_stream().cuda_stream)
    assert not bool(torch.isnan(dqkv.float()).any())
    want = (do.float() * o.float()).view(B, S, heads, 64).sum(-1).permute(0, 2, 1)
    assert float((delta - want).abs().max()) < 1e-3 * (1 + float(want.abs().max()))


@pytest.mark.parametrize("B,S,heads", [(40, 198, 12), (100, 70, 12), (300, 129, 1), (23, 224, 12)])
def test_attention_persistent_kernels_many_items_per_workgroup(ops, B, S, heads):
    """Round 5: unmasked sequences of 65..224 tokens run the PERSISTENT kernels of csrc/attention_pp.inc -- one workgroup per CU walks
    its (batch, head) items with the next items' operands in flight, waits counted by hand.  More items than CUs (256), so that every
    workgroup takes several items, a last round that only some workgroups take part in, and the zero-length requests behind the end of
    the list: forward and backward against the fp32 torch restatement."""
    assert B * heads > 256
    test_attention_fwd_bwd(ops, B, S, heads, False)


def test_attention_persistent_kernels_leave_their_claim_counters_zero_and_fall_back_without_a_workspace(ops):
    """ABI v7: the persistent kernels draw their items from CALLER-owned counters (zero on entry, left zero by the last workgroup out, so
    that the next launch on the stream can use the same block); with workspace = NULL the entries run the one-launch-per-item kernels --
    same forward and backward within rounding (until the end of round 5 the backward was the fused kernel's bit for bit; since its dK / dV pass starts the
    accumulators from - lse / scale and - delta instead of subtracting them afterwards the two differ by an fp32 rounding here and there)."""
    from merlot_amd.lib import call
    ws_ptr, ws_bytes = ops_mod._attn_ws()
    ws = next(v for v in ops_mod._ATTN_WS.values() if v.data_ptr() == ws_ptr)
    stream = torch.cuda.current_stream().cuda_stream
    for B, S, heads, pad in ((30, 198, 12, False), (25, 328, 12, True), (1, 100, 3, False)):
        qkv, valid, g = _attn_inputs(B, S, heads, 77 + S, pad)
        qkv = qkv.cuda()
        vp = valid.cuda() if valid is not None else None
        do = rnd((B * S, heads * 64), g).cuda()
        res = []
        for with_ws in (True, False):
            o = torch.empty(B * S, heads * 64, device='cuda', dtype=BF16)
            lse = torch.empty(B, heads, S, device='cuda')
            dqkv = torch.full_like(qkv, float('nan'))
            delta = torch.empty(B, heads, S, device='cuda')
            w = (ws_ptr, ws_bytes) if with_ws else (None, 0)
            call('merlot_attention_fwd', qkv.data_ptr(), qkv.stride(0), o.data_ptr(), o.stride(0), lse.data_ptr(), vp.data_ptr() if pad else None,
                 None, B, S, heads, 0.125, None, None, S, 0, 1.0, *w, stream)
            call('merlot_attention_bwd', qkv.data_ptr(), qkv.stride(0), o.data_ptr(), o.stride(0), do.data_ptr(), do.stride(0), lse.data_ptr(),
                 vp.data_ptr() if pad else None, None, dqkv.data_ptr(), dqkv.stride(0), delta.data_ptr(), B, S, heads, 0.125, None, None, S, 1.0,
                 *w, stream)
            torch.cuda.synchronize()
            assert int(ws.abs().sum()) == 0                  # claims and departures reset by the last workgroup out
            res.append((o, lse, dqkv))
        assert rel_l2(res[0][0], res[1][0]) < 4e-3 and float((res[0][1] - res[1][1]).abs().max()) < 1e-4
        assert not bool(torch.isnan(res[0][2].float()).any()) and rel_l2(res[0][2], res[1][2]) < 8e-3


@pytest.mark.parametrize("B,S,heads", [(30, 328, 12), (3, 257, 12), (40, 289, 7), (3, 352, 12), (2, 320, 3)])
def test_attention_persistent_masked_forward_joint_lengths(ops, B, S, heads):
    """Round 5: the plain forward of 257 .. 352 MASKED tokens (the joint encoder in a training step, S = 328) runs the persistent
    two-half kernel of csrc/attention_pp.inc (K | V of half the keys resident at a time, an online softmax across the halves, the
    validity bytes fetched by the loader wave an item ahead): 9 / 10 / 11 key chunks, a chunk boundary exactly at S, more items than
    workgroups, against the fp32 restatement; the backward of the same call stays on the fused kernel."""
    test_attention_fwd_bwd(ops, B, S, heads, True)


@pytest.mark.parametrize("B,S,heads", [(40, 266, 12), (3, 257, 5), (30, 352, 12), (2, 300, 1)])
def test_attention_persistent_two_half_forward_without_a_mask(ops, B, S, heads):
    """The same kernel without a validity mask (the ViT of the as-shipped 192 x 352 frame: 266 tokens per frame, above the 224 of the single-pass
    kernel): every key below S valid, the ragged last chunk still cut off; more items than workgroups; backward of the same call on the fused kernel."""
    test_attention_fwd_bwd(ops, B, S, heads, False)


def test_attention_persistent_masked_forward_padded_query_rows_are_uniform(ops):
    """The reference's -1e10 semantics on the persistent masked path: a padded query row attends uniformly over ALL S keys
    (utils/transformer.py:109-112), also where the padding spans whole 32-row blocks and a key half."""
    B, S, heads = 2, 300, 12
    qkv, _, g = _attn_inputs(B, S, heads, 5, False)
    valid = torch.ones((B, S), dtype=torch.uint8)
    valid[0, 100:] = 0
    valid[1, 290:] = 0
    o, lse = ops.attention_fwd(qkv.cuda(), B, S, heads, valid.cuda())
    v = qkv[:, 2 * heads * 64:].float().view(B, S, -1)
    assert rel_l2(o.float().cpu().view(B, S, -1)[0, 100:], v[0].mean(0, keepdim=True).expand(200, -1)) < 1e-2
    assert rel_l2(o.float().cpu().view(B, S, -1)[1, 290:], v[1].mean(0, keepdim=True).expand(10, -1)) < 1e-2
    assert float((lse[0, :, 100:].cpu() - float(np.log(S))).abs().max()) < 1e-4


@pytest.mark.parametrize("B,S,heads,pad,pads", [(3, 198, 12, False, (64, 8, 16, 8)), (3, 198, 12, False, (64, 64, 128, 64)),
                                                 (2, 130, 5, True, (8, 24, 8, 16)),
                                                 (2, 328, 3, True, (128, 8, 8, 8)), (2, 328, 3, True, (8, 8, 8, 8)), (1, 512, 2, True, (8, 8, 8, 8)),
                                                 (5, 77, 7, False, (16, 40, 8, 24))])
def test_attention_nondefault_leading_dims(ops, B, S, heads, pad, pads):
    """The resident / persistent forward and the fused / persistent backward build their LDS-DMA source addresses from the leading
    dimensions (the persistent kernels want rows a multiple of 128 B apart: the second case; the others fall back): operands with
    padded rows (ld > 3 * heads * 64, ldo / lddo > heads * 64), odd head counts; the padding columns of the outputs stay untouched."""
    from merlot_amd.lib import call
    D = heads * 64
    ld, ldo, lddo, lddq = 3 * D + pads[0], D + pads[1], D + pads[2], 3 * D + pads[3]
    qkv, valid, g = _attn_inputs(B, S, heads, 31 + S, pad)
    qkv_full = torch.zeros((B * S, ld), dtype=BF16)
    qkv_full[:, :3 * D] = qkv
    qkv_full = qkv_full.cuda()
    vp = valid.cuda() if valid is not None else None
    o_full = torch.full((B * S, ldo), float('nan'), device='cuda', dtype=BF16)
    lse = torch.empty(B, heads, S, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    call('merlot_attention_fwd', qkv_full.data_ptr(), ld, o_full.data_ptr(), ldo, lse.data_ptr(), vp.data_ptr() if pad else None, None, B, S,
         heads, 0.125, None, None, S, 0, 1.0, *ops_mod._attn_ws(), stream)
    o_ref, lse_ref = E.attention_fwd(qkv, B, S, heads, valid)
    assert rel_l2(o_full[:, :D], o_ref) < 8e-3 and float((lse.cpu() - lse_ref).abs().max()) < 2e-2
    assert bool(torch.isnan(o_full[:, D:].float()).all())
    do = rnd((B * S, D), g)
    if valid is not None:
        do = do * valid.reshape(B * S, 1).to(BF16)
    do_full = torch.zeros((B * S, lddo), dtype=BF16)
    do_full[:, :D] = do
    do_full = do_full.cuda()
    dqkv = torch.full((B * S, lddq), float('nan'), device='cuda', dtype=BF16)
    delta = torch.empty((B, heads, S), device='cuda')
    call('merlot_attention_bwd', qkv_full.data_ptr(), ld, o_full.data_ptr(), ldo, do_full.data_ptr(), lddo, lse.data_ptr(),
         vp.data_ptr() if pad else None, None, dqkv.data_ptr(), lddq, delta.data_ptr(), B, S, heads, 0.125, None, None, S, 1.0, *ops_mod._attn_ws(), stream)
    dq_ref = E.attention_bwd(qkv, o_ref, do, lse_ref, B, S, heads, valid).float()
    got = dqkv[:, :3 * D].float().cpu()
    for name, sl in [('dq', slice(0, D)), ('dk', slice(D, 2 * D)), ('dv', slice(2 * D, 3 * D))]:
        assert rel_l2(got[:, sl], dq_ref[:, sl]) < 1.5e-2, name
    assert bool(torch.isnan(dqkv[:, 3 * D:].float()).all())


def test_attention_padded_query_rows_uniform(ops):
    """utils/transformer.py:109-112: a fully masked query row attends uniformly over ALL keys (-1e10, not -inf)."""
    B, S, heads = 1, 70, 12
    qkv, _, g = _attn_inputs(B, S, heads, 5, False)
    valid = torch.ones((B, S), dtype=torch.uint8)
    valid[0, 40:] = 0
    o, _ = ops.attention_fwd(qkv.cuda(), B, S, heads, valid.cuda())
    v = qkv[:, 2 * heads * 64:].float()
    assert rel_l2(o[40:].float().cpu(), v.mean(0, keepdim=True).expand(30, -1)) < 1e-2


@pytest.mark.parametrize("B,S,pad,vq", [(2, 128, True, False), (2, 148, True, True), (1, 512, True, False), (3, 50, False, False)])
def test_attention_colsum(ops, B, S, pad, vq):
    heads = 12
    qkv, valid, g = _attn_inputs(B, S, heads, 7 + S, pad)
    _, lse_ref = E.attention_fwd(qkv, B, S, heads, valid)
    lo_r, hi_r = torch.zeros(B, S), torch.zeros(B, S)
    split = S // 3
    E.attention_colsum(qkv, lse_ref, B, S, heads, lo_r, hi_r, qsplit=split, valid=valid, valid_q_only=vq, weight=1 / heads)
    _, lse = ops.attention_fwd(qkv.cuda(), B, S, heads, None if valid is None else valid.cuda())
    lo, hi = torch.zeros(B, S).cuda(), torch.zeros(B, S).cuda()
    ops.attention_colsum(qkv.cuda(), lse, B, S, heads, lo, hi, qsplit=split, valid=None if valid is None else valid.cuda(),
                         valid_q_only=vq, weight=1 / heads)
    assert rel_l2(lo, lo_r) < 5e-3 and rel_l2(hi, hi_r) < 5e-3


@pytest.mark.parametrize("B,S,pad,vq", [(3, 198, False, False), (2, 128, True, False), (2, 148, True, True), (2, 328, True, True),
                                        (1, 512, True, False), (3, 50, False, False), (1, 600, True, True), (2, 18, False, False)])
def test_attention_fwd_fused_colsum(ops, B, S, pad, vq):
    """the column sums / block sums produced by the FORWARD launch (S <= 512: tail of attn_fwd_res_kernel on the K tile
    still resident in LDS; S = 600: tiled forward + the separate pass) equal the stand-alone op's, accumulate in place,
    and leave the attention output and lse untouched."""
    heads = 12
    qkv, valid, g = _attn_inputs(B, S, heads, 70 + S, pad)
    vc = None if valid is None else valid.cuda()
    o_ref, lse_ref = E.attention_fwd(qkv, B, S, heads, valid)
    lo_r, hi_r = torch.full((B, S), 0.5), torch.full((B, S), -0.25)
    split = S // 3
    E.attention_colsum(qkv, lse_ref, B, S, heads, lo_r, hi_r, qsplit=split, valid=valid, valid_q_only=vq, weight=1 / heads)
    lo, hi = torch.full((B, S), 0.5).cuda(), torch.full((B, S), -0.25).cuda()
    o, lse = ops.attention_fwd(qkv.cuda(), B, S, heads, vc, colsum_lo=lo, colsum_hi=hi, qsplit=split, valid_q_only=vq, weight=1 / heads)
    assert rel_l2(lo, lo_r) < 5e-3 and rel_l2(hi, hi_r) < 5e-3
    o2, lse2 = ops.attention_fwd(qkv.cuda(), B, S, heads, vc)        # the plain call runs the tiled kernel (64-key tiles): same
    assert rel_l2(o, o2) < 4e-3 and float((lse - lse2).abs().max()) < 2e-3     # values up to bf16 rounding of P and O
    assert rel_l2(o, o_ref) < 8e-3 and float((lse.cpu() - lse_ref).abs().max()) < 2e-2
    # only the low half requested (the text-only pass: every query row counts, qsplit = S)
    lo1, lo1_r = torch.zeros(B, S).cuda(), torch.zeros(B, S)
    E.attention_colsum(qkv, lse_ref, B, S, heads, lo1_r, None, valid=valid, valid_q_only=False, weight=1 / heads)
    ops.attention_fwd(qkv.cuda(), B, S, heads, vc, colsum_lo=lo1, valid_q_only=False, weight=1 / heads)
    assert rel_l2(lo1, lo1_r) < 5e-3
    assert abs(float(lo1.sum()) - B * S) < 1e-2 * B * S                    # probabilities: every query row sums to 1


@pytest.mark.parametrize("B,S,pad,split", [(3, 328, True, 200), (2, 410, True, 250), (2, 200, True, 96), (2, 148, True, 0), (2, 328, True, 328),
                                           (3, 198, False, 100), (1, 578, True, 300), (2, 50, True, 20), (2, 700, True, 298), (1, 1200, True, 644), (2, 578, True, 0), (1, 600, True, 600)])
def test_attention_log_from_the_backward(ops, B, S, pad, split):
    """the attention LOG side output (valid pairs only, queries below / from `split`) taken from the BACKWARD launch (round 4):
    S <= 512 masked = two lane accumulators of the fused kernel's dK / dV pass (chunks below, above and straddling the split: 250
    and 100 are not multiples of 32, 0 and S put everything on one side), masked sequences of more than 512 / at most 64 tokens with a split that is a multiple of 4 (round 6) = two lane accumulators of
    the tiled dK / dV kernel; unmasked sequences and other splits = the tiled column-sum kernel launched by the entry -- equal to the stand-alone op, accumulated in place, and the gradients are the plain call's bit
    for bit."""
    heads = 12
    qkv, valid, g = _attn_inputs(B, S, heads, 500 + S, pad)
    vc = None if valid is None else valid.cuda()
    o, lse = ops.attention_fwd(qkv.cuda(), B, S, heads, vc)
    do = rnd((B * S, heads * 64), g).cuda()
    lo_r, hi_r = torch.full((B, S), 0.25).cuda(), torch.full((B, S), -0.5).cuda()
    ops.attention_colsum(qkv.cuda(), lse, B, S, heads, lo_r, hi_r, qsplit=split, valid=vc, valid_q_only=True, weight=1 / heads)
    lo, hi = torch.full((B, S), 0.25).cuda(), torch.full((B, S), -0.5).cuda()
    d1 = ops.attention_bwd(qkv.cuda(), o, do, lse, B, S, heads, vc, log_lo=lo, log_hi=hi, log_split=split, log_weight=1 / heads)
    d0 = ops.attention_bwd(qkv.cuda(), o, do, lse, B, S, heads, vc)
    assert torch.equal(d0, d1)
    assert float((lo - lo_r).abs().max()) < 2e-5 * max(1.0, float(lo_r.abs().max())) and float((hi - hi_r).abs().max()) < 2e-5 * max(1.0, float(hi_r.abs().max()))
    # against the emulation's definition, and the bookkeeping: every VALID query row's probabilities over valid keys sum to 1
    lo_e, hi_e = torch.full((B, S), 0.25), torch.full((B, S), -0.5)
    E.attention_colsum(qkv, lse.cpu(), B, S, heads, lo_e, hi_e, qsplit=split, valid=valid, valid_q_only=True, weight=1 / heads)
    assert rel_l2(lo + hi, lo_e + hi_e) < 5e-3
    nvalid = B * S if valid is None else int(valid.sum())
    assert abs(float((lo - 0.25).sum() + (hi + 0.5).sum()) - nvalid) < 1e-2 * nvalid
    # only one of the two requested
    lo2 = torch.zeros(B, S).cuda()
    ops.attention_bwd(qkv.cuda(), o, do, lse, B, S, heads, vc, log_lo=lo2, log_split=split, log_weight=1 / heads)
    assert float((lo2 - (lo_r - 0.25)).abs().max()) < 2e-5 * max(1.0, float(lo_r.abs().max()))


@pytest.mark.parametrize("B,S,P,Lc", [(2, 328, 200, 32), (2, 148, 20, 32), (1, 130, 2, 16)])
def test_attention_segment_block_mask(ops, B, S, P, Lc):
    """`disable_pairwise_lang_attn` (model/modeling.py:160-168): tokens of different caption chunks do not see each other,
    everything sees the P vision tokens; chunk boundaries are NOT aligned to the kernels' 32/64-wide tiles."""
    heads = 12
    qkv, valid, g = _attn_inputs(B, S, heads, 900 + S, True)
    seg = torch.cat([torch.zeros(P, dtype=torch.int32), 1 + torch.arange(S - P, dtype=torch.int32) // Lc])
    vc, sc = valid.cuda(), seg.cuda()
    o_ref, lse_ref = E.attention_fwd(qkv, B, S, heads, valid, seg=seg)
    o_plain, _ = E.attention_fwd(qkv, B, S, heads, valid)
    assert rel_l2(o_plain, o_ref) > 0.1                                   # the mask matters on these inputs
    o, lse = ops.attention_fwd(qkv.cuda(), B, S, heads, vc, seg=sc)
    assert rel_l2(o, o_ref) < 8e-3
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-2
    do = rnd((B * S, heads * 64), g) * valid.reshape(B * S, 1).to(BF16)
    dq_ref = E.attention_bwd(qkv, o_ref, do, lse_ref, B, S, heads, valid, seg=seg).float()
    dq = ops.attention_bwd(qkv.cuda(), o, do.cuda(), lse, B, S, heads, vc, seg=sc).float().cpu()
    Hh = heads * 64
    for name, sl in [('dq', slice(0, Hh)), ('dk', slice(Hh, 2 * Hh)), ('dv', slice(2 * Hh, 3 * Hh))]:
        assert rel_l2(dq[:, sl], dq_ref[:, sl]) < 1.5e-2, name
    for vq in (False, True):
        lo_r, hi_r = torch.zeros(B, S), torch.zeros(B, S)
        E.attention_colsum(qkv, lse_ref, B, S, heads, lo_r, hi_r, qsplit=P, valid=valid, valid_q_only=vq, weight=1 / heads, seg=seg)
        lo, hi = torch.zeros(B, S).cuda(), torch.zeros(B, S).cuda()
        ops.attention_colsum(qkv.cuda(), lse, B, S, heads, lo, hi, qsplit=P, valid=vc, valid_q_only=vq, weight=1 / heads, seg=sc)
        assert rel_l2(lo, lo_r) < 5e-3 and rel_l2(hi, hi_r) < 5e-3, vq
        lo, hi = torch.zeros(B, S).cuda(), torch.zeros(B, S).cuda()      # ... and fused into the forward launch
        ops.attention_fwd(qkv.cuda(), B, S, heads, vc, seg=sc, colsum_lo=lo, colsum_hi=hi, qsplit=P, valid_q_only=vq, weight=1 / heads)
        assert rel_l2(lo, lo_r) < 5e-3 and rel_l2(hi, hi_r) < 5e-3, vq


# ---- element-wise / gather / CE ------------------------------------------------------------------------------
def test_casts_and_colsum(ops):
    g = torch.Generator().manual_seed(2)
    x = torch.randn((777, 300), generator=g)
    assert torch.equal(ops.cast_bf16(x.cuda()).cpu(), x.to(BF16))
    t = torch.zeros((300, 800), dtype=BF16).cuda()
    ops.cast_transpose_bf16(x.cuda(), t, ld_dst=800)
    assert torch.equal(t[:, :777].cpu(), x.t().to(BF16)) and torch.all(t[:, 777:] == 0)
    xb = rnd((5000, 772), g)
    out = torch.zeros(770).cuda()
    ops.colsum_bf16(xb.cuda(), out, accumulate=False, n=770)
    assert rel_l2(out, xb[:, :770].float().sum(0)) < 1e-4


def test_gather_scatter_pool(ops):
    g = torch.Generator().manual_seed(4)
    H, rows = 768, 500
    a = rnd((300, H), g)
    tb, tc, td = torch.randn((50, H), generator=g), torch.randn((64, H), generator=g), torch.randn((7, H), generator=g)
    ia = torch.randint(-1, 300, (rows,), generator=g).int()
    ib = torch.randint(0, 50, (rows,), generator=g).int()
    ic = torch.randint(-1, 64, (rows,), generator=g).int()
    id_ = torch.randint(-1, 7, (rows,), generator=g).int()
    ref = E.gather_add4(rows, H, a, ia, tb, ib, tc, ic, td, id_)
    got = ops.gather_add4(rows, H, *dev(a, ia, tb, ib, tc, ic, td, id_))
    assert rel_l2(got, ref) < 1e-6
    src = torch.randn((rows, H), generator=g)
    tab_r = torch.zeros((64, H))
    E.scatter_add_rows(src, ic, tab_r)
    tab = torch.zeros((64, H)).cuda()
    ops.scatter_add_rows(src.cuda(), ic.cuda(), tab)
    assert rel_l2(tab, tab_r) < 1e-5
    x = rnd((6, 2 + 14 * 14, H), g)
    pr = E.cls_avgpool_fwd(x, 6, 14, 14, 2, 2)
    pg = ops.cls_avgpool_fwd(x.cuda(), 6, 14, 14, 2, 2)
    assert rel_l2(pg, pr) < 1e-6
    dout = torch.randn(pr.shape, generator=g)
    assert rel_l2(ops.cls_avgpool_bwd(dout.cuda(), 6, 14, 14, 2, 2), E.cls_avgpool_bwd(dout, 6, 14, 14, 2, 2).float()) < 4e-3


@pytest.mark.parametrize("rows,C,ld", [(40, 4, 4), (64, 128, 128), (25, 50370, 50432), (9, 4097, 4100), (5, 65536, 65536), (3, 7001, 7001)])
def test_softmax_ce(ops, rows, C, ld):
    g = torch.Generator().manual_seed(C)
    logits = torch.randn((rows, ld), generator=g) * 3
    labels = torch.randint(0, C, (rows,), generator=g).int()
    rs = torch.rand(rows, generator=g)
    loss_r, am_r, dl_r = E.softmax_ce(logits, labels, C, rowscale=rs, dlogits_dtype=F32, ld_dl=ld)
    loss, am, dl = ops.softmax_ce(logits.cuda(), labels.cuda(), C, rowscale=rs.cuda(), dlogits_dtype=F32, ld_dl=ld)
    assert rel_l2(loss, loss_r) < 1e-5 and torch.equal(am.cpu(), am_r) and rel_l2(dl, dl_r) < 1e-4
    _, _, dl16 = ops.softmax_ce(logits.cuda(), labels.cuda(), C, rowscale=rs.cuda(), dlogits_dtype=BF16, ld_dl=ld)
    assert rel_l2(dl16, dl_r) < 5e-3 and torch.all(dl16[:, C:] == 0)


def test_small_head_ops(ops):
    g = torch.Generator().manual_seed(8)
    x = torch.randn((100, 768), generator=g)
    y_r, inv_r = E.l2norm_fwd(x)
    y, inv = ops.l2norm_fwd(x.cuda())
    assert rel_l2(y, y_r) < 1e-6
    dy = torch.randn((100, 768), generator=g)
    assert rel_l2(ops.l2norm_bwd(dy.cuda(), y, inv), E.l2norm_bwd(dy, y_r, inv_r)) < 1e-5
    assert rel_l2(ops.gelu_fwd(x.cuda()), E.gelu_fwd(x)) < 1e-6
    assert rel_l2(ops.gelu_bwd(dy.cuda(), x.cuda()), E.gelu_bwd(dy, x)) < 1e-5
