"""The 8-bit weight-gradient path of BASELINE config #5 (judge row g1): merlot_quantize_f8 + merlot_gemm_f8_tn through the C-ABI, and
`model.fp8_backward` through the model.

The reference has no 8-bit path (precision policy: utils/model_utils.py:572-602; the contraction is tf.gradients of the dense layers,
utils/transformer.py:141-163), so the checker is arithmetic: the quantiser BIT-EXACT against torch's own float8 conversions of the same scaled
values, the GEMM equal to a matmul of the DEQUANTISED operands (every product of two 8-bit floats is exact in fp32; only the summation order
differs) to 3e-5 of the output's maximum -- all of the error of the path is the quantisation the contract allows."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {0: torch.float8_e4m3fn, 1: torch.float8_e5m2}
FMAX = {0: 448.0, 1: 57344.0}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ops():
    from merlot_amd import ops
    return ops


def test_tr_b8_lane_map_is_the_one_the_kernel_assumes():
    """ds_read_b64_tr_b8: within a 16-lane group, lane t receives as byte j the byte (t & 7) of the 8 bytes that lane 2 j + (t >> 3) addressed --
    column t of an [8 rows][16 columns] byte block whose row r is supplied by lanes 2 r (columns 0-7) and 2 r + 1 (columns 8-15)."""
    lib = ctypes.CDLL(os.path.join(ROOT, 'merlot_amd', 'libmerlot_probe.so'))
    lib.merlot_probe_tr8.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    src = torch.stack([torch.arange(64, dtype=torch.uint8).repeat_interleave(8), torch.arange(8, dtype=torch.uint8).repeat(64)]).cuda()
    out = torch.zeros_like(src)
    for i in range(2):
        assert lib.merlot_probe_tr8(src[i].data_ptr(), out[i].data_ptr(), None) == 0
    torch.cuda.synchronize()
    ol, ob = out[0].cpu().view(64, 8), out[1].cpu().view(64, 8)
    for t in range(64):
        for j in range(8):
            assert int(ol[t, j]) == (t & ~15) + 2 * j + ((t & 15) >> 3) and int(ob[t, j]) == (t & 7), (t, j)


@pytest.mark.parametrize('fmt', [0, 1])
@pytest.mark.parametrize('rows,cols,ld', [(1000, 768, 768), (37, 3072, 3072), (256, 64, 128), (130, 8, 8)])
def test_quantize_f8_current_is_bit_exact_against_torch(rows, cols, ld, fmt):
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(rows * 7 + cols)
    x = (torch.randn(rows, ld, generator=g) * torch.logspace(-3, 1, ld)[None, :]).to(torch.bfloat16).cuda()[:, :cols]
    y, scale = ops.quantize_f8(x, fmt)
    rp = (rows + 127) // 128 * 128
    assert y.shape == (rp, cols) and y.dtype == DT[fmt]
    amax = x.float().abs().max()
    s = torch.tensor(FMAX[fmt], device=x.device) / amax
    assert scale[2].item() == amax.item() and scale[3].item() == amax.item() and scale[0].item() == s.item()
    assert abs(scale[1].item() * s.item() - 1.0) < 1e-6
    want = (x.float() * s).clamp(-FMAX[fmt], FMAX[fmt]).to(DT[fmt])
    assert torch.equal(y[:rows].view(torch.uint8), want.view(torch.uint8))
    assert int(y[rows:].view(torch.uint8).max() if rp > rows else 0) == 0          # the padding rows are zeros


def test_quantize_f8_delayed_uses_the_previous_amax_and_records_the_new_one():
    ops = _ops()
    x1 = (torch.randn(512, 256) * 2.0).to(torch.bfloat16).cuda()
    x2 = (torch.randn(512, 256) * 5.0).to(torch.bfloat16).cuda()      # larger: part of it saturates under x1's scale
    y1, blk = ops.quantize_f8(x1, 1)
    a1 = float(x1.float().abs().max())
    y2, blk2 = ops.quantize_f8(x2, 1, scale=blk)
    assert blk2 is blk
    s1 = 57344.0 / a1
    assert abs(blk[0].item() / s1 - 1.0) < 1e-6 and blk[2].item() == a1 and blk[3].item() == float(x2.float().abs().max())
    want = (x2.float() * blk[0]).clamp(-57344.0, 57344.0).to(torch.float8_e5m2)
    assert torch.equal(y2[:512].view(torch.uint8), want.view(torch.uint8))
    assert float(y2.float().abs().max()) == 57344.0                   # saturated, not overflowed to inf
    y3, _ = ops.quantize_f8(x1, 1, scale=blk)                         # the third call scales by x2's amax
    assert blk[2].item() == float(x2.float().abs().max()) and blk[3].item() == a1


def _deq(y, blk):
    return y.float() * blk[1]


@pytest.mark.parametrize('R,M,N,fa,fb,acc', [(2048, 256, 256, 0, 0, False), (4096, 768, 768, 1, 0, True), (8192, 2304, 768, 1, 0, False),
                                             (6144, 200, 328, 0, 1, False), (2651, 3072, 768, 1, 1, True), (24576, 768, 3072, 1, 0, False),
                                             (9248, 768, 3072, 1, 0, True)])
def test_gemm_f8_tn_equals_matmul_of_the_dequantised_operands(R, M, N, fa, fb, acc):
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(R + M)
    a = (torch.randn(R, M, device='cuda', generator=g) * torch.rand(1, M, device='cuda', generator=g) * 3).bfloat16()
    b = (torch.randn(R, N, device='cuda', generator=g) * 0.7).bfloat16()
    pad16 = lambda v: (v + 15) // 16 * 16
    Rp = (R + 127) // 128 * 128
    a8 = torch.empty(Rp, pad16(M), device='cuda', dtype=DT[fa])[:, :M]
    b8 = torch.empty(Rp, pad16(N), device='cuda', dtype=DT[fb])[:, :N]
    a8, sa = ops.quantize_f8(a, fa, out=a8)
    b8, sb = ops.quantize_f8(b, fb, out=b8)
    out = torch.full((M, N), 0.5 if acc else float('nan'), device='cuda')
    ops.gemm_f8_tn(a8, sa, b8, sb, out, accumulate=acc, alpha=0.75)
    ref = 0.75 * (_deq(a8, sa)[:R].double().T @ _deq(b8, sb)[:R].double()) + (0.5 if acc else 0.0)
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 3e-5
    full = 0.75 * (a.double().T @ b.double()) + (0.5 if acc else 0.0)
    assert float((out.double() - full).norm() / full.norm()) < 9e-2          # the quantisation itself: 3 / 2 mantissa bits per operand


def test_gemm_f8_tn_rejects_what_the_kernel_cannot_take():
    ops = _ops()
    from merlot_amd.lib import MerlotHipError
    x = torch.randn(4096, 256, device='cuda').bfloat16()
    x8, sx = ops.quantize_f8(x, 0)
    out = torch.zeros(256, 256, device='cuda')
    with pytest.raises(MerlotHipError):
        ops.gemm_f8_tn(x8[:4000], sx, x8[:4000], sx, out)            # R % 128 != 0
    with pytest.raises(MerlotHipError):
        ops.gemm_f8_tn(x8[:1024], sx, x8[:1024], sx, out)            # R < 2048
    with pytest.raises(MerlotHipError):
        ops.gemm_f8_tn(x8[:, :100], sx, x8[:, :100], sx, out[:100, :100], m=100, n=100)   # M, N < 128
    assert ops.LIB.query('merlot_gemm_f8_tn_workspace_bytes', 768, 3072, 4000) == 0


@pytest.mark.parametrize('modes', ['w1,w2', 'w1,w2,wqkv,wproj', 'w1,w2,wqkv,wproj,e4m3'])
def test_config5_geometry_fp8_backward_gradients_close_to_the_bf16_backward(modes):
    """BASELINE config #5 geometry with `fp8_backward`: the weight gradients of the three stacks through merlot_gemm_f8_tn.  The forward is untouched
    (identical losses); every parameter gradient stays within the quantisation's noise of the bf16 backward's: rel-L2 per tensor <= 0.35, cosine >= 0.94 (the per-product rounding of 2-3 mantissa bits does not
    average out of a sum, and where the true sum is small against its terms -- feature columns that are nearly constant over the tokens -- the
    relative error of that column grows accordingly: measured worst 0.22 / 0.977 on the ViT's first fc1; medians 5e-2), untouched tensors (biases, LayerNorm, embeddings: their own gradients use bf16 operands) as before.  Two steps: the second runs on
    DELAYED scales (the first one's amax)."""
    from common import tiny_config, synth_batch, rel_l2
    from merlot_amd import MerlotModel, ParamStore
    from oracle import merlot_oracle as mo
    out = {}
    b = w = None
    for bwd in (False, modes):
        cfg = tiny_config(image_size=[384, 384], num_chunks_in_group=16, max_position_embeddings=1024, fp8_forward='ln', fp8_backward=bwd,
                          masking_use_attn=False)
        if b is None:
            b = synth_batch(cfg, E=1, num_chunks=16, seed=3)
            w = mo.init_weights(cfg, 0)
        st = ParamStore(cfg, 'cuda', seed=0)
        st.load_tf_weights({k: v.detach() for k, v in w.items()})
        steps = []
        for step in range(2):
            st.zero_grad()
            pm = MerlotModel(cfg, True, False, b['image'].cuda(), b['input_ids'].cuda(), mask_input=True,
                             shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']).cuda(), params=st,
                             noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
            loss = pm.mask_loss()[0] + pm.contrastive_loss()[0] + pm.temporal_loss(
                torch.from_numpy(b['shuffled_idx_img']).cuda(), torch.from_numpy(b['video_src_ids']).cuda())[0]
            loss.backward()
            torch.cuda.synchronize()
            steps.append((float(loss), {k: v.float().cpu() for k, v in st.export_tf_grads().items()}))
        out[bwd] = steps
    for step in range(2):
        l0, g0 = out[False][step]
        l1, g1 = out[modes][step]
        assert l0 == l1, (step, l0, l1)                                  # the forward does not change
        worst, cos_min = 0.0, 1.0
        for k, g in g0.items():
            if float(g.norm()) == 0:
                continue
            r = rel_l2(g1[k], g)
            c = float((g1[k].double().flatten() @ g.double().flatten()) / (g1[k].double().norm() * g.double().norm()))
            worst, cos_min = max(worst, r), min(cos_min, c)
            assert torch.isfinite(g1[k]).all()
            assert r < 0.35 and c > 0.94, (step, k, r, c)
        print(f'fp8_backward {modes} step {step}: worst rel-L2 {worst:.3e}, min cosine {cos_min:.5f}')


# ---------------------------------------------------------------------------------------------------------------------------------------------
# the 8-bit copies from the PRODUCING launches ('fuse'): each must be bit-equal to the stand-alone pass over the bf16 tensor with the same scale


def _block(scale, fmt):
    """a scale block as the delayed path finds it: {s, 1/s, amax, 0}"""
    return torch.tensor([scale, 1.0 / scale, FMAX[fmt] / scale, 0.0], device='cuda', dtype=torch.float32)


def _q(x16, s, fmt):
    return (x16.float() * s).clamp(-FMAX[fmt], FMAX[fmt]).to(DT[fmt])


@pytest.mark.parametrize('rows,H', [(4096, 768), (777, 768), (512, 1024)])
def test_ln_fwd_q8t_copy_equals_the_pass_over_its_bf16_output(rows, H):
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(rows)
    x = (torch.randn(rows, H, device='cuda', generator=g) * 2 + 0.3).bfloat16()
    gamma = (1 + 0.1 * torch.randn(H, device='cuda', generator=g)).float()
    beta = (0.1 * torch.randn(H, device='cuda', generator=g)).float()
    y16, _, mean, rstd = ops.ln_fwd(x, gamma, beta)
    blk = _block(37.0, 0)
    y16b, y8, mean_b, rstd_b = ops.ln_fwd_q8t(x, gamma, beta, blk, out_bf16=True)
    assert torch.equal(y16b, y16) and torch.equal(mean_b, mean) and torch.equal(rstd_b, rstd)
    assert y8.shape[0] == (rows + 127) // 128 * 128 and int(y8[rows:].view(torch.uint8).max() if y8.shape[0] > rows else 0) == 0   # whole K-tiles of 128 rows, the padding zero
    assert torch.equal(y8[:rows].view(torch.uint8), _q(y16, 37.0, 0).view(torch.uint8))
    assert blk[3].item() == float(y16.float().abs().max()) and blk[0].item() == 37.0
    _, y8b, _, _ = ops.ln_fwd_q8t(x, gamma, beta, blk)                # without the bf16 output
    assert torch.equal(y8b.view(torch.uint8), y8.view(torch.uint8))


@pytest.mark.parametrize('fmt', [0, 1])
@pytest.mark.parametrize('keep', [True, False])
@pytest.mark.parametrize('M', [256 * 260, 256 * 130 + 77])
def test_dgelu_epilogue_copy_is_the_8bit_rounding_of_its_fp32_results(fmt, keep, M):
    """the copy is taken from the epilogue's fp32 results (one rounding, not bf16 then 8 bits): bit-equal to the conversion of the same launch's f32 output"""
    ops = _ops()
    N, K = 3072, 768                                                   # more tiles than workgroups: several tiles per wave feed one amax; a ragged last row block
    g = torch.Generator(device='cuda').manual_seed(5)
    a = (torch.randn(M, K, device='cuda', generator=g) * 0.05).bfloat16()
    bt = (torch.randn(N, K, device='cuda', generator=g) * 0.05).bfloat16()
    u = torch.randn(M, N, device='cuda', generator=g).bfloat16()
    cs0 = torch.zeros(N, device='cuda')
    ref = ops.gemm_nt(a, bt, epilogue=ops.EPI_DGELU, aux_in=u, colsum_out=cs0)
    ref32 = ops.gemm_nt(a, bt, epilogue=ops.EPI_DGELU, aux_in=u, out_dtype=torch.float32)
    s = FMAX[fmt] / float(ref.float().abs().max()) * 0.7               # a delayed scale that is not this tensor's own
    blk = _block(s, fmt)
    cs1 = torch.zeros(N, device='cuda')
    c, c8 = ops.gemm_nt_q8(a, bt, blk, fmt, epilogue=ops.EPI_DGELU, aux_in=u, colsum_out=cs1, keep_bf16=keep)
    if keep:
        assert torch.equal(c, ref)
    else:
        assert c is None
    assert torch.equal(c8[:M].view(torch.uint8), _q(ref32, s, fmt).view(torch.uint8))
    assert int(c8[M:].view(torch.uint8).max() if c8.shape[0] > M else 0) == 0
    assert blk[3].item() == float(ref32.abs().max())
    assert torch.allclose(cs0, cs1, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('keep', [True, False])
@pytest.mark.parametrize('per_row', [True, False])
@pytest.mark.parametrize('M', [256 * 130, 256 * 40 + 200])
def test_gelu_epilogue_copy_is_the_8bit_rounding_of_its_fp32_results(keep, per_row, M):
    ops = _ops()
    N, K = 3072, 768
    g = torch.Generator(device='cuda').manual_seed(6)
    x = torch.randn(M, K, device='cuda', generator=g).bfloat16()
    w = (torch.randn(N, K, device='cuda', generator=g) * 0.03).bfloat16()
    bias = torch.randn(N, device='cuda', generator=g) * 0.1
    gamma, beta = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
    w8, sw = ops.quantize_e4m3(w)
    if per_row:
        _, x8, rs, _, _ = ops.ln_fwd_q8(x, gamma, beta)
        sx = None
    else:
        x8, sx = ops.quantize_e4m3(x)
        rs = None
    u0 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    ref = ops.gemm_fp8_nt(x8, sx, w8, sw, bias=bias, epilogue=ops.EPI_GELU, aux_out=u0, a_row_scale=rs)
    ref32 = ops.gemm_fp8_nt(x8, sx, w8, sw, bias=bias, epilogue=ops.EPI_GELU, aux_out=torch.empty_like(u0), a_row_scale=rs, out_dtype=torch.float32)
    s = 448.0 / float(ref.float().abs().max()) * 1.3                   # part of the range saturates
    blk = _block(s, 0)
    u1 = torch.empty_like(u0)
    c, c8 = ops.gemm_fp8_nt_q8(x8, sx, w8, sw, blk, bias=bias, aux_out=u1, a_row_scale=rs, keep_bf16=keep)
    assert torch.equal(u1, u0)
    if keep:
        assert torch.equal(c, ref)
    assert torch.equal(c8[:M].view(torch.uint8), _q(ref32, s, 0).view(torch.uint8))
    assert blk[3].item() == float(ref32.abs().max())


def test_what_the_8bit_conversions_do_beyond_their_range():
    """v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 without a clamp in front (recorded, not relied upon: every producer clamps)"""
    lib = ctypes.CDLL(os.path.join(ROOT, 'merlot_amd', 'libmerlot_probe.so'))
    lib.merlot_probe_cvt8.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    x = torch.tensor([448.0, 449.0, 480.0, 1e3, 1e6, float('inf'), -1e6, 57344.0, 6e4, 7e4, 1e9, 0.0], device='cuda')
    out = torch.zeros(2 * x.numel(), dtype=torch.uint8, device='cuda')
    assert lib.merlot_probe_cvt8(x.data_ptr(), out.data_ptr(), x.numel(), None) == 0
    torch.cuda.synchronize()
    o = out.cpu().view(2, -1)
    print('e4m3:', [f'{v:g}->0x{int(b):02x}' for v, b in zip(x.tolist(), o[0])])
    print('e5m2:', [f'{v:g}->0x{int(b):02x}' for v, b in zip(x.tolist(), o[1])])
    assert int(o[0][0]) == 0x7e and int(o[1][7]) == 0x7b                 # the formats' maxima convert exactly


@pytest.mark.parametrize('p', [0.0, 0.1])
@pytest.mark.parametrize('fmt', [0, 1])
def test_ln_bwd_copy_equals_the_pass_over_the_branch_gradient(p, fmt):
    ops = _ops()
    rows, H = 5000, 768
    g = torch.Generator(device='cuda').manual_seed(8)
    x = torch.randn(rows, H, device='cuda', generator=g).bfloat16()
    dy = (torch.randn(rows, H, device='cuda', generator=g) * 1e-3).bfloat16()
    dres = (torch.randn(rows, H, device='cuda', generator=g) * 1e-3).bfloat16()
    gamma, beta = torch.ones(H, device='cuda'), torch.zeros(H, device='cuda')
    _, _, mean, rstd = ops.ln_fwd(x, gamma, beta)
    outs = []
    for blk in (None, _block(57344.0 / 4e-3 if fmt else 448.0 / 4e-3, fmt)):
        dg, dbt, bb = torch.zeros(H, device='cuda'), torch.zeros(H, device='cuda'), torch.zeros(H, device='cuda')
        r = ops.ln_bwd(dy, x, mean, rstd, gamma, dg, dbt, dres=dres, branch_bias_grad=bb, drop_p=p, drop_seed=77, db8_block=blk, db8_fmt=fmt)
        outs.append((r, dg, dbt, bb, blk))
    (dx0, br0), (dx1, br1, br8) = outs[0][0], outs[1][0]
    assert torch.equal(dx0, dx1) and torch.equal(br0, br1)
    for k in (1, 2, 3):
        assert torch.allclose(outs[0][k], outs[1][k], rtol=1e-4, atol=1e-6)
    blk = outs[1][4]
    assert torch.equal(br8[:rows].view(torch.uint8), _q(br0, blk[0].item(), fmt).view(torch.uint8)) and br8.shape[0] % 128 == 0
    assert blk[3].item() == float(br0.float().abs().max())


@pytest.mark.parametrize('fmt', [0, 1])
@pytest.mark.parametrize('M,N,K', [(4096, 768, 3072), (1000, 768, 3072), (66560, 768, 2304)])
def test_gemm_f8_nt_with_an_e5m2_operand_equals_matmul_of_the_dequantised_operands(M, N, K, fmt):
    """merlot_gemm_f8_nt: the input-gradient GEMM on the copy the GELU' epilogue wrote (A in e5m2 or e4m3, the weights in e4m3)"""
    ops = _ops()
    g = torch.Generator(device='cuda').manual_seed(M + fmt)
    a = (torch.randn(M, K, device='cuda', generator=g) * 1e-3).bfloat16()
    w = (torch.randn(N, K, device='cuda', generator=g) * 0.03).bfloat16()
    a8, sa = ops.quantize_f8(a, fmt, row_multiple=1)
    w8, sw = ops.quantize_f8(w, 0, row_multiple=1)
    out = ops.gemm_f8_nt(a8, sa, w8, sw)
    ref = _deq(a8, sa).double() @ _deq(w8, sw).double().T
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 6e-3      # bf16 output rounding
    assert float((out.double() - ref).norm() / ref.norm()) < 3e-3


@pytest.mark.parametrize('B,S,masked,fmt', [(4, 578, False, 1), (2, 700, True, 1), (3, 578, False, 0), (5, 40, True, 1)])
def test_attention_bwd_copy_equals_the_pass_over_dqkv(B, S, masked, fmt):
    """merlot_attention_bwd_q8 (the tiled dQ / dK dV pair: config #5's sequence lengths): dqkv identical to the plain entry's, the copy bit-equal to the
    conversion of the bf16 dqkv with the block's scale, the amax recorded; shapes another kernel takes are refused."""
    ops = _ops()
    from merlot_amd.lib import MerlotHipError
    heads = 12
    g = torch.Generator(device='cuda').manual_seed(S)
    qkv = torch.randn(B * S, 3 * heads * 64, device='cuda', generator=g).bfloat16()
    valid = None
    if masked:
        valid = torch.ones(B, S, dtype=torch.uint8, device='cuda')
        valid[:, S - 7:] = 0
    out, lse = ops.attention_fwd(qkv, B, S, heads, valid)
    dout = (torch.randn(B * S, heads * 64, device='cuda', generator=g) * 1e-2).bfloat16()
    ref = ops.attention_bwd(qkv, out, dout, lse, B, S, heads, valid)
    assert ops.attention_bwd_writes_q8(S, False)
    s = FMAX[fmt] / float(ref.float().abs().max()) * 0.8
    blk = _block(s, fmt)
    dqkv, dq8 = ops.attention_bwd(qkv, out, dout, lse, B, S, heads, valid, q8_block=blk, q8_fmt=fmt)
    assert torch.equal(dqkv, ref)
    assert torch.equal(dq8[:B * S].view(torch.uint8), _q(ref, s, fmt).view(torch.uint8)) and dq8.shape[0] % 128 == 0
    assert blk[3].item() == float(ref.float().abs().max())
    assert not ops.attention_bwd_writes_q8(328, False) and ops.attention_bwd_writes_q8(328, True)
    if S > 512:
        q2 = qkv[:B * 328].contiguous()
        o2, l2 = ops.attention_fwd(q2, B, 328, heads, None)
        with pytest.raises(MerlotHipError):
            ops.attention_bwd(q2, o2, dout[:B * 328].contiguous(), l2, B, 328, heads, None, q8_block=blk, q8_fmt=fmt)


def test_scale_rotate_turns_recorded_amaxes_into_scales():
    ops = _ops()
    blocks = torch.tensor([[2.0, 0.5, 224.0, 112.0], [3.0, 1 / 3.0, 5.0, 0.0], [1.0, 1.0, 0.0, 7.0]], device='cuda')
    fmts = torch.tensor([0, 1, 1], device='cuda', dtype=torch.int32)
    ops.f8_scale_rotate(blocks, 3, fmts)
    b = blocks.cpu()
    assert b[0].tolist() == [4.0, 0.25, 112.0, 0.0]                    # e4m3: 448 / 112
    assert b[1].tolist() == [3.0, 1 / 3.0, 5.0, 0.0] or abs(b[1][1].item() - 1 / 3.0) < 1e-7   # nothing recorded: unchanged
    assert b[2].tolist() == [8192.0, 1 / 8192.0, 7.0, 0.0]            # e5m2: 57 344 / 7


@pytest.mark.parametrize('modes', ['w1,w2,fuse', 'w1,w2,wqkv,wproj,fuse,noa', 'w1,w2,fuse,noa,dgrad1', 'w1,w2,wqkv,fuse,noa,dgrad1,dgradqkv'])
def test_config5_geometry_fused_fp8_backward_three_steps(modes):
    """`fp8_backward` with 'fuse' at a batch whose row counts are multiples of 256 (16 examples of 16 frames at 384^2: 147 968 ViT rows, 45 312 joint
    rows, 8 192 text rows): step 0 calibrates every site (current scaling), steps 1 and 2 run on the producers' own copies with delayed scales.  Against
    `fp8_forward = 'ln'` with the bf16 backward on the same weights and inputs: every loss within 2e-2 (the contract of SURVEY.md 7(vii)), touched weight
    gradients within the 8-bit products' noise (rel-L2 <= 0.35, cosine >= 0.94 per tensor; medians far below), everything finite, and step 2's scales
    are the amaxes step 1 recorded."""
    from common import tiny_config, synth_batch, rel_l2
    from merlot_amd import MerlotModel, ParamStore
    from oracle import merlot_oracle as mo
    out = {}
    b = w = None
    for bwd in (False, modes):
        cfg = tiny_config(image_size=[384, 384], num_chunks_in_group=16, max_position_embeddings=1024, fp8_forward='ln', fp8_backward=bwd,
                          masking_use_attn=False)
        if b is None:
            b = synth_batch(cfg, E=16, num_chunks=16, seed=3)
            w = mo.init_weights(cfg, 0)
        st = ParamStore(cfg, 'cuda', seed=0)
        st.load_tf_weights({k: v.detach() for k, v in w.items()})
        steps = []
        for step in range(3 if bwd else 1):
            st.zero_grad()
            pm = MerlotModel(cfg, True, False, b['image'].cuda(), b['input_ids'].cuda(), mask_input=True,
                             shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']).cuda(), params=st,
                             noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
            losses = [pm.mask_loss()[0], pm.contrastive_loss()[0], pm.temporal_loss(
                torch.from_numpy(b['shuffled_idx_img']).cuda(), torch.from_numpy(b['video_src_ids']).cuda())[0]]
            sum(losses).backward()
            torch.cuda.synchronize()
            steps.append(([float(l) for l in losses], {k: v.float().cpu() for k, v in st.export_tf_grads().items()}))
            del pm
        out[bwd] = steps
        if bwd:
            f8 = st.f8_scales
            assert len(f8.calibrated) == len(f8.index) > 0
            pool = f8.pool[:len(f8.index)].cpu()
            assert torch.isfinite(pool).all() and float(pool[:, 0].min()) > 0
    l0, g0 = out[False][0]
    for step in range(3):
        l1, g1 = out[modes][step]
        for x, y in zip(l0, l1):
            assert abs(x - y) < 2e-2, (step, l0, l1)
        rels, worst, cos_min = [], 0.0, 1.0
        for k, g in g0.items():
            if float(g.norm()) == 0:
                continue
            assert torch.isfinite(g1[k]).all(), (step, k)
            r = rel_l2(g1[k], g)
            c = float((g1[k].double().flatten() @ g.double().flatten()) / (g1[k].double().norm() * g.double().norm()))
            rels.append(r)
            worst, cos_min = max(worst, r), min(cos_min, c)
            assert r < 0.35 and c > 0.94, (step, k, r, c)
        print(f'fp8_backward {modes} step {step}: losses {l1} (bf16 backward {l0}); gradients: median rel-L2 {sorted(rels)[len(rels) // 2]:.3e}, worst {worst:.3e}, min cosine {cos_min:.5f}')


def test_fused_fp8_backward_at_row_counts_that_are_no_multiple_of_anything():
    """the fused producers at ragged row counts (3 examples of 16 frames at 384^2: 27 744 ViT rows = 108 row blocks + 96 rows, 8 496 joint rows): the copies are
    allocated in whole K-tiles of 128 rows with zero padding and the ping-pong kernel's ragged last row block writes its part -- same losses and gradient noise
    as at aligned row counts."""
    from common import tiny_config, synth_batch, rel_l2
    from merlot_amd import MerlotModel, ParamStore
    from oracle import merlot_oracle as mo
    modes = 'w1,w2,wqkv,fuse,noa,dgrad1,dgradqkv'
    out = {}
    b = w = None
    for bwd in (False, modes):
        cfg = tiny_config(image_size=[384, 384], num_chunks_in_group=16, max_position_embeddings=1024, fp8_forward='ln', fp8_backward=bwd, masking_use_attn=False)
        if b is None:
            b = synth_batch(cfg, E=3, num_chunks=16, seed=4)
            w = mo.init_weights(cfg, 0)
        st = ParamStore(cfg, 'cuda', seed=0)
        st.load_tf_weights({k: v.detach() for k, v in w.items()})
        for step in range(2 if bwd else 1):
            st.zero_grad()
            pm = MerlotModel(cfg, True, False, b['image'].cuda(), b['input_ids'].cuda(), mask_input=True,
                             shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']).cuda(), params=st,
                             noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
            losses = [pm.mask_loss()[0], pm.contrastive_loss()[0], pm.temporal_loss(
                torch.from_numpy(b['shuffled_idx_img']).cuda(), torch.from_numpy(b['video_src_ids']).cuda())[0]]
            sum(losses).backward()
            torch.cuda.synchronize()
            del pm
        out[bwd] = ([float(l) for l in losses], {k: v.float().cpu() for k, v in st.export_tf_grads().items()})
        if bwd:
            assert any(k.endswith('/a') for k in st.f8_scales.calibrated) and any(k.endswith('/du') for k in st.f8_scales.calibrated)   # the fused sites ran
    (l0, g0), (l1, g1) = out[False], out[modes]
    for x, y in zip(l0, l1):
        assert abs(x - y) < 2e-2, (l0, l1)
    rels = []
    for k, g in g0.items():
        if float(g.norm()) == 0:
            continue
        assert torch.isfinite(g1[k]).all(), k
        r = rel_l2(g1[k], g)
        c = float((g1[k].double().flatten() @ g.double().flatten()) / (g1[k].double().norm() * g.double().norm()))
        rels.append(r)
        assert r < 0.35 and c > 0.94, (k, r, c)
    print(f'ragged rows, step 1 (delayed scales, fused producers): median rel-L2 {sorted(rels)[len(rels) // 2]:.3e}, worst {max(rels):.3e}')
