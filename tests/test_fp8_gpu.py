"""fp8 (OCP e4m3fn) operand path of BASELINE config #5: merlot_quantize_e4m3 + merlot_gemm_fp8_nt through the C-ABI.

The reference has no fp8 path (precision policy: utils/model_utils.py:572-602), so the checker here is arithmetic, not a
reference run: the quantiser must be BIT-EXACT against torch's own e4m3fn conversion of the same scaled values, and the
GEMM must equal an fp32 matmul of the DEQUANTISED operands (every e4m3 x e4m3 product is exact in fp32; only the
accumulation order differs) -- i.e. all of the fp8 error is the quantisation the contract allows, none is the kernel's."""
import pytest
import torch

pytestmark = pytest.mark.gpu

E4M3 = torch.float8_e4m3fn


def _ops():
    from merlot_amd import ops
    return ops


def _dev():
    return torch.device('cuda', 0)


@pytest.mark.parametrize('rows,cols,ld', [(1000, 768, 768), (37, 3072, 3072), (256, 64, 128), (5, 8, 8)])
def test_quantize_is_bit_exact_against_torch_e4m3fn(rows, cols, ld):
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(rows * 7 + cols)
    x = (torch.randn(rows, ld, generator=g) * torch.logspace(-3, 1, ld)[None, :]).to(torch.bfloat16).to(_dev())[:, :cols]
    y, scale = ops.quantize_e4m3(x)
    amax = x.float().abs().max()
    s = torch.tensor(448.0, device=x.device) / amax
    assert scale[2].item() == amax.item()
    assert scale[0].item() == s.item()
    assert abs(scale[1].item() * s.item() - 1.0) < 1e-6
    want = (x.float() * s).clamp(-448.0, 448.0).to(E4M3)
    assert torch.equal(y.view(torch.uint8), want.view(torch.uint8))


def test_quantize_zero_tensor_and_extremes():
    ops = _ops()
    x = torch.zeros(16, 64, device=_dev(), dtype=torch.bfloat16)
    y, scale = ops.quantize_e4m3(x)
    assert scale[0].item() == 1.0 and scale[1].item() == 1.0 and scale[2].item() == 0.0
    assert int(y.view(torch.uint8).max()) == 0
    x[3, 5] = -3.0e38                                      # near bf16 max: maps to -448 exactly, everything else to (signed) zero
    x[4, 6] = 1.0
    y, scale = ops.quantize_e4m3(x)
    assert y[3, 5].float().item() == -448.0
    assert y[4, 6].float().item() == 0.0


def _quant_ref(x):
    s = 448.0 / x.float().abs().max()
    q = (x.float() * s).clamp(-448.0, 448.0).to(E4M3)
    return q, s


SHAPES = [(256, 256, 256), (512, 768, 768), (4000, 768, 768), (300, 2304, 768), (9248, 3072, 768), (5000, 768, 3072),
          (777, 1000, 384), (45312, 2304, 768)]


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_gemm_fp8_equals_fp32_matmul_of_the_dequantised_operands(M, N, K):
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(_dev())
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(_dev())
    # asymmetric on purpose (a transposed or row/column-swapped store cannot pass)
    a[:, 0] += 3.0
    b[0, :] -= 0.2
    a8, sa = ops.quantize_e4m3(a)
    b8, sb = ops.quantize_e4m3(b)
    bias = torch.randn(N, generator=g).to(_dev())
    out = ops.gemm_fp8_nt(a8, sa, b8, sb, bias=bias, out_dtype=torch.float32)
    want = torch.addmm(bias, a8.float() * sa[1], (b8.float() * sb[1]).t())
    err = (out - want).abs().max().item()
    assert err <= 2e-5 * want.abs().max().item() + 1e-6, err
    # and the quantisation error itself stays where per-tensor e4m3 puts it (2^-4 relative per element, averaging down over K)
    exact = torch.addmm(bias, a.float(), b.float().t())
    rel = ((out - exact).norm() / exact.norm()).item()
    assert rel < 4e-2, rel


def test_gemm_fp8_epilogues_match_the_bf16_kernels_epilogues():
    """GELU + aux_out, residual + dropout, bf16 output: the same epilogue code as merlot_gemm_bf16_nt, driven by the fp8 main
    loop -- compared against the bf16 kernel fed the DEQUANTISED operands (exactly representable in bf16: e4m3 has 4
    significant bits), so the two differ only by accumulation order and the scale multiplication."""
    ops = _ops()
    M, N, K = 2048, 1024, 768
    g = torch.Generator(device='cpu').manual_seed(5)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(_dev())
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(_dev())
    bias = torch.randn(N, generator=g).to(_dev())
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(_dev())
    a8, sa = ops.quantize_e4m3(a)
    b8, sb = ops.quantize_e4m3(b)
    # the raw e4m3 values are exactly representable in bf16; the two scales go into alpha
    ad = a8.float().to(torch.bfloat16)
    bd = b8.float().to(torch.bfloat16)
    alpha = float(sa[1].item() * sb[1].item())
    for kw in (dict(epilogue=ops.EPI_GELU, want_aux=True), dict(epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=77),
               dict(epilogue=ops.EPI_NONE)):
        want_aux = kw.pop('want_aux', False)
        aux8 = torch.empty(M, N, device=_dev(), dtype=torch.bfloat16) if want_aux else None
        aux16 = torch.empty(M, N, device=_dev(), dtype=torch.bfloat16) if want_aux else None
        o8 = ops.gemm_fp8_nt(a8, sa, b8, sb, bias=bias, aux_out=aux8, **kw)
        o16 = ops.gemm_nt(ad, bd, bias=bias, alpha=alpha, aux_out=aux16, **kw)
        d = (o8.float() - o16.float()).abs()
        # bf16 outputs: an accumulation-order difference can flip the last bit of a few outputs, never more
        assert d.max().item() <= 2.0 ** -7 * o16.float().abs().max().item(), d.max().item()
        assert (d > 0).float().mean().item() < 0.02
        if want_aux:
            da = (aux8.float() - aux16.float()).abs()
            assert da.max().item() <= 2.0 ** -7 * aux16.float().abs().max().item()
        if 'dropout_p' in kw:                             # the same counter-based mask: identical zeros
            z8 = (o8.float() - res.float()) == 0
            z16 = (o16.float() - res.float()) == 0
            assert (z8 != z16).float().mean().item() < 1e-3


@pytest.mark.parametrize('rows,H', [(1000, 768), (37, 256), (4, 1024)])
def test_ln_fwd_q8_emits_the_per_row_e4m3_copy_of_its_own_bf16_output(rows, H):
    """merlot_ln_fwd_q8: y_bf16 / mean / rstd identical to merlot_ln_fwd, y_fp8 = e4m3(bf16(y) * 448 / max|bf16(y[row])|) bit for bit
    against torch's conversion, row_scale = max / 448; an all-zero row (gamma = beta = 0) gets scale 1."""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(rows + H)
    x = (torch.randn(rows, H, generator=g) * 3 + 0.5).to(torch.bfloat16).to(_dev())
    gamma = (1 + 0.2 * torch.randn(H, generator=g)).to(_dev())
    beta = (0.1 * torch.randn(H, generator=g)).to(_dev())
    y16, _, mean, rstd = ops.ln_fwd(x, gamma, beta)
    q16, q8, rs, qmean, qrstd = ops.ln_fwd_q8(x, gamma, beta)
    assert torch.equal(q16, y16) and torch.equal(qmean, mean) and torch.equal(qrstd, rstd)
    amax = y16.float().abs().amax(dim=1)
    s = torch.tensor(448.0, device=x.device) / amax
    assert torch.allclose(rs, 1.0 / s, rtol=2e-7, atol=0)
    want = (y16.float() * s[:, None]).clamp(-448.0, 448.0).to(E4M3)
    assert torch.equal(q8.view(torch.uint8), want.view(torch.uint8))
    z16, z8, zs, _, _ = ops.ln_fwd_q8(x, torch.zeros_like(gamma), torch.zeros_like(beta))
    assert int(z8.view(torch.uint8).max()) == 0 and float(zs.min()) == 1.0 == float(zs.max())


@pytest.mark.parametrize('M,N,K', [(512, 768, 768), (4000, 2304, 768), (777, 1000, 384)])
def test_gemm_fp8_with_per_row_scales(M, N, K):
    """a_row_scale: row m of the product is multiplied by its own dequantisation factor -- interior and edge tiles, bf16 and
    f32 outputs, with the GELU epilogue (the scale must be applied BEFORE bias and activation)."""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(M * 3 + N)
    a = (torch.randn(M, K, generator=g) * torch.logspace(-2, 1, M)[:, None]).to(torch.bfloat16).to(_dev())   # rows of very different size
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(_dev())
    bias = torch.randn(N, generator=g).to(_dev())
    amax = a.float().abs().amax(dim=1)
    s = 448.0 / amax
    a8 = (a.float() * s[:, None]).clamp(-448, 448).to(E4M3)
    rs = (1.0 / s).contiguous()
    b8, sb = ops.quantize_e4m3(b)
    out = ops.gemm_fp8_nt(a8, None, b8, sb, bias=bias, out_dtype=torch.float32, a_row_scale=rs)
    want = ((a8.double() @ b8.double().t()) * rs.double()[:, None] * sb[1].double() + bias.double()).float()   # exact products, fp64 sums
    # per row (the rows differ by 3 orders of magnitude), relative to the row's largest output
    assert ((out - want).abs().amax(dim=1) / want.abs().amax(dim=1)).max().item() <= 1e-4      # measured 3.0e-5 (the MFMA's fp32 tree sum); a neighbouring row's scale would be 1.4e-2 off
    exact = torch.addmm(bias, a.float(), b.float().t())
    # per-row scaling keeps the SMALL rows accurate too (a per-tensor scale would flush them): row-wise relative error
    rel_rows = (out - exact).norm(dim=1) / exact.norm(dim=1)
    assert rel_rows.max().item() < 8e-2
    u8 = torch.empty(M, N, device=_dev(), dtype=torch.bfloat16)
    o8 = ops.gemm_fp8_nt(a8, None, b8, sb, bias=bias, epilogue=ops.EPI_GELU, aux_out=u8, a_row_scale=rs)
    assert (u8.float() - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()
    assert (o8.float() - torch.nn.functional.gelu(want)).abs().max().item() <= 2.0 ** -7 * want.abs().max().item() + 1e-3
    with pytest.raises(ValueError, match='one entry per row'):
        ops.gemm_fp8_nt(a8, None, b8, sb, a_row_scale=rs[:-1])


def _attn_ref(qkv, B, S, heads, valid=None):
    """fp32 softmax attention on the bf16 values (utils/transformer.py:98-127 semantics, -1e10 on padded keys)."""
    D = heads * 64
    x = qkv.float().view(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    sc = (q @ k.transpose(-1, -2)) * 0.125
    if valid is not None:
        sc = sc + (1.0 - valid.float())[:, None, None, :] * -1e10
    return (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, D)


def test_gemm_fp8_rejects_what_the_kernel_cannot_take():
    ops = _ops()
    a = torch.randn(256, 192, device=_dev()).to(torch.bfloat16)
    a8, sa = ops.quantize_e4m3(a)
    with pytest.raises(RuntimeError, match='K %'):
        ops.gemm_fp8_nt(a8, sa, a8, sa)                  # K = 192 is not a multiple of 128
    a = torch.randn(256, 128, device=_dev()).to(torch.bfloat16)
    a8, sa = ops.quantize_e4m3(a)
    with pytest.raises(RuntimeError, match='K >= 256'):
        ops.gemm_fp8_nt(a8, sa, a8, sa)


def test_config5_geometry_fp8_forward_loss_within_2e2_of_bf16():
    """BASELINE config #5 geometry (384^2 frames, 16-segment groups: Sv = 578, joint S = 2832) with `fp8_forward`: the QKV /
    fc1 / fc2 GEMMs of all three stacks on e4m3 operands.  Contract: every loss
    within 2e-2 of the bf16 path on the same weights and inputs; the backward (bf16) still runs and its gradients stay
    close to the bf16 path's."""
    from common import tiny_config, synth_batch, rel_l2
    from merlot_amd import MerlotModel, ParamStore
    from oracle import merlot_oracle as mo
    out = {}
    b = None
    for fp8 in (False, True, 'ln'):
        cfg = tiny_config(image_size=[384, 384], num_chunks_in_group=16, max_position_embeddings=1024, fp8_forward=fp8,
                          masking_use_attn=False, attention_log_in_backward=True)      # MLM targets from the noise alone: identical in both runs
        if b is None:
            b = synth_batch(cfg, E=1, num_chunks=16, seed=3)
            w = mo.init_weights(cfg, 0)
        st = ParamStore(cfg, 'cuda', seed=0)
        st.load_tf_weights({k: v.detach() for k, v in w.items()})
        st.zero_grad()
        pm = MerlotModel(cfg, True, False, b['image'].cuda(), b['input_ids'].cuda(), mask_input=True,
                         shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']).cuda(), params=st,
                         noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
        l1, l2, l3 = pm.mask_loss()[0], pm.contrastive_loss()[0], pm.temporal_loss(
            torch.from_numpy(b['shuffled_idx_img']).cuda(), torch.from_numpy(b['video_src_ids']).cuda())[0]
        (l1 + l2 + l3).backward()
        torch.cuda.synchronize()
        out[fp8] = dict(losses=[float(l1), float(l2), float(l3)], viz=pm.encoder_hidden_states['viz'].float().cpu(),
                        log=torch.stack([pm.attention_log[k] for k in sorted(pm.attention_log)]).float().cpu(),
                        lang=pm.encoder_hidden_states['lang'].float().cpu(),
                        masked=pm.lang_mask_info['masked_idx'].cpu(), grads={k: v.float().cpu() for k, v in st.export_tf_grads().items()})
    for mode in (True, 'ln'):     # QKV + fc1 (per-row, from the LayerNorm) + fc2 (per-tensor) | without fc2
        for a, c in zip(out[False]['losses'], out[mode]['losses']):
            assert abs(a - c) < 2e-2, (mode, out[False]['losses'], out[mode]['losses'])
        assert abs(sum(out[False]['losses']) - sum(out[mode]['losses'])) < 2e-2
        assert torch.equal(out[mode]['masked'], out[False]['masked'])
        # the attention log is complete after the step in every mode (with attention_log_in_backward and fp8_forward = 'all'
        # it stayed all zeros -- the fp8 attention has no log in its backward and nobody finished the forward's)
        assert abs(float(out[mode]['log'].sum()) - 1.0) < 1e-3 and float((out[mode]['log'] - out[False]['log']).abs().max()) < 2e-2, (mode, out[mode]['log'])
        for k in ('viz', 'lang'):
            assert rel_l2(out[mode][k], out[False][k]) < 5e-2, (mode, k)
        assert all(torch.isfinite(g).all() for g in out[mode]['grads'].values())
        rels = [rel_l2(out[mode]['grads'][k], g) for k, g in out[False]['grads'].items() if float(g.norm()) > 0]
        assert sorted(rels)[len(rels) // 2] < 0.15, (mode, sorted(rels)[len(rels) // 2])
