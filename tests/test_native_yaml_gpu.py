"""GPU: the as-shipped merlot.yaml geometry through the whole HIP model.  model/configs/merlot.yaml:30,36 ships
`resnet_layers: [3, 4, 9]` at `image_size: [192, 352]` -- a NON-SQUARE frame (12 x 22 patches, 6 x 11 after pooling: 266 ViT tokens per
frame, joint S = 4 * 67 + 128 = 396) and the ResNet-hybrid stem at three times the depth every other test runs it.  Two problems,
both executed by the reference's own program under the TensorFlow shim (tests/golden/make_reference_golden.py native ->
ref_shim_native.npz): `p64x96` (patch stem, 64 x 96) and `r192x352` (the shipped frame and stem depth, 2 + 2 + 2 transformer layers,
one example of four segments).  What would catch an h / w swap: position_embedder2d, the 2 x 2 pooling and `final_pe` index the
frame as [h1, w1] (utils/model_utils.py:710-739, utils/vision_transformer.py:255-267, model/modeling.py:99-126); the fixture keeps the
ViT rows at both ends of the first and of the last patch row of every frame.

Bounds (measured values: profiles/r05_k_native_shapes.txt): integer outputs exact; hidden states rel-L2 <= 2e-2 on the patch stem,
<= 4e-2 behind the deep stem against the fp32 reference run and <= 3e-2 against the oracle under the reference's own bf16 policy
(measured 2.3e-2 / 1.9e-2; the policy alone moves the fp32 stem output by 3.2 %); losses <= 1e-2; gradients per tensor class.
The deep stem runs with the residual branches damped (tests/native_shapes.py: a randomly initialised [3, 4, 9] stem is chaotic in
bf16 -- the reference's own bf16 policy against its fp32: 36 % -- and nothing can be compared on it); its own 164 gradients are compared
by direction (cosine) and by the directional derivative of the HIP forward, as tests/test_stem_model_gpu.py does at depth [1, 1, 2]."""
import numpy as np
import pytest
import torch

from common import rel_l2, head
from grad_parity import tensor_class
import native_shapes as ns
from oracle import merlot_oracle as mo

pytestmark = pytest.mark.gpu


def _forward_checks(name, hid_ref, hid_lang):
    cfg, batch, w, noise, fx = ns.load(name)
    pm, losses, g = ns.run_hip(cfg, batch, w, noise)
    assert (pm.P, pm.L) == tuple(int(v) for v in fx['P_L'])
    assert np.array_equal(pm.lang_mask_info['masked_idx'].cpu().numpy(), fx['masked_idx'])
    assert np.array_equal(pm.lang_mask_info['masked_ids'].cpu().numpy(), fx['masked_ids'])
    assert rel_l2(pm.lang_transformer_info['attention_summs'].reshape(pm.B, pm.L), torch.from_numpy(fx['attention_summs'])) < 1e-2
    hs = pm.vision_transformer_info['hidden_state'].float().cpu()
    assert tuple(hs.shape) == tuple(int(v) for v in fx['vit_hidden_shape'])
    assert rel_l2(hs[:, torch.from_numpy(fx['vit_rows']).long(), :], torch.from_numpy(fx['vit_hidden_rows'])) < hid_ref
    assert rel_l2(pm.encoder_hidden_states['viz'], torch.from_numpy(fx['encoder_viz'])) < hid_ref
    assert rel_l2(pm.img_trg_h, torch.from_numpy(fx['img_trg_h'])) < hid_lang
    assert rel_l2(pm.lang_trg_h, torch.from_numpy(fx['lang_trg_h'])) < hid_lang
    assert rel_l2(pm.encoder_hidden_states['lang'], torch.from_numpy(fx['encoder_lang'])) < hid_lang
    for a, b in zip(losses, fx['losses']):
        assert abs(a - float(b)) < 1e-2, (losses, fx['losses'])
    assert abs(sum(losses) - float(fx['loss'])) < 2e-2
    return cfg, batch, w, noise, fx, pm, g


def _norm_ratios(g, fx, sel):
    norms = dict(zip([str(n) for n in fx['grad_names']], fx['grad_norms']))
    return [abs(float(g[n].double().norm()) - v) / v for n, v in norms.items() if sel(n) and not n.endswith('key_layer/bias')]


REL = {'bias': 6e-2, 'ln': 6e-2, 'pos': 6e-2, 'emb': 3e-2, 'kernel': 6e-2}      # measured <= 5.6e-2 (the contrastive head at 8 segments), 4.3e-2 kernels


def test_non_square_frame_on_the_patch_stem_matches_the_reference_program():
    cfg, batch, w, noise, fx, pm, g = _forward_checks('p64x96', 2e-2, 2e-2)
    r = _norm_ratios(g, fx, lambda n: True)
    assert len(r) == 111 and np.median(r) < 1e-2 and max(r) < 5e-2, (np.median(r), max(r))          # measured 1.7e-3 / 2.5e-2
    for k in fx:
        if k.startswith('grad/'):
            n = k[5:]
            assert rel_l2(torch.from_numpy(head(g[n].numpy())), torch.from_numpy(fx[k])) < REL[tensor_class(n)], n
    # every gradient, by tensor class, against the fp32 oracle on the same problem (the fixture holds heads of a few tensors only)
    _, lo, go = ns.run_oracle(cfg, batch, w, noise, False)
    bad = []
    for n, gr in go.items():
        if n.endswith('key_layer/bias') or float(gr.norm()) == 0.0:
            continue
        rel = float((g[n].double() - gr.double()).norm() / gr.double().norm())
        ratio = abs(float(g[n].double().norm() / gr.double().norm()) - 1.0)
        if rel > REL[tensor_class(n)] or ratio > 3e-2:                                               # measured <= 5.6e-2 / 2.5e-2
            bad.append((n, rel, ratio))
    assert not bad, bad


def test_as_shipped_frame_and_stem_depth_match_the_reference_program():
    # (bounds at 2.5 - 3 x the measured values: the GroupNorm statistics of the 16-block stem are summed with fp32 atomics, two runs of this test differ in
    # the second digit of these numbers, and one run in ~25 of the whole suite crossed a bound set at 1.6 x)
    cfg, batch, w, noise, fx, pm, g = _forward_checks('r192x352', 6e-2, 3e-2)
    assert cfg['image_size'] == [192, 352] and cfg['resnet_layers'] == [3, 4, 9] and pm.P == 4 * (6 * 11 + 1)
    rest = _norm_ratios(g, fx, lambda n: not ns.is_stem(n))
    stem = _norm_ratios(g, fx, ns.is_stem)
    assert len(rest) == 109 and np.median(rest) < 1e-2 and max(rest) < 3e-2, (np.median(rest), max(rest))      # measured 1.4e-3 / 7.9e-3
    assert len(stem) == 164 and np.median(stem) < 2e-2 and max(stem) < 1.5e-1, (np.median(stem), max(stem))    # measured 3.4e-3 / 3.3e-2
    # against the oracle under the reference's bf16 policy for the stem (what the HIP stem implements)
    m, lo, go = ns.run_oracle(cfg, batch, w, noise, True)
    assert rel_l2(pm.vision_transformer_info['hidden_state'], m.vision_transformer_info['hidden_state']) < 5e-2      # measured 1.9e-2
    assert rel_l2(pm.encoder_hidden_states['viz'], m.encoder_hidden_states['viz']) < 3.5e-2                          # measured 1.2e-2
    bad, cos = [], {}
    for n, gr in go.items():
        if n.endswith('key_layer/bias') or float(gr.norm()) == 0.0:
            continue
        a, b = g[n].double().flatten(), gr.double().flatten()
        if ns.is_stem(n) and not n.endswith('conv_postresnet_proj/bias'):
            cos[n] = float(torch.dot(a, b) / (a.norm() * b.norm()))
            continue
        rel = float((a - b).norm() / b.norm())
        # behind the deep stem the ViT's own gradients inherit its 2 % activation noise: measured <= 5.1e-2 (layer01 query kernel)
        if rel > 1.2e-1 or abs(float(a.norm() / b.norm()) - 1.0) > 3e-2:
            bad.append((n, rel))
    assert not bad, bad
    assert len(cos) == 163 and min(cos.values()) > 0.90 and np.median(list(cos.values())) > 0.985, \
        (min(cos.values()), np.median(list(cos.values())))                                           # measured 0.960 / 0.997


def test_deep_stem_gradient_is_the_derivative_of_the_hip_forward_on_the_shipped_frame():
    """The directional-derivative check of tests/test_stem_model_gpu.py at depth [3, 4, 9] and 192 x 352: L(w) = <viz hidden states,
    cot> evaluated by the HIP forward at w +- h d for directions d built from the ORACLE's gradients (all 164 stem tensors together,
    and each third alone); the central difference, <g_hip, d> and <g_oracle, d> must agree."""
    from merlot_amd import MerlotModel, ParamStore
    cfg, batch, w, noise, fx = ns.load('r192x352')
    w = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    with mo.bf16_stem():
        m = mo.MerlotOracle(cfg, w, batch['image'], batch['input_ids'], mask_input=False, shuffled_idx_img=batch['shuffled_idx_img'])
    cot = torch.randn(m.encoder_hidden_states['viz'].shape, generator=torch.Generator().manual_seed(0))
    (m.encoder_hidden_states['viz'] * cot).sum().backward()
    stem = sorted(k for k, v in w.items() if v.grad is not None and ns.is_stem(k))
    assert len(stem) == 164

    def hip(weights, backward):
        st = ParamStore(cfg, 'cuda', seed=0)
        st.load_tf_weights({k: v.detach() for k, v in weights.items()})
        st.zero_grad()
        with (torch.enable_grad() if backward else torch.no_grad()):
            pm = MerlotModel(cfg, True, False, batch['image'].cuda(), batch['input_ids'].cuda(), mask_input=False,
                             shuffled_idx_img=torch.from_numpy(batch['shuffled_idx_img']).cuda(), params=st)
            loss = (pm.encoder_hidden_states['viz'].double() * cot.cuda().double()).sum()
            if backward:
                loss.float().backward()
        torch.cuda.synchronize()
        return float(loss), (st.export_tf_grads() if backward else None)

    _, g_hip = hip(w, True)
    for names in [stem] + [stem[i::3] for i in range(3)]:
        d = {k: (w[k].grad / (w[k].grad.norm() + 1e-30)) * w[k].detach().norm() for k in names}
        h = 1e-4
        plus = {k: (v.detach() + h * d[k]) if k in d else v.detach() for k, v in w.items()}
        minus = {k: (v.detach() - h * d[k]) if k in d else v.detach() for k, v in w.items()}
        fd = (hip(plus, False)[0] - hip(minus, False)[0]) / (2 * h)
        a_hip = sum(float((g_hip[k].float().cpu() * d[k]).sum()) for k in names)
        a_orc = sum(float((w[k].grad * d[k]).sum()) for k in names)
        print(f'{len(names)} tensors: <g_hip, d> {a_hip:.4e}  <g_oracle, d> {a_orc:.4e}  central difference {fd:.4e}')
        tol = 0.08 if len(names) == len(stem) else 0.15       # measured 2.8 % / 5.6 % at worst (profiles/r05_l_native_tests.txt): the same atomics noise as above
        assert abs(a_hip - a_orc) < tol * abs(a_orc), (len(names), a_hip, a_orc, fd)
        assert abs(a_hip - fd) < tol * abs(fd), (len(names), a_hip, a_orc, fd)
