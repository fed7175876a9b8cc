"""CPU: the host logic of merlot_amd (MerlotModel mirror, autograd wiring, parameter arena) with the HIP ops
swapped for their torch emulation (tests/emu_ops.py), checked against the oracle on BASELINE config #1.
This isolates host-side bugs from kernel bugs; the kernels themselves are checked on the GPU (-m gpu)."""
import numpy as np
import pytest
import torch

from common import tiny_config, synth_batch, rel_l2, head
from oracle import merlot_oracle as mo


@pytest.fixture(scope='module')
def oracle_run():
    cfg = tiny_config()
    w = mo.init_weights(cfg, 0)
    for t in w.values():
        t.requires_grad_(True)
    b = synth_batch(cfg)
    m = mo.MerlotOracle(cfg, w, b['image'], b['input_ids'], mask_input=True, shuffled_idx_img=b['shuffled_idx_img'],
                        noise=b['noise'])
    loss, info = m.total_loss(b['shuffled_idx_img'], b['video_src_ids'])
    loss.backward()
    return cfg, w, b, m, loss, info


def test_model_forward_backward_matches_oracle(emu, oracle_run):
    from merlot_amd import MerlotModel, ParamStore
    cfg, w, b, m, loss, info = oracle_run
    st = ParamStore(cfg, 'cpu', seed=0)
    assert st.load_tf_weights({k: v.detach() for k, v in w.items()}) == []
    pm = MerlotModel(cfg, True, False, b['image'], b['input_ids'], mask_input=True,
                     shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']), params=st,
                     noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
    # shape algebra identical to model/modeling.py:234-248
    assert (pm.B, pm.L, pm.P, pm.viz_chunk_length) == (m.B, m.L, m.P, m.viz_chunk_length) == (2, 128, 20, 5)
    # integer outputs bit-exact
    assert np.array_equal(pm.lang_mask_info['masked_idx'].numpy(), m.lang_mask_info['masked_idx'].numpy())
    assert np.array_equal(pm.lang_mask_info['masked_ids'].numpy(), m.lang_mask_info['masked_ids'].numpy())
    assert rel_l2(pm.lang_transformer_info['attention_summs'].reshape(pm.B, pm.L), m.attention_summs()) < 1e-3
    for k in ['viz', 'lang']:
        assert rel_l2(pm.encoder_hidden_states[k], m.encoder_hidden_states[k]) < 2e-2
    assert rel_l2(pm.img_trg_h, m.img_trg_h) < 2e-2 and rel_l2(pm.lang_trg_h, m.lang_trg_h) < 2e-2
    l1, i1 = pm.mask_loss()
    l2, i2 = pm.contrastive_loss()
    l3, i3 = pm.temporal_loss(torch.from_numpy(b['shuffled_idx_img']), torch.from_numpy(b['video_src_ids']))
    assert abs(float(l1) - float(info['lang']['loss'])) < 1e-2
    assert abs(float(l2) - float(info['contr']['loss_all'])) < 1e-2
    assert abs(float(l3) - float(info['temporal']['loss'])) < 1e-2
    for k, v in m.attention_log.items():
        assert abs(float(pm.attention_log[k]) - float(v)) < 1e-3
    st.zero_grad()
    (l1 + l2 + l3).backward()
    gt = st.export_tf_grads()
    for k, v in w.items():
        if v.grad is None or k.endswith('key_layer/bias'):       # key bias grad is identically 0 (softmax shift)
            continue
        assert rel_l2(gt[k], v.grad) < 0.12, (k, rel_l2(gt[k], v.grad))
    rels = [rel_l2(gt[k], v.grad) for k, v in w.items() if v.grad is not None and not k.endswith('key_layer/bias')]
    assert np.median(rels) < 2.5e-2


def test_tf_weight_roundtrip(emu):
    from merlot_amd import ParamStore
    cfg = tiny_config()
    w = mo.init_weights(cfg, 3)
    st = ParamStore(cfg, 'cpu', init=False)
    st.load_tf_weights(w)
    out = st.export_tf_weights()
    assert set(out) == set(w)
    for k in w:
        assert torch.equal(out[k], w[k]), k
    # arena layout: every parameter view starts 256-byte aligned, groups contiguous
    for name, (off, n, shp) in st.offsets.items():
        assert off % 64 == 0
    s0, e0 = st.group_range('encoder/layer00/')
    s1, e1 = st.group_range('encoder/layer01/')
    assert e0 == s1 and e1 > s1


def test_inference_mode_without_masking(emu, oracle_run):
    """downstream use: mask_input=False, shuffled_idx_img=None, is_training=False (get_zero_shot_logits.py:58-66)."""
    from merlot_amd import MerlotModel, ParamStore
    cfg, w, b, _, _, _ = oracle_run
    st = ParamStore(cfg, 'cpu', seed=0)
    st.load_tf_weights({k: v.detach() for k, v in w.items()})
    with torch.no_grad():
        ref = mo.MerlotOracle(cfg, {k: v.detach() for k, v in w.items()}, b['image'], b['input_ids'], mask_input=False,
                              log_attention_probs=False)
        pm = MerlotModel(cfg, False, False, b['image'], b['input_ids'], mask_input=False, params=st,
                         log_attention_probs=False)
    assert rel_l2(pm.encoder_hidden_states['lang'], ref.encoder_hidden_states['lang']) < 2e-2
    xa, xb = pm.pooled_segments()
    ra, rb = ref.pooled_segments()
    assert rel_l2(pm.allpairs_temporal_logits(xa, xb, 'lang_viz_temporal'),
                  ref.allpairs_temporal_logits(ra, rb, 'lang_viz_temporal')) < 3e-2


def test_constructor_errors_mirror_reference(emu):
    from merlot_amd import MerlotModel, ParamStore
    cfg = tiny_config()
    st = ParamStore(cfg, 'cpu', seed=0)
    b = synth_batch(cfg)
    with pytest.raises(ValueError):
        MerlotModel(cfg, True, False, b['image'], b['input_ids'][:, :3], params=st)        # image/ids batch mismatch
    with pytest.raises(ValueError):
        MerlotModel(tiny_config(num_chunks_in_group=3), True, False, b['image'], b['input_ids'], params=st)
    with pytest.raises(NotImplementedError):                                                # VCR-style duplication
        MerlotModel(tiny_config(num_texts=4), True, False, b['image'], b['input_ids'], params=st)
    with pytest.raises(ValueError):                                                         # hybrid stem needs P = 16
        MerlotModel(tiny_config(resnet_layers=[1, 1, 1], patch_size=8), True, False, b['image'], b['input_ids'], params=st)
    with pytest.raises(ValueError):
        MerlotModel(cfg, True, False, b['image'], b['input_ids'])                          # no params store


def test_mask_inputs_kernel_algorithm_vs_oracle(emu):
    """the rank-by-counting formulation used by csrc/index.hip (transcribed in emu_ops.mask_inputs) == oracle."""
    import os
    from merlot_amd.modeling import masking_constants
    from oracle import index_oracle as ix
    k = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'index_kat.npz'))
    cfg = tiny_config()
    for tag, L in [('L128', 128), ('L160', 160)]:
        c = masking_constants(L, cfg)
        oc = ix.masking_constants(L, cfg)
        assert np.float32(c['w_nontopk']) == oc['nontopk_val']
        assert np.float32(c['w_topk']) == np.float32(1.0) * oc['topk_minus_nontopk'] + oc['nontopk_val']
        t = lambda n: torch.from_numpy(k[f'{tag}_{n}'])
        B = k[f'{tag}_ids'].shape[0]
        mids, midx = emu.mask_inputs(t('ids'), t('summ'), t('gumbel'), t('span_lower'), t('span_upper'),
                                     t('random_ids').reshape(B, L), t('option').reshape(B, L), c['num_topk'],
                                     c['num_to_mask'], c['w_nontopk'], c['w_topk'], c['log_nontopk'], c['log_topk'],
                                     c['max_weight'])
        assert np.array_equal(mids.numpy(), k[f'{tag}_masked_ids']) and np.array_equal(midx.numpy(), k[f'{tag}_masked_idx'])


def test_model_fn_builder_mirrors_reference_model_fn(emu, oracle_run):
    """model_fn (model/modeling.py:671-713): features dict in, summed loss + metric names out; `transpose_input`
    (HWCN infeed layout, :683-685) and the non-training reshape (:686-687) handled as the reference does."""
    from merlot_amd import ParamStore, model_fn_builder
    from merlot_amd.config import NeatConfig
    cfg, w, b, m, loss, info = oracle_run
    outs = {}
    for transposed in (False, True):
        config = NeatConfig.from_dict({'model': dict(cfg, transpose_input=transposed),
                                       'data': {'num_chunks': 4, 'chunk_text_len': 32},
                                       'device': {'use_tpu': False, 'output_dir': '/tmp/unused'}, 'optimizer': {}})
        st = ParamStore(cfg, 'cpu', seed=0)
        st.load_tf_weights({k: v.detach() for k, v in w.items()})
        images = b['image'].permute(1, 2, 3, 0).contiguous() if transposed else b['image']
        features = {'images': images, 'input_ids': b['input_ids'], 'shuffled_idx_img': torch.from_numpy(b['shuffled_idx_img']),
                    'video_src_ids': torch.from_numpy(b['video_src_ids']),
                    'noise': {k: torch.from_numpy(v) for k, v in b['noise'].items()}}
        outs[transposed] = model_fn_builder(config)(features, None, 'train', {'store': st})
    for o in outs.values():
        assert abs(float(o['loss']) - float(loss)) < 3e-2
        assert {'lang/loss', 'lang/acc', 'contr/lang_to_viz', 'contr/viz_to_lang', 'contr/loss_all', 'temporal/loss',
                'temporal/lang_viz_loss', 'temporal/viz_viz_acc', 'attn/encoder/viz2lang'} <= set(o['metrics'])
    assert float(outs[True]['loss']) == float(outs[False]['loss'])


def test_resnet_hybrid_stem_matches_oracle(emu):
    """`resnet_layers: [1, 1, 2]` (projection shortcuts, two stride-2 groups, one identity block) through
    the host code with emulated kernels vs the oracle: stem tokens, ViT hidden state and the gradients of every stem
    variable.  Compared with the oracle's bf16-policy variant of the stem (rounding where the reference's bf16 graph
    rounds: 23 layers in sequence move the fp32 result by ~4 %, see oracle.merlot_oracle.bf16_stem) and, loosely, with
    the fp32 graph."""
    from merlot_amd import MerlotModel, ParamStore
    cfg = tiny_config(resnet_layers=[1, 1, 2])
    w = mo.init_weights(cfg, 2)
    for t in w.values():
        t.requires_grad_(True)
    b = synth_batch(cfg, E=1, num_chunks=4, seed=4)
    m32 = mo.MerlotOracle(cfg, w, b['image'], b['input_ids'], mask_input=False, shuffled_idx_img=b['shuffled_idx_img'])
    with mo.bf16_stem():
        m = mo.MerlotOracle(cfg, w, b['image'], b['input_ids'], mask_input=False, shuffled_idx_img=b['shuffled_idx_img'])
    g = torch.Generator().manual_seed(0)
    cot = torch.randn(m.encoder_hidden_states['viz'].shape, generator=g)
    (m.encoder_hidden_states['viz'] * cot).sum().backward()
    st = ParamStore(cfg, 'cpu', seed=0)
    assert st.load_tf_weights({k: v.detach() for k, v in w.items()}) == []
    st.zero_grad()
    pm = MerlotModel(cfg, True, False, b['image'], b['input_ids'], mask_input=False,
                     shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']), params=st)
    assert rel_l2(pm.vision_transformer_info['hidden_state'], m.vision_transformer_info['hidden_state']) < 2e-2
    assert rel_l2(pm.encoder_hidden_states['viz'], m.encoder_hidden_states['viz']) < 2e-2
    assert rel_l2(pm.encoder_hidden_states['viz'], m32.encoder_hidden_states['viz']) < 6e-2      # vs the fp32 graph
    (pm.encoder_hidden_states['viz'] * cot).sum().backward()
    gt = st.export_tf_grads()
    rels = {k: rel_l2(gt[k], v.grad) for k, v in w.items()
            if v.grad is not None and ('resnet50lite' in k or 'conv_postresnet_proj' in k)}
    assert len(rels) == 56
    # bf16 values AND bf16 gradients through 23 layers, 64 positions at the deep end, an incoherent cotangent: two
    # realisations of that rounding noise differ by 15-25 % per tensor (the bf16-policy graph itself is 32 % median /
    # 40 % max away from the fp32 graph on this problem).  The wiring is pinned exactly by the fp32 test below.
    bad = {k: r for k, r in rels.items() if r > 0.45}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    assert np.median(list(rels.values())) < 0.25


@pytest.mark.parametrize('implicit', [True, False])
def test_resnet_hybrid_stem_autograd_wiring_exact_in_fp32(emu, monkeypatch, implicit):
    """The stem's host code (tape, branch joins, weight-standardisation backward, im2col / col2im plumbing resp. the implicit 3x3
    convolution with its flipped-tap input gradient) with every emulated kernel and every cast switched to fp32: tokens and all 56
    gradients must then agree with the fp32 oracle to round-off."""
    import emu_ops
    from merlot_amd import layers as L, ParamStore
    import merlot_amd.ops as real_ops
    monkeypatch.setattr(emu_ops, 'BF16', torch.float32)
    monkeypatch.setattr(L, 'BF16', torch.float32)
    monkeypatch.setattr(real_ops, 'cast_bf16', lambda x, out=None: x.float() if out is None else out.copy_(x))
    emu_gemm_nt = real_ops.gemm_nt                           # its default out_dtype was bound to bf16 at definition
    monkeypatch.setattr(real_ops, 'gemm_nt', lambda *a, **k: emu_gemm_nt(*a, **{**{'out_dtype': torch.float32}, **k}))
    cfg = tiny_config(resnet_layers=[1, 1, 2], resnet_implicit_conv=implicit)
    w = mo.init_weights(cfg, 2)
    for t in w.values():
        t.requires_grad_(True)
    b = synth_batch(cfg, E=1, num_chunks=4, seed=4)
    scope = 'vision_backbone/vision_transformer'
    c = mo.lite_resnet50(b['image'] - 0.5, w, scope, cfg['resnet_layers'])
    pk = w[f'{scope}/conv_postresnet_proj/kernel']
    ref = c.reshape(-1, c.shape[-1]) @ pk.reshape(pk.shape[2], -1) + w[f'{scope}/conv_postresnet_proj/bias']
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(0))
    (ref * cot).sum().backward()
    st = ParamStore(cfg, 'cpu', seed=0)
    st.load_tf_weights({k: v.detach() for k, v in w.items()})
    st.bf16 = st.master.clone()                              # fp32 "working copies"
    st.version = st.master_version
    lin = st.lin(f'{scope}/conv_postresnet_proj', need_T=False)
    lin.wbT = lin.w.t().contiguous()
    st._lins[lin.name] = lin
    st.zero_grad()
    tok = L.ResNetStemFn.apply(b['image'].float(), st, cfg, torch.zeros(1, requires_grad=True))
    assert rel_l2(tok, ref) < 1e-4
    (tok * cot).sum().backward()
    gt = st.export_tf_grads()
    for k, v in w.items():
        if v.grad is not None:
            assert rel_l2(gt[k], v.grad) < 2e-3, (k, rel_l2(gt[k], v.grad))


def test_two_live_stem_graphs_share_the_standardised_kernels_safely(emu):
    """StemWeights' buffers are one set per store and tape entries hold views of them.  (a) A second stem forward from the
    SAME master weights between a graph's forward and its backward (an evaluation pass inside a training step) leaves the older
    graph's gradients unchanged; (b) once the master weights have changed and another forward has re-standardised, the older
    graph's backward refuses instead of back-propagating with the newer kernels."""
    from merlot_amd import layers as L, ParamStore
    cfg = tiny_config(resnet_layers=[1, 1, 2])
    w = mo.init_weights(cfg, 2)
    b = synth_batch(cfg, E=1, num_chunks=4, seed=4)
    img = b['image'].to(torch.bfloat16)

    def grads(second_forward, change_weights=False):
        st = ParamStore(cfg, 'cpu', seed=0)
        st.load_tf_weights(w)
        st.refresh(True)
        st.zero_grad()
        tok = L.ResNetStemFn.apply(img, st, cfg, torch.zeros(1, requires_grad=True))
        if change_weights:
            st.master.mul_(1.01)
            st.master_version += 1
        if second_forward:
            with torch.no_grad():
                L.ResNetStemFn.apply(img, st, cfg, None)
        cot = torch.randn(tok.shape, generator=torch.Generator().manual_seed(0)).to(tok.dtype)
        (tok.float() * cot.float()).sum().backward()
        return {k: v.clone() for k, v in st.export_tf_grads().items() if 'resnet50lite' in k}

    g0, g1 = grads(False), grads(True)
    assert len(g0) == 54 and all(torch.equal(g0[k], g1[k]) for k in g0)
    with pytest.raises(RuntimeError, match='master weights changed'):
        grads(True, change_weights=True)


def test_clip_by_global_norm_matches_reference_rule():
    """utils/optimization.py:233-237 (tf.clip_by_global_norm): g * clip / max(||g||, clip) over ALL gradients at once."""
    from merlot_amd import ParamStore
    from merlot_amd.optimization import AdamOptimizer
    cfg = tiny_config()
    st = ParamStore(cfg, 'cpu', seed=0)
    g = torch.Generator().manual_seed(5)
    for scale, clip in ((3.0, 1.0), (1e-4, 1.0)):
        for name in st.names():
            st.g(name).copy_(torch.randn(st.g(name).shape, generator=g) * scale)
        ref = {n: st.g(n).clone() for n in st.names()}
        norm_ref = float(np.sqrt(sum(float((v.double() ** 2).sum()) for v in ref.values())))
        opt = AdamOptimizer.__new__(AdamOptimizer)
        opt.store, opt.clip_norm, opt.frozen = st, clip, []
        norm = float(opt.clip_local_gradients())
        assert abs(norm - norm_ref) < 1e-4 * norm_ref
        factor = clip / max(norm_ref, clip)
        for n in st.names():
            assert torch.allclose(st.g(n), ref[n] * factor, rtol=1e-5, atol=1e-8), n


def test_optimizer_hyperparameter_semantics_follow_the_reference(emu):
    """utils/optimization.py defaults and overrides.  (1) clip_norm defaults to 1.0 (:57);
    (2) `freeze_scope` is the deprecated spelling of a learning_rate-0 override (:124-127): those variables are dropped
    before tf.gradients (:145-152) -- no update, no Adam slots touched, not part of the clipped global norm;
    (3) per-parameter beta_1 / beta_2 / epsilon overrides reach the update (:340-345), incl. the bias correction."""
    from merlot_amd import ParamStore
    from merlot_amd.optimization import AdamOptimizer, build_optimizer_from_config, learning_rate_scale
    cfg = tiny_config()
    st = ParamStore(cfg, 'cpu', seed=0)
    opt = build_optimizer_from_config(st, {'type': 'adam_optimizer', 'learning_rate': 1e-3, 'num_train_steps': 100,
                                           'num_warmup_steps': 10})
    assert opt.clip_norm == 1.0 and opt.eps == 1e-6 and opt.b2 == 0.98 and opt.b1 == 0.9
    with pytest.raises(ValueError, match="isn't a changable optimization parameter"):
        AdamOptimizer(st, 1e-3, 100, 10, param_overrides=[[['nothing_matches_this'], {'momentum': 0.5}]])

    over = [[['LayerNorm', 'bias'], {'weight_decay_rate': 0}], [['^lm_head/'], {'beta_2': 0.5, 'epsilon': 1e-3, 'beta_1': 0.8}]]
    opt = AdamOptimizer(st, 1e-3, 100, 10, weight_decay_rate=0.1, param_overrides=over, freeze_scope='contrastive',
                        clip_norm=2.0)
    assert not opt.single_launch
    g = torch.Generator().manual_seed(1)
    st.grad.copy_(torch.randn(st.grad.shape, generator=g) * 1e-3)
    for name, (off, n, _) in st.offsets.items():                     # arena padding carries no gradient
        st.grad[off + n:off + (n + 63) // 64 * 64] = 0
    before = st.master.clone()
    frozen_names = [n for n in st.names() if n.startswith('contrastive')]
    live = torch.cat([st.g(n).reshape(-1) for n in st.names() if not n.startswith('contrastive')])
    norm = float(opt.clip_local_gradients())
    assert abs(norm - float(live.double().norm())) < 1e-6 * norm          # frozen gradients are not in the norm
    opt.step_count = 3                                                    # warm-up: scale = 3 / 10
    grads = {n: st.g(n).clone() for n in st.names()}
    opt.step()
    scale = learning_rate_scale(3, 100, 10)
    assert scale == 0.3
    for n in frozen_names:
        assert torch.equal(st.p(n), st.view(before, n)) and float(st.view(opt.m, n).abs().sum()) == 0.0
    for n, (b1, b2, eps, wd) in {'lm_head/projection/kernel': (0.8, 0.5, 1e-3, 0.1), 'lm_head/projection/bias': (0.8, 0.5, 1e-3, 0.0),
                                 'encoder/layer00/output/kernel': (0.9, 0.98, 1e-6, 0.1),
                                 'encoder/layer00/LayerNorm_mlp_ln0/gamma': (0.9, 0.98, 1e-6, 0.0)}.items():
        p0, gr = st.view(before, n).double(), grads[n].double()
        t = 4.0
        lr = 1e-3 * scale * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        nm, nv = (1 - b1) * gr, (1 - b2) * (gr * gr + 1e-30)
        want = p0 - lr * (nm / (nv.sqrt() + eps) + wd * p0)
        assert torch.allclose(st.p(n).double(), want, rtol=2e-5, atol=1e-9), n


def test_uniform_param_override_reaches_the_single_launch_update(emu):
    """an override that matches EVERY variable (here beta_2 / epsilon for '.*') collapses to one (lr, b1, b2, eps)
    key, so the whole arena updates in one launch -- which must use the overridden values, not the constructor defaults."""
    from merlot_amd import ParamStore
    from merlot_amd.optimization import AdamOptimizer, learning_rate_scale
    cfg = tiny_config()
    st = ParamStore(cfg, 'cpu', seed=0)
    opt = AdamOptimizer(st, 1e-3, 100, 10, weight_decay_rate=0.1, clip_norm=0.0,
                        param_overrides=[[['LayerNorm', 'bias'], {'weight_decay_rate': 0}], [['.*'], {'beta_2': 0.95, 'epsilon': 1e-4}]])
    assert opt.single_launch and opt.single_key == (1e-3, 0.9, 0.95, 1e-4)
    g = torch.Generator().manual_seed(2)
    st.grad.copy_(torch.randn(st.grad.shape, generator=g) * 1e-3)
    before = st.master.clone()
    opt.step_count = 5
    grads = {n: st.g(n).clone() for n in ('encoder/layer00/output/kernel', 'encoder/layer00/output/bias')}
    opt.step()
    scale = learning_rate_scale(5, 100, 10)
    for n, wd in (('encoder/layer00/output/kernel', 0.1), ('encoder/layer00/output/bias', 0.0)):
        p0, gr = st.view(before, n).double(), grads[n].double()
        b1, b2, eps, t = 0.9, 0.95, 1e-4, 6.0
        lr = 1e-3 * scale * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        nm, nv = (1 - b1) * gr, (1 - b2) * (gr * gr + 1e-30)
        want = p0 - lr * (nm / (nv.sqrt() + eps) + wd * p0)
        assert torch.allclose(st.p(n).double(), want, rtol=2e-5, atol=1e-9), n


@pytest.mark.parametrize('name', ['unshared', 'langonly_groups', 'block_mask', 'img_mask'])
def test_config_variants_match_reference_program(emu, name):
    """the host wiring of `share_params: False` (own `langonly_encoder` weights, its own depth) and
    `langonly_num_chunks_in_group` against what the reference program computed (tests/golden/ref_shim_variants.npz)."""
    import os
    from merlot_amd import MerlotModel, ParamStore
    from test_reference_shim import VARIANTS
    fx = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_shim_variants.npz'))
    p = name + '/'
    over = dict(VARIANTS[name])
    img_mask = over.pop('_img_mask', None)
    cfg = tiny_config(**over)
    b = synth_batch(cfg, E=2, num_chunks=4, Lc=32, seed=3)
    w = mo.init_weights(cfg, seed=8, perturb=True)
    st = ParamStore(cfg, 'cpu', seed=0)
    assert st.load_tf_weights(w) == []
    assert sorted(st.export_tf_weights()) == [str(n) for n in fx[p + 'variable_names']]
    noise = {k: torch.from_numpy(fx[p + 'noise/' + k]) for k in ('gumbel', 'span_lower', 'span_upper', 'random_ids', 'option')}
    sidx = torch.from_numpy(b['shuffled_idx_img'])
    pm = MerlotModel(cfg, True, False, b['image'], b['input_ids'], mask_input=True, shuffled_idx_img=sidx, params=st, noise=noise,
                     img_mask=None if img_mask is None else torch.tensor(img_mask))
    assert np.array_equal(pm.lang_mask_info['masked_idx'].numpy(), fx[p + 'masked_idx'])
    assert np.array_equal(pm.lang_mask_info['masked_ids'].numpy(), fx[p + 'masked_ids'])
    loss = pm.mask_loss()[0] + pm.contrastive_loss()[0] + pm.temporal_loss(sidx, torch.from_numpy(b['video_src_ids']))[0]
    assert abs(float(loss) - float(fx[p + 'loss'])) < 3e-2
    assert rel_l2(torch.from_numpy(head(pm.encoder_hidden_states['lang'].detach().float().numpy().reshape(-1, 768))),
                  torch.from_numpy(fx[p + 'encoder_lang'])) < 2e-2
    loss.backward()
    gt = st.export_tf_grads()
    for k in fx.files:
        if k.startswith(p + 'grad/'):
            n = k[len(p) + 5:]
            assert rel_l2(torch.from_numpy(head(gt[n].numpy())), torch.from_numpy(fx[k])) < 0.12, n


def test_token_id_range_assertion_is_deferred_not_dropped(emu, oracle_run):
    """utils/model_utils.py:256-258 asserts 0 <= id < vocab inside the graph.  Here the flag is computed on the device (no
    host round trip in the middle of the forward), the lookup is clamped (memory-safe), and the error is raised by
    `check_token_ids()` / by the trainer after the backward is queued and before the optimizer update -- the bad batch never
    reaches the weights or a checkpoint."""
    from merlot_amd import MerlotModel, ParamStore, model_fn_builder
    from merlot_amd.config import NeatConfig
    from merlot_amd.train import Trainer
    cfg, w, b, m, loss, info = oracle_run
    st = ParamStore(cfg, 'cpu', seed=0)
    good = MerlotModel(cfg, True, False, b['image'], b['input_ids'], mask_input=True, shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']),
                       params=st, noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
    good.check_token_ids()
    assert not bool(good.token_id_flag())
    ids = b['input_ids'].clone()
    ids[0, 0, 3] = cfg['vocab_size']                      # one past the end
    bad = MerlotModel(cfg, True, False, b['image'], ids, mask_input=True, shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']),
                      params=st, noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
    assert bool(bad.token_id_flag())
    with pytest.raises(ValueError, match='out of range'):
        bad.check_token_ids()
    # the trainer: the step that embedded the bad id raises, names itself and leaves the weights untouched
    config = NeatConfig.from_dict({'model': dict(cfg), 'data': {'num_chunks': 4, 'chunk_text_len': 32},
                                   'device': {'use_tpu': False, 'output_dir': '/tmp/unused'},
                                   'optimizer': {'type': 'adam_optimizer', 'learning_rate': 1e-4, 'num_train_steps': 10, 'num_warmup_steps': 0}})
    tr = Trainer(config, 'cpu', None, seed=0)
    feats = {'images': b['image'], 'input_ids': ids, 'shuffled_idx_img': torch.from_numpy(b['shuffled_idx_img']),
             'video_src_ids': torch.from_numpy(b['video_src_ids']), 'noise': {k: torch.from_numpy(v) for k, v in b['noise'].items()}}
    before = tr.store.master.clone()
    with pytest.raises(ValueError, match='training step 0'):
        tr.step(feats)                                    # raised by the step itself, BEFORE its update is applied
    assert torch.equal(tr.store.master, before) and tr.step_idx == 0


def test_attention_log_taken_from_the_backward_equals_the_forward_values(emu, oracle_run):
    """`attention_log_in_backward` (merlot_amd/modeling.py; the Trainer's default): the four viz / lang block fractions of
    model/modeling.py:186-203 are zeros until `backward()` has run and equal the forward-time values afterwards; without a backward
    (no_grad) they are complete at construction."""
    from merlot_amd import MerlotModel, ParamStore
    cfg, w, b, m, loss, info = oracle_run
    outs = {}
    for mode in (False, True):
        c = dict(cfg, attention_log_in_backward=mode)
        st = ParamStore(c, 'cpu', seed=0)
        st.load_tf_weights({k: v.detach() for k, v in w.items()})
        pm = MerlotModel(c, True, False, b['image'], b['input_ids'], mask_input=True,
                         shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']), params=st,
                         noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
        before = {k: float(v) for k, v in pm.attention_log.items()}
        l1, _ = pm.mask_loss()
        st.zero_grad()
        l1.backward()
        outs[mode] = (before, {k: float(v) for k, v in pm.attention_log.items()})
    assert all(v == 0.0 for v in outs[True][0].values())                 # deferred: nothing before the backward
    assert outs[False][0] == outs[False][1]                               # immediate: untouched by the backward
    for k, v in outs[False][1].items():
        assert abs(outs[True][1][k] - v) < 1e-6, k
        assert abs(v - float(m.attention_log[k])) < 1e-3
    assert abs(sum(outs[True][1].values()) - 1.0) < 1e-5
    with torch.no_grad():                                                 # no backward will come: complete at construction
        c = dict(cfg, attention_log_in_backward=True)
        st = ParamStore(c, 'cpu', seed=0)
        st.load_tf_weights({k: v.detach() for k, v in w.items()})
        pm = MerlotModel(c, True, False, b['image'], b['input_ids'], mask_input=True,
                         shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']), params=st,
                         noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
        for k, v in outs[False][1].items():
            assert abs(float(pm.attention_log[k]) - v) < 1e-6


def test_training_steps_leave_no_tensors_behind(emu):
    """A step's activations must be released by reference counting when the step ends -- not by a later pass of Python's cycle collector
    (which does not see through autograd's C++ nodes and runs at unpredictable times).  Round 4 found two cycles through the graph
    (an output kept as a plain ctx attribute in L2NormFn; the attention-log callback, which closed over the model, stored in the encoder's
    node): at the bench's batch every step left ~2.5 GB behind and `bench.py --steps 20 --warmup 5` ran out of the 288 GB.  With the
    collector DISABLED the number of live tensors must not grow from step to step."""
    import gc
    from merlot_amd.config import NeatConfig
    from merlot_amd.train import Trainer, synthetic_batch
    cfg = tiny_config(hidden_dropout_prob=0.0)
    config = NeatConfig.from_dict({'model': dict(cfg), 'data': {'num_chunks': 4, 'chunk_text_len': 32},
                                   'device': {'use_tpu': False, 'output_dir': '/tmp/unused'},
                                   'optimizer': {'type': 'adam_optimizer', 'learning_rate': 1e-4, 'num_train_steps': 10, 'num_warmup_steps': 0}})
    tr = Trainer(config, 'cpu', None, seed=0)
    batch = synthetic_batch(config, 2, torch.device('cpu'), seed=1)

    def live():
        return sum(1 for o in gc.get_objects() if isinstance(o, torch.Tensor))
    tr.step(batch)
    gc.collect()
    gc.disable()
    try:
        out = tr.step(batch)
        n1 = live()
        out = tr.step(batch)
        n2 = live()
        out = tr.step(batch)
        n3 = live()
    finally:
        gc.enable()
    assert not out['loss'].requires_grad                  # values, not a graph
    assert n1 == n2 == n3, (n1, n2, n3)
