"""CPU: the host logic of merlot_amd (MerlotModel mirror, autograd wiring, parameter arena) with the HIP ops
swapped for their torch emulation (tests/emu_ops.py), checked against the oracle on BASELINE config #1.
This isolates host-side bugs from kernel bugs; the kernels themselves are checked on the GPU (-m gpu)."""
import numpy as np
import pytest
import torch

from common import tiny_config, synth_batch, rel_l2
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
    with pytest.raises(NotImplementedError):
        MerlotModel(tiny_config(resnet_layers=[3, 4, 9]), True, False, b['image'], b['input_ids'], params=st)
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
