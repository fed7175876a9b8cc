"""GPU: the HIP path DIRECTLY against the fixtures produced by the reference's own program (run under
oracle/tf_shim.py in the build container, tests/golden/make_reference_golden.py) -- no oracle in the comparison.
Tolerances as everywhere (bf16 compute vs an fp32 reference): hidden states rel-L2 <= 2e-2, scalar
losses <= 1e-2 abs (summed loss 2e-2), gradients per sampled tensor within the per-class rel-L2 bounds of
tests/test_grad_classes_gpu.py (3e-2; dense kernels 4e-2; the contrastive head at 8 segments 6e-2 -- rounds 1-4 asserted 0.12 / 0.2), integer
outputs exact; optimizer: bf16 states exact up to one bf16 ulp on <= 0.1 % of elements, parameters rtol 1e-5."""
import os

import numpy as np
import pytest
import torch

from common import tiny_config, synth_batch, rel_l2, head
from grad_parity import tensor_class
from oracle import merlot_oracle as mo           # weight generator only (init_weights is seeded and shared)
from oracle import optimizer_oracle as oo

pytestmark = pytest.mark.gpu
GRAD_REL = {'bias': 3e-2, 'ln': 3e-2, 'pos': 3e-2, 'emb': 3e-2, 'kernel': 4e-2}


def grad_bound(name):
    return 6e-2 if name.startswith('contrastive/') else GRAD_REL[tensor_class(name)]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def test_hip_model_matches_reference_program_config1():
    from merlot_amd import MerlotModel, ParamStore
    fx = _load('ref_shim_config1.npz')
    cfg = tiny_config()
    w = mo.init_weights(cfg, 0)
    assert abs(sum(float(v.double().sum()) for v in w.values()) - float(fx['weights_checksum'])) < 1e-6
    st = ParamStore(cfg, 'cuda', seed=0)
    st.load_tf_weights(w)
    sidx = torch.from_numpy(fx['shuffled_idx_img']).cuda()
    noise = {k: torch.from_numpy(fx['noise/' + k]) for k in ('gumbel', 'span_lower', 'span_upper', 'random_ids',
                                                               'option')}
    st.zero_grad()
    pm = MerlotModel(cfg, True, False, torch.from_numpy(fx['image']).cuda(),
                     torch.from_numpy(fx['input_ids']).long().cuda(), mask_input=True, shuffled_idx_img=sidx,
                     params=st, noise=noise)
    assert np.array_equal(pm.lang_mask_info['masked_ids'].cpu().numpy(), fx['out/masked_ids'])
    assert np.array_equal(pm.lang_mask_info['masked_idx'].cpu().numpy(), fx['out/masked_idx'])
    assert rel_l2(pm.lang_transformer_info['attention_summs'].reshape(pm.B, pm.L),
                  torch.from_numpy(fx['out/attention_summs'])) < 1e-2
    assert rel_l2(pm.encoder_hidden_states['viz'], torch.from_numpy(fx['out/encoder_viz'])) < 2e-2
    assert rel_l2(pm.encoder_hidden_states['lang'], torch.from_numpy(fx['out/encoder_lang'])) < 2e-2
    assert rel_l2(pm.img_trg_h, torch.from_numpy(fx['out/img_trg_h'])) < 2e-2
    assert rel_l2(pm.lang_trg_h, torch.from_numpy(fx['out/lang_trg_h'])) < 2e-2
    l1, i1 = pm.mask_loss()
    l2, i2 = pm.contrastive_loss()
    l3, i3 = pm.temporal_loss(sidx, torch.from_numpy(fx['video_src_ids']).cuda())
    assert abs(float(l1) - float(fx['out/lang/loss'])) < 1e-2
    assert abs(float(i1['acc']) - float(fx['out/lang/acc'])) < 1e-6
    for k in ('lang_to_viz', 'viz_to_lang', 'loss_all'):
        assert abs(float(i2[k]) - float(fx['out/contr/' + k])) < 1e-2, k
    for k in ('lang_viz_loss', 'viz_viz_loss', 'loss'):
        assert abs(float(i3[k]) - float(fx['out/temporal/' + k])) < 1e-2, k
    for k in ('viz2viz', 'viz2lang', 'lang2viz', 'lang2lang'):
        assert abs(float(pm.attention_log['encoder/' + k]) - float(fx['out/attention_log/encoder/' + k])) < 2e-3
    assert abs(float(l1 + l2 + l3) - float(fx['out/loss'])) < 2e-2
    (l1 + l2 + l3).backward()
    torch.cuda.synchronize()
    gt = st.export_tf_grads()
    norms = dict(zip([str(n) for n in fx['grad_names']], fx['grad_norms']))
    rels = []
    for n, ref_norm in norms.items():
        if n.endswith('key_layer/bias'):
            continue
        rels.append(abs(float(gt[n].double().norm()) - ref_norm) / ref_norm)
    assert np.median(rels) < 1e-2 and max(rels) < 5e-2, (np.median(rels), max(rels))
    for k in fx.files:
        if k.startswith('grad/'):
            n = k[5:]
            r = rel_l2(torch.from_numpy(head(gt[n].float().cpu().numpy())), torch.from_numpy(fx[k]))
            assert r < grad_bound(n), (n, r)


def test_hip_sort_story_matches_reference_model_fn():
    from merlot_amd import MerlotModel, ParamStore
    from oracle import index_oracle as ix
    fx = _load('ref_shim_sort_story.npz')
    cfg = tiny_config(num_chunks_in_group=5)
    bs, n, dup = 2, 5, 2
    w = mo.init_weights(cfg, seed=int(fx['weights_seed']), perturb=True)
    b = synth_batch(cfg, E=bs, num_chunks=n, Lc=32, seed=int(fx['batch_seed']))
    H, W = cfg['image_size']
    with torch.no_grad():
        st = ParamStore(cfg, 'cuda', seed=0)
        st.load_tf_weights(w)
        image = b['image'].float().reshape(bs, n, H, W, 3)
        images = image.repeat(dup, 1, 1, 1, 1).reshape(bs * dup * n, H, W, 3)
        sents = b['input_ids'].repeat(dup, 1, 1)
        sidx = ix.sort_story_shuffled_idx(fx['u_shuffle'], n)
        pm = MerlotModel(cfg, False, False, images.cuda(), sents.cuda(), mask_input=False,
                         shuffled_idx_img=torch.from_numpy(sidx.reshape(-1)).cuda(), params=st,
                         log_attention_probs=False)
        h_lang, h_viz = pm.pooled_segments()
        for name, xa, xb in (('lang_viz', h_lang, h_viz), ('viz_viz', h_viz, h_viz)):
            logits = pm.allpairs_temporal_logits(xa, xb, f'{name}_temporal')
            probs = torch.softmax(logits, -1)[:, 1:].reshape(bs, dup, n, n, 3).mean(1).cpu().numpy()
            assert float(np.abs(probs - fx[f'{name}_probs']).max()) < 2e-2, name
            for s in range(bs):
                assert ix.best_permutation(probs[s])[0] == ix.best_permutation(fx[f'{name}_probs'][s])[0]


def test_hip_sort_story_product_model_fn():
    """the same through the product's own zero-shot model_fn and permutation search (merlot_amd/sort_story.py)."""
    from merlot_amd import ParamStore, sort_story as ss
    fx = _load('ref_shim_sort_story.npz')
    cfg = tiny_config(num_chunks_in_group=5)
    bs, n = 2, 5
    w = mo.init_weights(cfg, seed=int(fx['weights_seed']), perturb=True)
    b = synth_batch(cfg, E=bs, num_chunks=n, Lc=32, seed=int(fx['batch_seed']))
    st = ParamStore(cfg, 'cuda', seed=0)
    st.load_tf_weights(w)
    H, W = cfg['image_size']
    feats = {'images': b['image'].reshape(bs, n, H, W, 3).cuda(), 'sentences': b['input_ids'].cuda()}
    out = ss.model_fn_builder(cfg)(feats, None, 'infer', {'store': st, 'batch_size': bs, 'u_shuffle': fx['u_shuffle']})
    for name in ('lang_viz', 'viz_viz'):
        probs = out[f'{name}_probs'].cpu().numpy()
        assert float(np.abs(probs - fx[f'{name}_probs']).max()) < 2e-2, name
        for s in range(bs):
            assert ss.best_permutation(probs[s])[0] == ss.best_permutation(fx[f'{name}_probs'][s])[0]


def test_hip_inference_2d_ids_matches_reference_program():
    """2-D input_ids, is_training=False, no masking, shuffled_idx_img=None, ragged captions (one without padding, one that
    is START + padding) -- the way the downstream callers construct MerlotModel."""
    from merlot_amd import MerlotModel, ParamStore
    fx = _load('ref_shim_inference2d.npz')
    cfg = tiny_config()
    w = mo.init_weights(cfg, seed=int(fx['weights_seed']), perturb=True)
    with torch.no_grad():
        st = ParamStore(cfg, 'cuda', seed=0)
        st.load_tf_weights(w)
        pm = MerlotModel(cfg, False, False, torch.from_numpy(fx['image']).cuda(),
                         torch.from_numpy(fx['input_ids']).long().cuda(), mask_input=False, shuffled_idx_img=None,
                         params=st)
    assert (pm.B, pm.L, pm.P, pm.num_chunks) == (3, 32, 5, 1)
    assert rel_l2(pm.encoder_hidden_states['viz'], torch.from_numpy(fx['encoder_viz'])) < 2e-2
    # rows of the START+padding caption: compare on valid tokens (padded positions carry no contract downstream) and
    # on everything (padded query rows attend uniformly in both programs)
    assert rel_l2(pm.encoder_hidden_states['lang'], torch.from_numpy(fx['encoder_lang'])) < 2e-2
    for k, v in zip(fx['attention_log_keys'], fx['attention_log']):
        assert abs(float(pm.attention_log[str(k)]) - float(v)) < 2e-3, k


def test_hip_adamw_matches_reference_optimizer():
    """merlot_adamw_step vs what utils/optimization.py's AdamOptimizer wrote (two consecutive steps, bf16 m,
    sign-encoded v, decay on kernels only)."""
    from merlot_amd import ops
    from merlot_amd.optimization import learning_rate_scale
    import math
    fx = _load('ref_shim_optimizer.npz')
    lr0, nts, nws = float(fx['learning_rate']), int(fx['num_train_steps']), int(fx['num_warmup_steps'])
    b1, b2, eps = 0.9, float(fx['beta_2']), float(fx['epsilon'])
    for i in (0, 1):
        gs = int(fx[f's{i}/global_step'])
        t = gs + 1.0
        lr = lr0 * learning_rate_scale(gs, nts, nws) * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        for n in sorted({k.split('/', 2)[2] for k in fx.files if k.startswith(f's{i}/param/')}):
            decay = 0.0 if any(s in n for s in ('LayerNorm', 'layer_norm', 'GroupNorm', 'bias')) else 0.1
            p = torch.from_numpy(fx[f's{i}/param/{n}'].reshape(-1).copy()).cuda()
            g = torch.from_numpy(fx[f's{i}/grad/{n}'].reshape(-1).copy()).cuda()
            m = torch.from_numpy(fx[f's{i}/adam_m/{n}'].reshape(-1).copy()).to(torch.bfloat16).cuda()
            v = torch.from_numpy(fx[f's{i}/adam_v/{n}'].reshape(-1).copy()).to(torch.bfloat16).cuda()
            ops.adamw_step(p, g, m, v, lr, b1, b2, eps, decay)
            torch.cuda.synchronize()
            rm, rv = fx[f's{i}/new_adam_m/{n}'].reshape(-1), fx[f's{i}/new_adam_v/{n}'].reshape(-1)
            gm, gv = m.float().cpu().numpy(), v.float().cpu().numpy()
            for got, ref in ((gm, rm), (gv, rv)):
                assert (got != ref).mean() <= 1e-3, (n, (got != ref).mean())
            # never more than one bf16 ulp (v compared decoded: a sign flip at an exact tie is 2^-8 relative)
            assert np.allclose(gm, rm, rtol=2.0 ** -7, atol=1e-30)
            assert np.allclose(oo.decode_v(gv), oo.decode_v(rv), rtol=2.0 ** -7, atol=1e-38)
            assert np.allclose(p.cpu().numpy(), fx[f's{i}/new_param/{n}'].reshape(-1), rtol=1e-5, atol=1e-8), n


@pytest.mark.parametrize('name', ['unshared', 'langonly_groups', 'block_mask'])
def test_hip_config_variants_match_reference_program(name):
    """`share_params: False` (separate, shallower `langonly_encoder`) and `langonly_num_chunks_in_group` on the HIP path."""
    from merlot_amd import MerlotModel, ParamStore
    from test_reference_shim import VARIANTS
    fx = _load('ref_shim_variants.npz')
    p = name + '/'
    over = dict(VARIANTS[name])
    over.pop('_img_mask', None)
    cfg = tiny_config(**over)
    b = synth_batch(cfg, E=2, num_chunks=4, Lc=32, seed=3)
    w = mo.init_weights(cfg, seed=8, perturb=True)
    st = ParamStore(cfg, 'cuda', seed=0)
    assert st.load_tf_weights(w) == []
    noise = {k: torch.from_numpy(fx[p + 'noise/' + k]) for k in ('gumbel', 'span_lower', 'span_upper', 'random_ids', 'option')}
    sidx = torch.from_numpy(b['shuffled_idx_img']).cuda()
    st.zero_grad()
    pm = MerlotModel(cfg, True, False, b['image'].cuda(), b['input_ids'].cuda(), mask_input=True, shuffled_idx_img=sidx,
                     params=st, noise=noise)
    if np.array_equal(pm.lang_mask_info['masked_idx'].cpu().numpy(), fx[p + 'masked_idx']):   # bf16 top-k near-ties aside
        assert np.array_equal(pm.lang_mask_info['masked_ids'].cpu().numpy(), fx[p + 'masked_ids'])
        loss = pm.mask_loss()[0] + pm.contrastive_loss()[0] + pm.temporal_loss(sidx, torch.from_numpy(b['video_src_ids']).cuda())[0]
        assert abs(float(loss) - float(fx[p + 'loss'])) < 3e-2
        assert rel_l2(torch.from_numpy(head(pm.encoder_hidden_states['lang'].detach().float().cpu().numpy().reshape(-1, 768))),
                      torch.from_numpy(fx[p + 'encoder_lang'])) < 2e-2
        loss.backward()
        gt = st.export_tf_grads()
        for k in fx.files:
            if k.startswith(p + 'grad/'):
                n = k[len(p) + 5:]
                assert rel_l2(torch.from_numpy(head(gt[n].float().cpu().numpy())), torch.from_numpy(fx[k])) < grad_bound(n), n
    else:
        pytest.fail('masked_idx differs from the reference run (attention_summs tie?)')
