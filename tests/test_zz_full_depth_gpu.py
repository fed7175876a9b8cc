"""GPU, BASELINE config #2 at FULL depth (224^2 frames, ViT-B/16 12 layers + 12-layer text-only + 12-layer joint, 16 chunks
per example, merlot.yaml masking) -- too large for the CPU oracle, so checked through size-independent properties:
  * index work: masked_idx strictly increasing and in range, masked_ids differ from the input only at selected positions,
    special tokens (< 100) are never selected (model/modeling.py:423, 442, 473), temporal labels / shuffled idx in range;
  * batch independence: per-example outputs of a batch equal those of the sub-batch holding only those examples
    (frames / captions never interact across examples through the three encoders);
  * permutation equivariance: swapping two examples swaps their outputs;
  * losses at random initialisation sit where chance puts them (MLM ~ ln V), everything is finite, and the backward fills
    the whole gradient arena with finite numbers.
(last in file order on purpose: it is the heaviest test.)"""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run_properties(device, examples=4, layers=12):
    from merlot_amd import MerlotModel, NeatConfig, ParamStore
    from merlot_amd.train import synthetic_batch
    config = NeatConfig.from_yaml(os.path.join(ROOT, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
    cfg = config.model
    cfg.update(hidden_dropout_prob=0.0, num_hidden_layers=layers, num_vision_transformer_hidden_layers=layers,
               num_lang_transformer_hidden_layers=layers)
    nc, n, Lc = config.data['num_chunks'], cfg['num_chunks_in_group'], config.data.get('chunk_text_len', 32)
    b = synthetic_batch(config, examples, device, seed=77)
    st = ParamStore(cfg, device, seed=0)

    # ---- training graph: index work + losses + backward
    st.zero_grad()
    pm = MerlotModel(cfg, True, False, b['images'], b['input_ids'], mask_input=True, shuffled_idx_img=b['shuffled_idx_img'],
                     params=st, noise=b['noise'])
    B, L = pm.B, pm.L
    assert (B, L, pm.P) == (examples * nc // n, Lc * n, 50 * n)
    ids = b['input_ids'].reshape(B, L).cpu().long()
    midx = pm.lang_mask_info['masked_idx'].cpu().long()
    mids = pm.lang_mask_info['masked_ids'].reshape(B, L).cpu().long()
    assert midx.shape == (B, int(L * cfg['masking_rate']))
    assert bool((midx[:, 1:] > midx[:, :-1]).all()) and int(midx.min()) >= 0 and int(midx.max()) < L     # tf.sort, :473
    sel = torch.zeros((B, L), dtype=torch.bool)
    sel[torch.arange(B)[:, None], midx] = True
    assert bool((mids[~sel] == ids[~sel]).all())                       # only selected positions may change
    assert bool((ids[sel] >= 100).all())                               # specials carry -1e8 log-weight (:423, :442)
    assert bool(((mids[sel] == 1) | (mids[sel] == ids[sel]) | (mids[sel] >= 100)).all())   # MASK / keep / random id
    frac_mask = float((mids[sel] == 1).float().mean())
    assert 0.65 < frac_mask < 0.92                                     # 80 % MASK (+ chance hits), :474-487
    sidx = b['shuffled_idx_img'].cpu()
    assert bool(((sidx >= 0) & (sidx < n) | (sidx >= 16) & (sidx < 16 + n)).all())
    l1, i1 = pm.mask_loss()
    l2, i2 = pm.contrastive_loss()
    l3, i3 = pm.temporal_loss(b['shuffled_idx_img'], b['video_src_ids'])
    for v in (l1, l2, l3):
        assert math.isfinite(float(v))
    assert abs(float(l1) - math.log(cfg["vocab_size"])) < 1.5          # untrained LM head: ~ln V
    assert 0.0 < float(i2['loss_all']) < 2.0 * math.log(examples * nc) and 0.0 <= float(i1['acc']) <= 1.0
    (l1 + l2 + l3).backward()
    torch.cuda.synchronize() if str(device).startswith('cuda') else None
    g = st.grad
    assert bool(torch.isfinite(g).all())
    nz = [float(st.g(name).abs().sum()) > 0 for name in st.names() if 'key_layer' not in name]
    assert sum(nz) >= len(nz) - 2                                      # every tensor but (at most) degenerate biases got a gradient

    # ---- inference graph: batch independence and permutation equivariance of the encoders
    with torch.no_grad():
        def enc(images, ids_, sid):
            m = MerlotModel(cfg, False, False, images, ids_, mask_input=False, shuffled_idx_img=sid, params=st,
                            log_attention_probs=False)
            return m.encoder_hidden_states['viz'], m.encoder_hidden_states['lang'], m.img_trg_h
        gpe = nc // n                                                  # groups per example
        full = enc(b['images'], b['input_ids'], b['shuffled_idx_img'])
        sub = enc(b['images'][:2 * nc], b['input_ids'][:2], b['shuffled_idx_img'][:2 * nc])
        # measured: exactly 0 -- every kernel of the forward sums each output's K range in the same order whatever the batch
        # (the ring and the ping-pong GEMMs give bit-identical results, scripts/exp_p8.py), and nothing in it uses atomics
        assert _rel(sub[0], full[0][:2 * gpe]) <= 1e-6 and _rel(sub[1], full[1][:2 * gpe]) <= 1e-6
        assert _rel(sub[2], full[2][:2 * nc]) <= 1e-6
        perm = torch.arange(examples)
        perm[0], perm[1] = 1, 0
        img_p = b['images'].reshape(examples, nc, *b['images'].shape[1:])[perm].reshape(b['images'].shape)
        sid_p = b['shuffled_idx_img'].reshape(examples, nc)[perm].reshape(-1)
        sw = enc(img_p, b['input_ids'][perm], sid_p)
        gp = torch.arange(examples * gpe).reshape(examples, gpe)[perm].reshape(-1)
        assert _rel(sw[0], full[0][gp]) <= 1e-6 and _rel(sw[1], full[1][gp]) <= 1e-6
        assert _rel(full[0][:gpe], full[0][gpe:2 * gpe]) > 0.1            # and the two examples really differ
    return float(l1), float(l2), float(l3)


def test_full_depth_config2_properties():
    run_properties(torch.device('cuda', 0), examples=4, layers=12)
