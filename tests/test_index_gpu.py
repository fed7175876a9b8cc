"""GPU: the integer / index kernels are BIT-EXACT against the oracle given the same explicit noise."""
import os

import numpy as np
import pytest
import torch

from common import tiny_config
from oracle import index_oracle as ix

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _run_mask(ops, cfg, ids, summ, noise):
    from merlot_amd.modeling import masking_constants
    B, L = ids.shape
    c = masking_constants(L, cfg)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    span = c['do_spanbert']
    mids, midx = ops.mask_inputs(t(ids), t(summ) if c['use_attn'] else None, t(noise['gumbel']),
                                 t(noise['span_lower']) if span else None, t(noise['span_upper']) if span else None,
                                 t(noise['random_ids'].reshape(B, L)), t(noise['option'].reshape(B, L)), c['num_topk'],
                                 c['num_to_mask'], c['w_nontopk'], c['w_topk'], c['log_nontopk'], c['log_topk'],
                                 c['max_weight'])
    return mids.cpu().numpy(), midx.cpu().numpy()


def test_mask_inputs_golden_kats():
    from merlot_amd import ops
    k = np.load(os.path.join(G, 'index_kat.npz'))
    cfg = tiny_config()
    for tag in ['L128', 'L160', 'L512']:
        noise = {n: k[f'{tag}_{n}'] for n in ['gumbel', 'span_lower', 'span_upper', 'random_ids', 'option']}
        mids, midx = _run_mask(ops, cfg, k[f'{tag}_ids'], k[f'{tag}_summ'], noise)
        assert np.array_equal(midx, k[f'{tag}_masked_idx']), tag
        assert np.array_equal(mids, k[f'{tag}_masked_ids']), tag


@pytest.mark.parametrize("variant", ['default', 'no_spanbert', 'no_attn', 'ties'])
def test_mask_inputs_random_vs_oracle(variant):
    from merlot_amd import ops
    cfg = tiny_config()
    if variant == 'no_spanbert':
        cfg['masking_do_spanbert'] = False
    if variant == 'no_attn':
        cfg['masking_use_attn'] = False
    rng = np.random.RandomState(hash(variant) % 1000)
    for B, L in [(5, 128), (3, 160), (2, 96)]:
        ids = rng.randint(100, 50354, size=(B, L)).astype(np.int32)
        ids[:, ::32] = 2
        ids[rng.uniform(size=(B, L)) < 0.3] = 0
        nm = int(L * 0.2)
        summ = rng.gamma(2.0, 1.0, size=(B, L)).astype(np.float32)
        if variant == 'ties':
            summ = np.round(summ)                                    # many exact ties -> index tie-breaking matters
        noise = dict(gumbel=(-np.log(-np.log(rng.uniform(size=(B, L))))).astype(np.float32),
                     span_lower=rng.choice(3, size=(B, nm)).astype(np.int32),
                     span_upper=rng.choice(3, size=(B, nm)).astype(np.int32),
                     random_ids=rng.randint(100, 50370, size=B * L).astype(np.int32),
                     option=rng.choice(3, size=B * L, p=[0.1, 0.8, 0.1]).astype(np.int32))
        ref_ids, ref_idx = ix.mask_inputs(ids, summ, cfg, 50370, noise)
        mids, midx = _run_mask(ops, cfg, ids, summ, noise)
        assert np.array_equal(midx, ref_idx) and np.array_equal(mids, ref_ids)


def test_temporal_labels_and_shuffled_idx():
    from merlot_amd import ops
    k = np.load(os.path.join(G, 'index_kat.npz'))
    B, n = k['temporal_vsrc'].shape
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = ops.shuffled_idx(t(k['shuf_num']), t(k['shuf_us']), t(k['shuf_up']), B, n, 16).cpu().numpy()
    assert np.array_equal(out, k['shuf_out'])
    labels, w = ops.temporal_labels(t(k['temporal_vsrc']), t(k['shuf_out']), B, n)
    assert np.array_equal(labels.cpu().numpy(), k['temporal_labels'])
    assert np.array_equal(w.cpu().numpy(), k['temporal_weights'])
    _, w2 = ops.temporal_labels(t(k['temporal_vsrc']), t(k['shuf_out'] + 64), B, n)
    assert np.array_equal(w2.cpu().numpy(), k['temporal_weights_sortstory'])
    rng = np.random.RandomState(3)
    for B, n in [(7, 5), (3, 16), (11, 4)]:
        ns = rng.randint(0, n + 1, size=B).astype(np.int32)
        us, up = rng.uniform(size=(B, n)).astype(np.float32), rng.uniform(size=(B, n)).astype(np.float32)
        vs = rng.randint(0, 3, size=(B, n)).astype(np.int32)
        ref = ix.shuffled_idx_img(B, n, 0.5, ns, us, up)
        got = ops.shuffled_idx(t(ns), t(us), t(up), B, n, 16).cpu().numpy()
        assert np.array_equal(got, ref)
        labels, w = ops.temporal_labels(t(vs), t(ref), B, n)
        assert np.array_equal(labels.cpu().numpy(), ix.allpairs_temporal_labels(vs, n))
        assert np.array_equal(w.cpu().numpy(), ix.temporal_label_weights(ref, n))


def test_shuffled_idx_kernel_matches_the_reference_run():
    """`merlot_shuffled_idx` against what model/dataloader.py:224-257 itself produced under the TF shim
    (tests/golden/ref_shim_shuffle.npz, tests/golden/make_shuffle_golden.py) -- not only against the restatement."""
    from merlot_amd import ops
    fx = np.load(os.path.join(G, 'ref_shim_shuffle.npz'))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    checked = 0
    for i in range(int(fx['shuffle/count'])):
        pre = f'shuffle/{i}/'
        if pre + 'num_shuffle' not in fx.files:
            continue                                   # image_shuffle_prob < 1e-6: host-side arange, no kernel (:233-235)
        n, B = int(fx[pre + 'n']), int(fx[pre + 'B'])
        got = ops.shuffled_idx(t(fx[pre + 'num_shuffle']), t(fx[pre + 'u_select']), t(fx[pre + 'u_perm']), B, n, 16)
        assert np.array_equal(got.cpu().numpy(), fx[pre + 'out']), i
        checked += 1
    assert checked >= 6
