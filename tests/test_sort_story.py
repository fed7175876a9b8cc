"""CPU: the product's story-ordering module (merlot_amd/sort_story.py) against TRUE reference outputs -- the scoring
functions of downstream/sort_story/score_permutations.py executed on seeded inputs (tests/golden/sort_story_ref.npz) and
the reference's own zero-shot `model_fn` run under the shim (tests/golden/ref_shim_sort_story.npz)."""
import itertools
import os

import numpy as np
import torch

from common import tiny_config, synth_batch
from oracle import merlot_oracle as mo

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_scoring_and_metrics_match_reference_outputs():
    from merlot_amd import sort_story as ss
    ref = np.load(os.path.join(G, 'sort_story_ref.npz'))
    perms = list(itertools.permutations(list(range(5))))
    for s in range(ref['probs'].shape[0]):
        eq, gtlt = ss.score_permutation(ref['probs'][s], np.arange(5), perms[37])
        assert np.array_equal(eq, ref['eq_perm37'][s]) and np.array_equal(gtlt, ref['gtlt_perm37'][s])
        best, score = ss.best_permutation(ref['probs'][s])
        assert tuple(best) == tuple(ref['best_perm'][s]) and score == ref['best_score'][s]      # bit-exact
    for i, st in enumerate(ref['stories']):
        assert ss.pairwise_acc(list(st)) == ref['pairwise'][i]
        assert ss.absolute_distance(list(st)) == ref['absdist'][i]
        assert np.isclose(ss.spearman_acc(list(st)), ref['spearman'][i], rtol=0, atol=0, equal_nan=True)
    m = ss.story_metrics([tuple(s) for s in ref['stories']])
    assert m['pairwise'] == float(np.mean(ref['pairwise'])) and m['absolute_distance'] == float(np.mean(ref['absdist']))
    # ties resolve to the first permutation in itertools order (the reference's stable sort)
    flat = np.full((5, 5, 3), 1.0 / 3.0)
    assert ss.best_permutation(flat)[0] == perms[0]


def test_zero_shot_model_fn_matches_reference_model_fn(emu):
    """get_zero_shot_logits.py's model_fn (tile x2, argsort(u) + 64, temporal logits -> softmax[:, 1:] -> mean over the
    duplicates with the reference's [batch, dup] reshape) on the emulated host path vs the shim-executed reference."""
    from merlot_amd import ParamStore, sort_story as ss
    fx = np.load(os.path.join(G, 'ref_shim_sort_story.npz'))
    cfg = tiny_config(num_chunks_in_group=5)
    bs, n = 2, 5
    w = mo.init_weights(cfg, seed=int(fx['weights_seed']), perturb=True)
    b = synth_batch(cfg, E=bs, num_chunks=n, Lc=32, seed=int(fx['batch_seed']))
    st = ParamStore(cfg, 'cpu', seed=0)
    st.load_tf_weights(w)
    H, W = cfg['image_size']
    feats = {'images': b['image'].reshape(bs, n, H, W, 3), 'sentences': b['input_ids'], 'story_id': torch.arange(bs)}
    out = ss.model_fn_builder(cfg)(feats, None, 'infer', {'store': st, 'batch_size': bs, 'u_shuffle': fx['u_shuffle']})
    assert 'images' not in out and torch.equal(out['story_id'], feats['story_id'])
    for name in ('lang_viz', 'viz_viz'):
        probs = out[f'{name}_probs'].numpy()
        assert probs.shape == (bs, n, n, 3)
        assert float(np.abs(probs - fx[f'{name}_probs']).max()) < 2e-2, name
        for s in range(bs):
            assert ss.best_permutation(probs[s])[0] == ss.best_permutation(fx[f'{name}_probs'][s])[0]
