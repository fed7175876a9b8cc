"""Integer / index work of the MERLOT hot path, restated in numpy (CPU oracle).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it follows.  All randomness is an explicit input so integer
outputs are bit-comparable with the HIP path.

TF semantics honoured here:
  * ``tf.math.top_k`` / ``tf.nn.top_k``: descending, ties -> lower index first.
  * ``tf.argmax``: first maximal index.
  * ``tf.argsort``: ascending (stable), ``tf.sort``: ascending.
"""
import itertools

import numpy as np

MASK = 1  # utils/encode/encoder.py:17


def top_k(x, k):
    """tf.math.top_k on the last axis: values descending, ties -> lower index."""
    x = np.asarray(x)
    order = np.argsort(-x, axis=-1, kind='stable')
    idx = order[..., :k]
    return np.take_along_axis(x, idx, -1), idx.astype(np.int32)


def masking_constants(L, config):
    """model/modeling.py:390-419 -- python-double arithmetic then float32 casts as TF does."""
    topk_perc = config.get('masking_use_topk_from_attn_perc', 0.20)
    choose_topk_prob = config.get('masking_choose_topk_prob', 0.5)
    masking_rate = config.get('masking_rate', 0.2)
    num_topk = int(L * topk_perc)
    num_to_mask = int(L * masking_rate)
    nontopk_val = 0.01
    topk_val = nontopk_val * choose_topk_prob * (1.0 - topk_perc) / (topk_perc * (1.0 - choose_topk_prob))
    return dict(num_topk=num_topk, num_to_mask=num_to_mask,
                nontopk_val=np.float32(nontopk_val),
                topk_minus_nontopk=np.float32(topk_val - nontopk_val),
                do_spanbert=config.get('masking_do_spanbert', True),
                use_attn=config.get('masking_use_attn', True))


def mask_inputs(input_ids_2d, attention_summs, config, vocab_size, noise):
    """model/modeling.py:381-489 (+ utils/model_utils.py:640-649).

    input_ids_2d [B, L] int32; attention_summs [B, L] float32 (already reshaped,
    modeling.py:428-431) or None if masking_use_attn is False.
    noise: dict with
      'gumbel'      [B, L] float32  = -log(-log(U))          (model_utils.py:647)
      'span_lower'  [B, nm] int32 in {0,1,2}                 (modeling.py:449-451)
      'span_upper'  [B, nm] int32 in {0,1,2}                 (modeling.py:452-454)
      'random_ids'  [B*L] int32 in [100, vocab)              (modeling.py:477)
      'option'      [B*L] int32 in {0,1,2}                   (modeling.py:479-481)
    returns masked_ids [B, L] int32, masked_idx [B, nm] int32 (sorted ascending).
    """
    ids = np.asarray(input_ids_2d, dtype=np.int32)
    B, L = ids.shape
    c = masking_constants(L, config)
    nm = c['num_to_mask']
    sentinel = np.arange(L, dtype=np.int32)
    is_special = (ids < 100).astype(np.float32)                                   # :423

    if c['use_attn']:
        summ = np.asarray(attention_summs, dtype=np.float32) * (np.float32(1.0) - is_special)   # :433
        _, top_inds = top_k(summ, c['num_topk'])                                  # :435
        is_important = np.zeros((B, L), dtype=bool)
        np.put_along_axis(is_important, top_inds.astype(np.int64), True, axis=1)  # :436
        mask_weight = is_important.astype(np.float32) * c['topk_minus_nontopk'] + c['nontopk_val']  # :437
    else:
        mask_weight = np.ones((B, L), dtype=np.float32)                           # :439

    log_mask = np.log(mask_weight).astype(np.float32) - np.float32(1e8) * is_special   # :442
    scores = (log_mask + np.asarray(noise['gumbel'], dtype=np.float32)).astype(np.float32)
    _, idx = top_k(scores, nm)                                                    # model_utils.py:648
    idx = idx[:, ::-1]                                                            # :445

    if c['do_spanbert']:
        lo = np.asarray(noise['span_lower'], dtype=np.int32)
        up = np.asarray(noise['span_upper'], dtype=np.int32)
        span_start = idx - lo                                                     # :457
        span_end = idx + up                                                       # :458
        does_match = (sentinel[None, None] >= span_start[..., None]) & \
                     (sentinel[None, None] <= span_end[..., None])                # :461-464  [B, nm, L]
        which_match = np.argmax(does_match.astype(np.float32), 1).astype(np.float32)   # :465  first max
        which_match = which_match * (np.float32(1.0) - is_special)                # :466
        which_match = which_match + (np.float32(0.5) * mask_weight) / np.max(mask_weight)   # :468
        which_match = which_match.astype(np.float32)
        _, mask_idx = top_k(which_match, nm)                                      # :469
    else:
        mask_idx = idx                                                            # :471

    mask_idx = np.sort(mask_idx, 1).astype(np.int32)                              # :473
    all_options = np.stack([ids.reshape(-1),
                            np.full([B * L], MASK, dtype=np.int32),
                            np.asarray(noise['random_ids'], dtype=np.int32).reshape(-1)], 1)   # :474-478
    option = np.asarray(noise['option'], dtype=np.int32).reshape(-1)              # :479-481
    do_mask = np.zeros((B, L), dtype=bool)
    np.put_along_axis(do_mask, mask_idx.astype(np.int64), True, axis=1)           # :482-483
    option = option * do_mask.reshape(-1).astype(np.int32)                        # :484-485
    masked_ids = all_options[np.arange(B * L), option].reshape(B, L)              # :486
    return masked_ids.astype(np.int32), mask_idx


def video_src_ids(is_eoc):
    """model/dataloader.py:121-125.  is_eoc [num_chunks] bool (last forced True upstream)."""
    is_eoc = np.asarray(is_eoc).astype(np.int32)
    delta = np.concatenate([[0], is_eoc[:-1]]).astype(np.int32)
    return np.cumsum(delta).astype(np.int32)


def shuffled_idx_img(B, n, shuffle_prob, num_shuffle_img, u_select, u_perm, shuffle_offset=16):
    """model/dataloader.py:224-257.

    num_shuffle_img [B] int32: the categorical draw over
        [1-p, 1e-6, p/(n-1), ...] (dataloader.py:241-247) passed in explicitly;
    u_select, u_perm [B, n] float: the two tf.random_uniform draws (:248, :252).
    """
    if shuffle_prob < 1e-6:
        return np.tile(np.arange(n, dtype=np.int32)[None], [B, 1]).reshape(-1)    # :233-235
    num_shuffle_img = np.asarray(num_shuffle_img, dtype=np.int32)
    do_shuffle = np.argsort(np.asarray(u_select), 1, kind='stable') < num_shuffle_img[:, None]    # :248-249
    shuffled = np.where(do_shuffle,
                        shuffle_offset + np.argsort(np.asarray(u_perm), 1, kind='stable'),
                        np.tile(np.arange(n)[None], [B, 1]))                      # :250-254
    return shuffled.reshape(-1).astype(np.int32)


def num_shuffle_probs(n, shuffle_prob):
    """model/dataloader.py:241-242."""
    return [1.0 - shuffle_prob, 1e-6] + [shuffle_prob / (n - 1) for _ in range(n - 1)]


def allpairs_temporal_labels(video_src_ids_2d, n):
    """model/modeling.py:598-620.  video_src_ids_2d [B, n] -> labels [B*n*n] int32."""
    v = np.asarray(video_src_ids_2d, dtype=np.int32).reshape(-1, n)
    xa = np.tile(np.arange(n)[:, None], [1, n])
    xb = np.tile(np.arange(n)[None], [n, 1])
    base = (xa == xb).astype(np.int32) + 2 * (xa < xb).astype(np.int32) + 3 * (xa > xb).astype(np.int32)
    same = v[:, None] == v[:, :, None]                                            # :611
    labels = np.where(same, base[None], 0)
    return labels.reshape(-1).astype(np.int32)


def temporal_label_weights(shuffled_idx, n):
    """model/modeling.py:635, 649-652."""
    easy = (np.asarray(shuffled_idx).reshape(-1, n) < 64)
    is_easy = easy[:, :, None] & easy[:, None]
    w = (~is_easy).astype(np.float32) * np.float32(0.99) + np.float32(0.01)
    return w.reshape(-1).astype(np.float32)


def sort_story_shuffled_idx(u, num_chunks=5):
    """downstream/sort_story/get_zero_shot_logits.py:55-56 with the uniforms passed in."""
    u = np.asarray(u).reshape(-1, num_chunks)
    return (np.argsort(u, 1, kind='stable') + 64).astype(np.int32)


# ---- story-ordering permutation scoring (downstream/sort_story/score_permutations.py) ----

def score_permutation(log_probs_out_resh, xa_perm, xb_perm):
    """score_permutations.py:14-27 restated (note: despite the name the input holds PROBS)."""
    n = len(xa_perm)
    eq = np.ones([n, n])
    gtlt = np.ones([n, n])
    for i, ti in enumerate(xa_perm):
        for j, tj in enumerate(xb_perm):
            if ti == tj:
                eq[i, j] = log_probs_out_resh[i, j, 0]
            elif ti < tj:
                gtlt[i, j] = log_probs_out_resh[i, j, 1]
            else:
                gtlt[i, j] = log_probs_out_resh[i, j, 2]
    return eq, gtlt


def best_permutation(lv_scores):
    """score_permutations.py:58-68: brute-force all n! orders, stable sort by -score."""
    n = lv_scores.shape[0]
    perm_to_prob = {}
    for perm in itertools.permutations(list(range(n))):
        m, g = score_permutation(lv_scores, np.arange(n), perm)
        perm_to_prob[tuple(perm)] = np.log(m).sum() + np.log(g).sum()
    ranked = sorted(perm_to_prob.items(), key=lambda x: -x[1])
    return ranked[0][0], ranked[0][1]


def pairwise_acc(story):
    """score_permutations.py:37-44."""
    correct = 0
    total = len(story) * (len(story) - 1) // 2
    for a in range(len(story)):
        for b in range(a + 1, len(story)):
            if story[a] < story[b]:
                correct += 1
    return correct / total


def absolute_distance(story):
    """score_permutations.py:33-34."""
    return np.mean(np.abs(np.array(story) - np.arange(len(story))))


def spearman_acc(story):
    """score_permutations.py:30-31."""
    from scipy import stats
    return stats.spearmanr(story, list(range(len(story))))[0]
