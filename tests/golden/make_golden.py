"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only (it reads
/root/reference for part (A)); the fixtures are data -- inputs and expected outputs -- never reference source.

(A) TRUE reference outputs.  The only hot-path code of rowanz/merlot that can execute here (no tensorflow) is
    downstream/sort_story/score_permutations.py's four pure-python functions; they are AST-extracted from the
    reference file at run time, executed on seeded inputs, and their outputs stored in `sort_story_ref.npz`.
    The tokenizer constants of utils/encode/encoder.py are stored alongside.
(B) Oracle pins (NOT reference outputs -- parity unpinned, see oracle/__init__.py): expected outputs of the oracle
    on the seeded config-#1 batch (`config1_expected.npz`) and integer known-answer tests for mask_inputs /
    shuffled_idx / temporal labels (`index_kat.npz`), so the oracle cannot drift silently.
"""
import ast
import itertools
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
REF = '/root/reference'


def reference_sort_story_functions():
    src = open(os.path.join(REF, 'downstream/sort_story/score_permutations.py')).read()
    tree = ast.parse(src)
    wanted = {'score_permutation', 'spearman_acc', 'absolute_distance', 'pairwise_acc'}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted], type_ignores=[])
    import scipy
    from scipy import stats  # noqa: F401
    ns = {'np': np, 'scipy': scipy}
    exec(compile(mod, 'score_permutations_extract', 'exec'), ns)
    return ns


def make_sort_story_ref():
    ns = reference_sort_story_functions()
    rng = np.random.RandomState(123)
    n_story = 6
    probs = rng.dirichlet(np.ones(3), size=(n_story, 5, 5)).astype(np.float32)       # softmax[:, 1:]-like rows
    best_perm, best_score, eqs, gtlts = [], [], [], []
    perms = list(itertools.permutations(list(range(5))))
    for s in range(n_story):
        table = {}
        for perm in perms:                                                           # score_permutations.py:58-66
            m, g = ns['score_permutation'](probs[s], xa_perm=np.arange(5), xb_perm=perm)
            table[tuple(perm)] = np.log(m).sum() + np.log(g).sum()
        ranked = sorted(table.items(), key=lambda x: -x[1])
        best_perm.append(ranked[0][0])
        best_score.append(ranked[0][1])
        m, g = ns['score_permutation'](probs[s], xa_perm=np.arange(5), xb_perm=perms[37])
        eqs.append(m)
        gtlts.append(g)
    stories = [perms[i] for i in (0, 5, 17, 60, 119, 77)]
    np.savez(os.path.join(HERE, 'sort_story_ref.npz'), probs=probs, best_perm=np.array(best_perm),
             best_score=np.array(best_score), eq_perm37=np.array(eqs), gtlt_perm37=np.array(gtlts),
             stories=np.array(stories),
             spearman=np.array([ns['spearman_acc'](list(s)) for s in stories]),
             absdist=np.array([ns['absolute_distance'](list(s)) for s in stories]),
             pairwise=np.array([ns['pairwise_acc'](list(s)) for s in stories]))
    sys.path.insert(0, REF)
    from utils.encode import encoder as enc
    e = enc.get_encoder()
    text = "the quick brown fox jumps over the lazy dog while MERLOT watches youtube videos"
    np.savez(os.path.join(HERE, 'tokenizer_ref.npz'), PADDING=enc.PADDING, MASK=enc.MASK, START=enc.START, END=enc.END,
             NEXTCAPTION_START=enc.NEXTCAPTION_START, vocab_len=len(e.encoder), text=np.array(text),
             ids=np.array(e.encode(text), dtype=np.int32))


def make_oracle_pins():
    from common import tiny_config, synth_batch
    from oracle import merlot_oracle as mo, index_oracle as ix
    cfg = tiny_config()
    w = mo.init_weights(cfg, 0)
    for t in w.values():
        t.requires_grad_(True)
    b = synth_batch(cfg)
    m = mo.MerlotOracle(cfg, w, b['image'], b['input_ids'], mask_input=True, shuffled_idx_img=b['shuffled_idx_img'],
                        noise=b['noise'])
    loss, info = m.total_loss(b['shuffled_idx_img'], b['video_src_ids'])
    loss.backward()
    flat = {f'{a}/{k}': float(v) for a, d in info.items() for k, v in d.items()}
    gnorm = {k: float(v.grad.norm()) for k, v in w.items() if v.grad is not None}
    np.savez(os.path.join(HERE, 'config1_expected.npz'), loss=float(loss),
             metric_names=np.array(sorted(flat)), metric_values=np.array([flat[k] for k in sorted(flat)]),
             masked_idx=m.lang_mask_info['masked_idx'].numpy(), masked_ids=m.lang_mask_info['masked_ids'].numpy(),
             shuffled_idx_img=b['shuffled_idx_img'],
             attention_summs=m.attention_summs().detach().numpy(),
             img_trg_h=m.img_trg_h.detach().numpy()[:, :16], lang_trg_h=m.lang_trg_h.detach().numpy()[:, :16],
             enc_viz=m.encoder_hidden_states['viz'].detach().numpy()[:, ::5, :16],
             enc_lang=m.encoder_hidden_states['lang'].detach().numpy()[:, ::16, :16],
             attention_log=np.array([float(m.attention_log[k]) for k in sorted(m.attention_log)]),
             grad_names=np.array(sorted(gnorm)), grad_norms=np.array([gnorm[k] for k in sorted(gnorm)]))
    # integer KATs
    rng = np.random.RandomState(7)
    kats = {}
    for tag, (B, L) in {'L128': (3, 128), 'L160': (2, 160), 'L512': (2, 512)}.items():
        c = dict(cfg)
        ids = rng.randint(100, 50354, size=(B, L)).astype(np.int32)
        for bb in range(B):
            for st in range(0, L, 32):
                ids[bb, st] = 2
                ids[bb, st + rng.randint(8, 32):st + 32] = 0
        nm = int(L * 0.2)
        summ = rng.gamma(2.0, 1.0, size=(B, L)).astype(np.float32)
        summ[0, 5] = summ[0, 9]                                    # exercise top_k ties
        noise = dict(gumbel=(-np.log(-np.log(rng.uniform(size=(B, L))))).astype(np.float32),
                     span_lower=rng.choice(3, size=(B, nm), p=[0.625, 0.25, 0.125]).astype(np.int32),
                     span_upper=rng.choice(3, size=(B, nm), p=[0.625, 0.25, 0.125]).astype(np.int32),
                     random_ids=rng.randint(100, 50370, size=B * L).astype(np.int32),
                     option=rng.choice(3, size=B * L, p=[0.1, 0.8, 0.1]).astype(np.int32))
        mids, midx = ix.mask_inputs(ids, summ, c, 50370, noise)
        kats.update({f'{tag}_ids': ids, f'{tag}_summ': summ, f'{tag}_masked_ids': mids, f'{tag}_masked_idx': midx})
        kats.update({f'{tag}_{k}': v for k, v in noise.items()})
    B, n = 6, 4
    ns_ = rng.randint(0, 5, size=B).astype(np.int32)
    us, up = rng.uniform(size=(B, n)).astype(np.float32), rng.uniform(size=(B, n)).astype(np.float32)
    vs = np.array([[0, 0, 1, 1], [0, 0, 0, 0], [0, 1, 2, 3], [0, 0, 0, 1], [5, 5, 5, 5], [0, 1, 1, 1]], np.int32)
    sidx = ix.shuffled_idx_img(B, n, 0.4, ns_, us, up)
    kats.update(shuf_num=ns_, shuf_us=us, shuf_up=up, shuf_out=sidx, temporal_vsrc=vs,
                temporal_labels=ix.allpairs_temporal_labels(vs, n), temporal_weights=ix.temporal_label_weights(sidx, n),
                temporal_weights_sortstory=ix.temporal_label_weights(sidx + 64, n))
    np.savez(os.path.join(HERE, 'index_kat.npz'), **kats)


if __name__ == '__main__':
    torch.manual_seed(0)
    if os.path.isdir(REF):
        make_sort_story_ref()
    make_oracle_pins()
    print('wrote', sorted(f for f in os.listdir(HERE) if f.endswith('.npz')))
