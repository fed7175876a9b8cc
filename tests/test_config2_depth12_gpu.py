"""GPU, BASELINE config #2 at FULL depth against the ORACLE (VERDICT r1 weak #1e): merlot.yaml 4-segment groups, 224^2
frames, ViT-B/16 12 layers + 12-layer text-only pass + 12-layer joint encoder, 16 chunks per example (text-only
sequence 512, joint 4 groups x 328), one example -- the oracle (fp32, torch-CPU) does forward + backward of that in a
few seconds, so the full-depth path is compared value for value, not only through properties
(tests/test_zz_full_depth_gpu.py keeps the property checks at batch 4).

Tolerances (bf16 policy on the GPU vs the fp32 oracle): masked ids / idx exact, attention_summs <= 1e-2,
hidden states and contrastive targets rel-L2 <= 2e-2, losses <= 1e-2 abs.  Gradients (round 6): measured per tensor at this depth
against the fp32 oracle AND the torch emulation of the same bf16 policy (profiles/r06_c_grad_depth12.txt); the bounds are stated where they are
asserted below: rel-L2 <= 6.5e-2 per tensor (0.12 / 0.2 until round 5), class medians <= 2e-2, norm ratio within 8e-3 of the emulation's.
Reference: model/modeling.py:47-203, utils/transformer.py:141-247, utils/vision_transformer.py:173-274."""
import os

import pytest
import torch

from common import synth_batch
from test_model_gpu import _run_both, _check

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def config2():
    from merlot_amd import NeatConfig
    config = NeatConfig.from_yaml(os.path.join(ROOT, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
    cfg = dict(config.model)
    cfg['hidden_dropout_prob'] = 0.0                     # TF's dropout stream is not reproducible: parity runs use p = 0
    assert (cfg['num_hidden_layers'], cfg['num_vision_transformer_hidden_layers'],
            cfg['num_lang_transformer_hidden_layers']) == (12, 12, 12) and cfg['image_size'] == [224, 224]
    assert config.data['num_chunks'] == 16 and cfg['num_chunks_in_group'] == 4
    return cfg


def test_config2_full_depth_matches_oracle_forward_backward():
    cfg = config2()
    b = synth_batch(cfg, E=1, num_chunks=16, seed=11, two_videos=True)
    w, m, loss, info, st, pm = _run_both(cfg, b, with_grads=True)
    assert (pm.B, pm.P, pm.L) == (4, 200, 128)
    total = _check(cfg, b, w, m, info, st, pm, with_grads=True, grad_tol=REL_D12, contr_tol=REL_D12, median_tol=MEDIAN_D12)
    assert abs(total - float(loss)) < 2e-2


# Measured at this problem (profiles/r06_c_grad_depth12.txt, 411 tensors): after 36 layers a bf16 rounding that falls the other way in ONE element is amplified
# like any other perturbation, so the emulation -- which rounds at the same POINTS but sums in another order -- is as far from the HIP path per tensor as the
# fp32 oracle is: rel-L2 max 5.3e-2 (class medians 1.0 - 1.3e-2) against the emulation, 5.4e-2 (1.1 - 1.4e-2) against the oracle, and the emulation itself 5.3e-2
# from the oracle.  What the emulation does sharpen is the NORM ratio, the measure a wrong scale cannot hide from: | ratio - 1 | max 6.2e-3 against the
# emulation, 3.0e-2 against the oracle.  Bounds = those maxima with ~25 % head room; the per-tensor bound was 0.12 (0.2 behind l2-normalise) until round 5.
REL_D12 = 6.5e-2            # per tensor, every class, against either reference
MEDIAN_D12 = 2.0e-2         # per class
NORM_D12 = {'emu': 8e-3, 'oracle': 4e-2}


@pytest.mark.timeout(1800)
def test_config2_full_depth_gradients_by_class_against_the_bf16_emulation():
    from grad_parity import run_all, tensor_class
    from test_grad_classes_gpu import REL
    res = run_all('config2d12')
    lh, le, lo = res['loss']
    assert abs(lh - lo) < 2e-2 and abs(le - lo) < 2e-2 and abs(lh - le) < 1e-2, res['loss']
    bad, seen = [], set()
    rels = {}
    for n, gh in res['hip'].items():
        if n.endswith('key_layer/bias'):
            continue
        gh = gh.double()
        c = tensor_class(n)
        for ref_name in ('emu', 'oracle'):
            if n not in res[ref_name]:
                continue
            gr = res[ref_name][n].double()
            if float(gr.norm()) == 0.0:
                continue
            rel = float((gh - gr).norm() / gr.norm())
            ratio = float(gh.norm() / gr.norm()) - 1.0
            seen.add(c)
            rels.setdefault((ref_name, c), []).append(rel)
            if rel > REL_D12 or abs(ratio) > NORM_D12[ref_name]:
                bad.append((ref_name, c, n, rel, ratio))
    assert seen == set(REL), seen
    assert not bad, sorted(bad, key=lambda t: -t[3])[:12]
    import numpy as np
    med = {k: float(np.median(v)) for k, v in rels.items()}
    assert all(m < MEDIAN_D12 for m in med.values()), med
