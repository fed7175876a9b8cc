"""GPU, BASELINE config #2 at FULL depth against the ORACLE (VERDICT r1 weak #1e): merlot.yaml 4-segment groups, 224^2
frames, ViT-B/16 12 layers + 12-layer text-only pass + 12-layer joint encoder, 16 chunks per example (text-only
sequence 512, joint 4 groups x 328), one example -- the oracle (fp32, torch-CPU) does forward + backward of that in a
few seconds, so the full-depth path is compared value for value, not only through properties
(tests/test_zz_full_depth_gpu.py keeps the property checks at batch 4).

Tolerances (bf16 policy on the GPU vs the fp32 oracle): masked ids / idx exact, attention_summs <= 1e-2,
hidden states and contrastive targets rel-L2 <= 2e-2, losses <= 1e-2 abs, gradients <= 0.12 rel-L2 per tensor (0.2
behind l2-normalise), median <= 3e-2 -- the same numbers as the 2-layer tests: depth must not loosen them.
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
    total = _check(cfg, b, w, m, info, st, pm, with_grads=True)
    assert abs(total - float(loss)) < 2e-2
