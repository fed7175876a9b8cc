"""GPU, BASELINE config #2 at FULL depth against the ORACLE (VERDICT r1 weak #1e): merlot.yaml 4-segment groups, 224^2
frames, ViT-B/16 12 layers + 12-layer text-only pass + 12-layer joint encoder, 16 chunks per example (text-only
sequence 512, joint 4 groups x 328), one example -- the oracle (fp32, torch-CPU) does forward + backward of that in a
few seconds, so the full-depth path is compared value for value, not only through properties
(tests/test_zz_full_depth_gpu.py keeps the property checks at batch 4).

Tolerances (bf16 policy on the GPU vs the fp32 oracle): masked ids / idx exact, attention_summs <= 1e-2,
hidden states and contrastive targets rel-L2 <= 2e-2, losses <= 1e-2 abs.  Gradients (round 6): what 36 layers of bf16
rounding do to a gradient is NOISE against the fp32 oracle (measured: profiles/r06_c_grad_depth12.txt), so the oracle comparison keeps the
per-tensor bound the measurement supports and the sharp statement is made against the torch EMULATION of the same bf16 policy (the product's
host code on tests/emu_ops.py, rounding where the kernels round): per tensor class rel-L2 <= 3e-2 / 4e-2 and | norm ratio - 1 | <= 6e-3 --
the bounds of the 2-layer tests (tests/test_grad_classes_gpu.py), not loosened by depth.
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


# HIP vs the fp32 oracle at depth 12 + 12 + 12: bf16 rounding noise of 36 layers.  Bounds = the measured worst case of each class with head room
# (profiles/r06_c_grad_depth12.txt); the per-class 3e-2 / 4e-2 of the 2-layer problems is asserted against the emulation below.
ORACLE_REL_D12 = {'bias': 0.12, 'ln': 0.12, 'pos': 0.12, 'emb': 0.12, 'kernel': 0.12, 'contrastive': 0.2}
ORACLE_NORM_D12 = 6e-2


@pytest.mark.timeout(1800)
def test_config2_full_depth_gradients_by_class_against_the_bf16_emulation():
    from grad_parity import run_all, tensor_class
    from test_grad_classes_gpu import REL, NORM
    res = run_all('config2d12')
    lh, le, lo = res['loss']
    assert abs(lh - lo) < 2e-2 and abs(le - lo) < 2e-2 and abs(lh - le) < 1e-2, res['loss']
    bad, seen = [], set()
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
            if ref_name == 'emu':
                seen.add(c)
                lim_rel, lim_norm = REL[c], NORM['emu']
            else:
                lim_rel = ORACLE_REL_D12['contrastive' if n.startswith('contrastive/') else c]
                lim_norm = ORACLE_NORM_D12
            if rel > lim_rel or abs(ratio) > lim_norm:
                bad.append((ref_name, c, n, rel, ratio))
    assert seen == set(REL), seen
    assert not bad, sorted(bad, key=lambda t: -t[3])[:12]
