"""GPU: every parameter gradient of the HIP path by TENSOR CLASS, against the fp32 oracle and against the torch emulation of the same
bf16 policy (tests/grad_parity.py), with two measures per tensor:
  * rel-L2 <= 3e-2 for biases, LayerNorm gamma / beta, position / CLS tables and the word embeddings, <= 4e-2 for dense kernels
    (measured on BASELINE config #1: <= 2.1e-2 everywhere, profiles/r04_c_grad_classes.txt);
  * the NORM RATIO | ||g_hip|| / ||g_ref|| - 1 | <= 1.5e-2 against the oracle (measured <= 6.7e-3) and <= 6e-3 against the emulation
    (measured <= 2.1e-3; 7e-4 at the larger problem): rounding noise is incoherent and moves a norm by its square, a wrong scale
    on one small tensor (a bias, a gamma: what a 0.12 rel-L2 bound cannot see) moves it linearly.
Two problems: config #1 exactly (64^2, 2 + 2 + 2 layers, 2 examples x 4 segments) and config #2's geometry at 8 EXAMPLES x 16
segments (224^2, joint S = 328, 25 344 ViT tokens = 99 row tiles x 3-12 column tiles: several claimed tiles per workgroup, the
batch-dependent paths of the persistent GEMMs) with 2 + 2 + 2 layers so that the oracle finishes in seconds.
Reference: tf.gradients at utils/optimization.py:176 over model/modeling.py:47-668."""
import numpy as np
import pytest

from grad_parity import run_all, tensor_class

pytestmark = pytest.mark.gpu

REL = {'bias': 3e-2, 'ln': 3e-2, 'pos': 3e-2, 'emb': 3e-2, 'kernel': 4e-2}
NORM = {'oracle': 1.5e-2, 'emu': 6e-3}


def _check(which):
    res = run_all(which)
    lh, le, lo = res['loss']
    assert abs(lh - lo) < 2e-2 and abs(le - lo) < 2e-2, res['loss']
    bad, seen = [], set()
    for ref_name in ('oracle', 'emu'):
        ref = res[ref_name]
        for n, gh in res['hip'].items():
            if n.endswith('key_layer/bias') or n not in ref:     # the key bias gradient is identically 0 (softmax shift invariance)
                continue
            gr = ref[n].double()
            gh = gh.double()
            if float(gr.norm()) == 0.0:
                continue
            c = tensor_class(n)
            seen.add(c)
            rel = float((gh - gr).norm() / gr.norm())
            ratio = float(gh.norm() / gr.norm()) - 1.0
            if rel > REL[c] or abs(ratio) > NORM[ref_name]:
                bad.append((ref_name, c, n, rel, ratio))
    assert seen == set(REL), seen
    assert not bad, sorted(bad, key=lambda t: -t[3])[:12]
    return res


def test_gradients_by_tensor_class_config1():
    res = _check('config1')
    assert len(res['hip']) >= 110


@pytest.mark.timeout(900)
def test_gradients_by_tensor_class_config2_geometry_8_examples():
    _check('config2x8')
