"""The GEMM epilogues' GELU / GELU' polynomials (merlot_amd/csrc/common.h), evaluated here on the CPU exactly as the kernel does
(fp32 Horner in t = min(|x|, 4.5) * 2/4.5 - 1) against the exact erf forms of utils/model_utils.py:96-110: the error bounds the
header and DESIGN.md quote are the ones the COMMITTED coefficients have."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coeffs(name):
    src = open(os.path.join(ROOT, 'merlot_amd', 'csrc', 'common.h')).read()
    m = re.search(r'__device__ constexpr float %s\[(\d+)\] = \{([^}]*)\};' % name, src)
    c = np.array([float(v.rstrip('f')) for v in m.group(2).split(',')], dtype=np.float32)
    assert len(c) == int(m.group(1))
    return c


def _horner(c, a):
    t = (np.minimum(a, np.float32(4.5)) * np.float32(2.0 / 4.5) + np.float32(-1.0)).astype(np.float32)
    acc = np.full_like(t, c[-1])
    for k in range(len(c) - 2, -1, -1):
        acc = (acc * t + c[k]).astype(np.float32)
    return acc


def test_gelu_epilogue_polynomials_meet_their_stated_error():
    x = np.linspace(-9, 9, 1800001).astype(np.float32)
    xd = x.astype(np.float64)
    cdf = 0.5 * (1.0 + erf(xd / np.sqrt(2.0)))
    pdf = np.exp(-0.5 * xd * xd) / np.sqrt(2 * np.pi)
    r = _horner(_coeffs('GELU_R'), np.abs(x))
    y = np.maximum(x, np.float32(0)) - r                        # gelu_fast2
    err = np.abs(y - xd * cdf)
    assert err.max() < 1.7e-5, err.max()
    # below half a bf16 ulp (2^-9 |y|) wherever |gelu| >= 0.008
    big = np.abs(xd * cdf) >= 0.008
    assert (err[big] <= np.abs(xd * cdf)[big] * 2.0 ** -9).all()
    d = _horner(_coeffs('GELU_D'), np.abs(x))
    g = np.where(x >= 0, np.float32(1.0) - d, d)                # gelu_grad_fast2
    errg = np.abs(g - (cdf + xd * pdf))
    assert errg.max() < 1.2e-4, errg.max()
    assert len(_coeffs('GELU_R')) == 11 and len(_coeffs('GELU_D')) == 11
