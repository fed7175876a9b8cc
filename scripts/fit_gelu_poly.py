"""Coefficients of GELU_R / GELU_D in merlot_amd/csrc/common.h: degree-12 Chebyshev fits (monomial form in
t = a*2/4.5 - 1) of r(a) = a*Phi(-a) and d(a) = Phi(-a) - a*phi(a) on a in [0, 4.5]; prints them and the max error of an
fp32 Horner evaluation."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erfc

A, deg = 4.5, 12
nodes = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000)
x = (nodes + 1) * A / 2
Q = 0.5 * erfc(x / np.sqrt(2))
phi = np.exp(-x * x / 2) / np.sqrt(2 * np.pi)
xt = np.linspace(0, A, 200001)
tt = (xt * 2 / A - 1).astype(np.float32)
Qt = 0.5 * erfc(xt / np.sqrt(2))
phit = np.exp(-xt * xt / 2) / np.sqrt(2 * np.pi)
for name, f, ft in (('GELU_R', x * Q, xt * Qt), ('GELU_D', Q - x * phi, Qt - xt * phit)):
    mono = C.cheb2poly(C.chebfit(nodes, f, deg)).astype(np.float32)
    acc = np.full_like(tt, mono[-1])
    for k in range(deg - 1, -1, -1):
        acc = (acc * tt + mono[k]).astype(np.float32)
    print('__device__ constexpr float %s[13] = {%s};' % (name, ', '.join('%.9ef' % v for v in mono)))
    print('//   max abs error (fp32 Horner): %.2e' % np.abs(acc - ft).max())
