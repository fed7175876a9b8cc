"""Independent anchors for the riskiest primitives of oracle/tf_shim.py (VERDICT r1 weak #1a/#1b, next-round item 6).

The "reference run" fixtures execute the reference's control flow on the shim's primitives; a misreading of a TF
primitive shared by the shim and the oracle would be invisible there.  Nobody can run TensorFlow 1.15 here, so each
primitive below is checked against something written by SOMEBODY ELSE or against its documented contract evaluated by a
different algorithm -- none of these tests routes through oracle/input_oracle.py or oracle/merlot_oracle.py for its
expectation:
  * tf.math.top_k / tf.argsort / tf.argmax tie order  -> the documented contract ("if two elements are equal, the
    lower-index element appears first"; argmax returns the smallest index of a maximum) evaluated by plain Python loops;
  * tf.random.categorical                           -> only its DISTRIBUTION matters (every draw is recorded and injected
    on both sides): frequencies against softmax(logits);
  * tf.image.resize_images(align_corners=True): bilinear and bicubic -> PyTorch's own kernels
    (F.interpolate(align_corners=True); its bicubic uses the same Keys coefficient A = -0.75 as TF's legacy
    ResizeBicubic, without TF's 1/1024 weight table: agreement to the table's resolution), area -> the invariants of any area-averaging kernel (no
    independent implementation of its align_corners geometry exists here), nearest -> the index formula of the TF 1.15 kernel
    (`round(i * (in-1)/(out-1))`) evaluated per pixel in Python;
  * tf.nn.moments / layer_norm                       -> torch.nn.functional.layer_norm;  erf-GELU -> F.gelu;
    tf.nn.softmax -> torch.softmax in float64;  tf.math.l2_normalize -> F.normalize.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tf_shim


@pytest.fixture(scope='module')
def tf():
    return tf_shim.build_modules()['tensorflow']


def test_top_k_argsort_argmax_tie_rules(tf):
    x = torch.tensor([[0.5, 2.0, 2.0, -1.0, 2.0, 0.5, 0.5], [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]])
    vals, idx = tf.math.top_k(x, k=5)
    for r in range(x.shape[0]):
        order = sorted(range(x.shape[1]), key=lambda j: (-float(x[r, j]), j))        # descending, lower index first
        assert idx[r].tolist() == order[:5] and vals[r].tolist() == [float(x[r, j]) for j in order[:5]]
        asc = sorted(range(x.shape[1]), key=lambda j: (float(x[r, j]), j))
        assert tf.argsort(x, 1)[r].tolist() == asc
        assert int(tf.argmax(x, 1)[r]) == min(j for j in range(x.shape[1]) if x[r, j] == x[r].max())


def test_random_categorical_distribution(tf):
    tf_shim.STATE.reset(seed=5)
    logits = torch.log(torch.tensor([[0.6, 1e-6, 0.1, 0.1, 0.2]]))
    draws = tf.random.categorical(logits, num_samples=200000, dtype=torch.int32).reshape(-1)
    freq = torch.bincount(draws.long(), minlength=5).double() / draws.numel()
    want = torch.softmax(logits.double(), -1)[0]
    assert float((freq - want).abs().max()) < 4e-3 and freq[1] < 1e-4


@pytest.mark.parametrize("hw,out", [((37, 53), (64, 48)), ((120, 90), (31, 77)), ((16, 16), (16, 16)), ((9, 200), (64, 64))])
def test_resize_bilinear_and_bicubic_against_pytorch_kernels(tf, hw, out):
    g = torch.Generator().manual_seed(sum(hw))
    img = torch.rand((*hw, 3), generator=g)
    nchw = img.permute(2, 0, 1)[None]
    got = tf.image.resize_images(img, list(out), method=tf.image.ResizeMethod.BILINEAR, align_corners=True)
    ref = F.interpolate(nchw, size=out, mode='bilinear', align_corners=True)[0].permute(1, 2, 0)
    assert float((got - ref).abs().max()) < 2e-6
    got = tf.image.resize_images(img, list(out), method=tf.image.ResizeMethod.BICUBIC, align_corners=True)
    ref = F.interpolate(nchw, size=out, mode='bicubic', align_corners=True)[0].permute(1, 2, 0)
    assert float((got - ref).abs().max()) < 3e-3          # TF tabulates the Keys weights at 1/1024 steps; PyTorch does not


@pytest.mark.parametrize("hw,f", [((64, 96), 2), ((90, 60), 3), ((32, 32), 1), ((40, 100), 4)])
def test_resize_area_invariants(tf, hw, f):
    """AREA with align_corners=True (scale (in-1)/(out-1)) has no counterpart in PyTorch or PIL, so it is anchored by
    what any area-averaging kernel must satisfy: same-size resize is the identity, constants stay constant, every output
    is a convex combination of inputs (range-bounded) and the image mean is preserved up to the edge-weight asymmetry."""
    g = torch.Generator().manual_seed(hw[0] + f)
    img = torch.rand((*hw, 3), generator=g)
    area = tf.image.ResizeMethod.AREA
    same = tf.image.resize_images(img, list(hw), method=area, align_corners=True)
    assert float((same - img).abs().max()) < 1e-6
    const = torch.full((*hw, 3), 0.37)
    out = tf.image.resize_images(const, [hw[0] // f, hw[1] // f], method=area, align_corners=True)
    assert float((out - 0.37).abs().max()) < 5e-6
    small = tf.image.resize_images(img, [max(2, hw[0] // f), max(2, hw[1] // f)], method=area, align_corners=True)
    assert float(small.min()) >= float(img.min()) - 1e-6 and float(small.max()) <= float(img.max()) + 1e-6
    assert abs(float(small.mean()) - float(img.mean())) < 0.03


def test_resize_nearest_index_formula(tf):
    img = torch.arange(7 * 11, dtype=torch.float32).reshape(7, 11, 1).repeat(1, 1, 3)
    oh, ow = 5, 16
    got = tf.image.resize_images(img, [oh, ow], method=tf.image.ResizeMethod.NEAREST_NEIGHBOR, align_corners=True)
    sy, sx = (7 - 1) / (oh - 1), (11 - 1) / (ow - 1)
    for y in range(oh):
        for x in range(ow):
            iy = min(int(math.floor(y * np.float32(sy) + 0.5)), 6)               # roundf(y * scale), TF 1.15 resize_nearest_neighbor_op
            ix = min(int(math.floor(x * np.float32(sx) + 0.5)), 10)
            assert float(got[y, x, 0]) == float(img[iy, ix, 0])


def test_numeric_primitives_against_pytorch(tf):
    g = torch.Generator().manual_seed(3)
    x = torch.randn((5, 7, 768), generator=g) * 3 + 1
    mean, var = tf.nn.moments(x, [2], keep_dims=True) if 'keep_dims' in tf.nn.moments.__code__.co_varnames else tf.nn.moments(x, [2], keepdims=True)
    assert float((mean - x.mean(-1, keepdim=True)).abs().max()) < 1e-5
    assert float((var - x.var(-1, unbiased=False, keepdim=True)).abs().max()) < 1e-4      # population variance
    sm = tf.nn.softmax(x, axis=-1) if 'axis' in tf.nn.softmax.__code__.co_varnames else tf.nn.softmax(x)
    assert float((sm.double() - torch.softmax(x.double(), -1)).abs().max()) < 1e-6
    n = tf.math.l2_normalize(x, axis=-1)
    assert float((n - F.normalize(x, dim=-1, eps=1e-6)).abs().max()) < 1e-6
    gelu = x * 0.5 * (1.0 + tf.erf(x / math.sqrt(2.0)))                  # utils/model_utils.py:96-110 on the shim's erf
    assert float((gelu - F.gelu(x)).abs().max()) < 1e-5


@pytest.mark.parametrize("hw,out", [((64, 96), (32, 48)), ((90, 61), (31, 20)), ((37, 53), (64, 70)), ((120, 90), (33, 77)),
                                    ((16, 16), (16, 16))])
def test_resize_area_against_a_brute_force_box_integral(tf, hw, out):
    """VERDICT r2 item 4c.  tf.image.resize_area's published definition: output pixel o covers the source interval
    [o * scale, (o + 1) * scale) per axis (scale = (in - 1) / (out - 1) under align_corners, the legacy scaler), the result is
    the integral of the piecewise-constant source image over that box divided by the box area, source indices clamped to the
    image.  Here that integral is evaluated directly -- exact overlap lengths min(i + 1, b) - max(i, a) in rational arithmetic,
    float64 accumulation, one output pixel at a time -- with no reference to oracle/input_oracle.py's span / case analysis."""
    from fractions import Fraction
    g = torch.Generator().manual_seed(hw[0] * 7 + out[1])
    img = torch.rand((*hw, 2), generator=g)
    got = tf.image.resize_images(img, list(out), method=tf.image.ResizeMethod.AREA, align_corners=True).double().numpy()
    src = img.double().numpy()

    def overlaps(o, n_in, n_out):
        s = Fraction(n_in - 1, n_out - 1)
        a, b = o * s, (o + 1) * s
        w = {}
        for i in range(math.floor(a), math.ceil(b)):
            ov = min(Fraction(i + 1), b) - max(Fraction(i), a)
            if ov > 0:
                j = min(max(i, 0), n_in - 1)
                w[j] = w.get(j, Fraction(0)) + ov
        return w, s

    wy = [overlaps(y, hw[0], out[0]) for y in range(out[0])]
    wx = [overlaps(x, hw[1], out[1]) for x in range(out[1])]
    worst = 0.0
    for y in range(0, out[0], max(1, out[0] // 9)):          # a lattice of output pixels incl. both borders is sample enough
        for x in list(range(0, out[1], max(1, out[1] // 9))) + [out[1] - 1]:
            (ry, sy), (rx, sx) = wy[y], wx[x]
            acc = np.zeros(2)
            for i, a in ry.items():
                for j, b in rx.items():
                    acc += float(a * b) * src[i, j]
            want = acc / float(sy * sx)
            worst = max(worst, float(np.abs(got[y, x] - want).max()))
    (ry, sy), (rx, sx) = wy[out[0] - 1], wx[out[1] - 1]
    want = sum(float(a * b) * src[i, j] for i, a in ry.items() for j, b in rx.items()) / float(sy * sx)
    worst = max(worst, float(np.abs(got[-1, -1] - want).max()))
    assert worst < 2e-5, worst


def test_conv2d_and_avg_pool_layout_conventions_on_a_hand_computed_case(tf):
    """the shim evaluates tf.nn.conv2d / tf.layers.conv2d / tf.nn.avg_pool2d with F.conv2d / F.avg_pool2d on
    permuted tensors.  The conventions that permutation must realise -- input NHWC, kernel HWIO, cross-correlation (no kernel
    flip), VALID windows anchored top-left, SAME = zero padding split low-first -- are checked against plain Python loops over
    the definition out[n, y, x, o] = sum_{ky, kx, c} in[n, y*s + ky, x*s + kx, c] * k[ky, kx, c, o] on an asymmetric input."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 5, 6, 3), generator=g)                       # N H W C, H != W, C != O
    k = torch.randn((2, 3, 3, 4), generator=g)                       # kh kw I O, kh != kw
    got = tf.nn.conv2d(x, k, strides=[1, 1, 1, 1], padding='VALID')
    assert list(got.shape) == [2, 4, 4, 4]
    for n in range(2):
        for y in range(4):
            for xx in range(4):
                for o in range(4):
                    want = sum(float(x[n, y + ky, xx + kx, c]) * float(k[ky, kx, c, o])
                               for ky in range(2) for kx in range(3) for c in range(3))
                    assert abs(float(got[n, y, xx, o]) - want) < 1e-5
    # stride 2, VALID: windows at 0, 2 (H: (5 - 2) // 2 + 1 = 2 rows, W: (6 - 3) // 2 + 1 = 2 columns)
    got2 = tf.nn.conv2d(x, k, strides=[1, 2, 2, 1], padding='VALID')
    assert list(got2.shape) == [2, 2, 2, 4]
    want = sum(float(x[1, 2 + ky, 2 + kx, c]) * float(k[ky, kx, c, 3]) for ky in range(2) for kx in range(3) for c in range(3))
    assert abs(float(got2[1, 1, 1, 3]) - want) < 1e-5
    # SAME, 3x3, stride 1: one zero row / column on every side
    k3 = torch.randn((3, 3, 3, 2), generator=g)
    got3 = tf.nn.conv2d(x, k3, strides=[1, 1, 1, 1], padding='SAME')
    assert list(got3.shape) == [2, 5, 6, 2]
    for (y, xx) in [(0, 0), (4, 5), (2, 3), (0, 5)]:
        want = 0.0
        for ky in range(3):
            for kx in range(3):
                iy, ix = y + ky - 1, xx + kx - 1
                if 0 <= iy < 5 and 0 <= ix < 6:
                    want += sum(float(x[0, iy, ix, c]) * float(k3[ky, kx, c, 1]) for c in range(3))
        assert abs(float(got3[0, y, xx, 1]) - want) < 1e-5
    # tf.layers.conv2d: the variable it creates is 'kernel' [kh, kw, in, filters] (HWIO) + 'bias' [filters]
    tf_shim.STATE.reset(seed=0, injected={'probe/kernel': k.numpy(), 'probe/bias': np.array([0.5, -1.0, 2.0, 0.25], np.float32)})
    got4 = tf.layers.conv2d(x, 4, (2, 3), strides=(1, 1), padding='valid', name='probe')
    assert float((got4 - (got + torch.tensor([0.5, -1.0, 2.0, 0.25]))).abs().max()) < 1e-6
    tf_shim.STATE.reset(seed=0)
    # avg_pool2d 2x2 / 2 on NHWC: the mean of the 4 pixels of each window, channels untouched
    xp = torch.randn((1, 4, 6, 3), generator=g)
    pooled = tf.nn.avg_pool2d(xp, 2, 2, 'VALID')
    assert list(pooled.shape) == [1, 2, 3, 3]
    for y in range(2):
        for xx in range(3):
            for c in range(3):
                want = sum(float(xp[0, 2 * y + dy, 2 * xx + dx, c]) for dy in range(2) for dx in range(2)) / 4.0
                assert abs(float(pooled[0, y, xx, c]) - want) < 1e-6


# ---- round 4 (VERDICT r3 weak 1a / next #5c): GroupNorm moments layout, tf.image crop / pad corner cases -------------------------
def test_sufficient_statistics_and_normalize_moments_against_torch(tf):
    """`tf.nn.sufficient_statistics` / `tf.nn.normalize_moments` (what utils/model_utils.py:196-201 builds GroupNorm's one-pass
    moments from): count, sum x, sum x^2 over the given axes; mean = sum / count, variance = sum x^2 / count - mean^2 (population
    variance, TF's documented definition) -- against torch.mean / torch.var(unbiased=False) in float64."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 5, 7, 4, 6, generator=g) * 2.0 + 0.7
    cnt, m_ss, v_ss, shift = tf.nn.sufficient_statistics(x, [1, 2, 4], keep_dims=True)
    assert shift is None and float(cnt) == 5 * 7 * 6
    mean, var = tf.nn.normalize_moments(cnt, m_ss, v_ss, shift=None)
    xd = x.double()
    assert tuple(mean.shape) == (3, 1, 1, 4, 1)
    assert torch.allclose(mean.double(), xd.mean(dim=(1, 2, 4), keepdim=True), atol=1e-6)
    assert torch.allclose(var.double(), xd.var(dim=(1, 2, 4), unbiased=False, keepdim=True), atol=2e-5)


_GN_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from oracle import tf_shim
tf_shim.install()
sys.path.insert(1, '/root/reference')
from utils import model_utils as ref
import tensorflow as tf
out = {}
for name, (n, h, w, c) in {'c64': (2, 6, 5, 64), 'c256': (1, 4, 4, 256)}.items():
    g = torch.Generator().manual_seed(c)
    x = torch.randn(n, h, w, c, generator=g) * 1.5 + 0.3
    with tf.variable_scope('anchor_' + name):
        y = ref.group_norm(tf.constant(x), epsilon=1e-4, name='t')
    out[name + '_x'] = x.numpy(); out[name + '_y'] = np.asarray(y.detach() if hasattr(y, 'detach') else y)
np.savez(sys.argv[2], **out)
'''


@pytest.mark.skipif(not __import__('os').path.isdir('/root/reference/utils'), reason="reference sources only exist in the build container")
def test_reference_group_norm_under_the_shim_against_torch_group_norm(tmp_path):
    """The reference's OWN `group_norm` (utils/model_utils.py:133-222: reshape [N,H,W,C] -> [N,H,W,32,C/32], moments over
    H, W and the within-group channel axis, eps inside the rsqrt, gamma 1 / beta 0 at creation) executed under the shim, against
    `torch.nn.functional.group_norm` on the NCHW view -- an implementation by somebody else with the same documented conventions
    (contiguous channel groups, population variance).  This anchors the moments LAYOUT the ResNet-stem fixtures rest on."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'gn.npz')
    r = subprocess.run([sys.executable, '-c', _GN_SCRIPT, root, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(out)
    for name in ('c64', 'c256'):
        x, y = torch.from_numpy(d[name + '_x']), torch.from_numpy(d[name + '_y'])
        ref = F.group_norm(x.permute(0, 3, 1, 2).double(), 32, eps=1e-4).permute(0, 2, 3, 1)
        assert float((y.double() - ref).abs().max()) < 1e-5, name


def test_pad_to_bounding_box_and_slice_corner_cases(tf):
    """`image[oy:oy+H, ox:ox+W]` + `tf.image.pad_to_bounding_box(image, 0, 0, H, W)` (utils/model_utils.py:923-924) on the corner
    cases of resize_and_pad: the scaled frame smaller than the target in one or both dimensions (zeros appended BELOW / to the RIGHT
    only, content anchored at the top-left), equal to it (unchanged), larger (cropped at the offset first; the slice clamps at the
    border like a Python slice), a 1-pixel frame -- against explicit per-pixel loops over the documented contract."""
    rng = np.random.RandomState(0)
    for (h, w, H, W, oy, ox) in [(5, 7, 8, 9, 0, 0), (8, 9, 8, 9, 0, 0), (12, 7, 8, 9, 3, 0), (12, 15, 8, 9, 4, 6), (1, 1, 4, 4, 0, 0),
                                 (9, 9, 8, 9, 1, 0), (8, 20, 8, 9, 0, 11)]:
        img = rng.uniform(size=(h, w, 3)).astype(np.float32)
        cropped = torch.from_numpy(img)[oy:oy + H, ox:ox + W, :]
        got = np.asarray(tf.image.pad_to_bounding_box(cropped, 0, 0, H, W))
        want = np.zeros((H, W, 3), np.float32)
        for y in range(H):
            for x in range(W):
                if oy + y < h and ox + x < w:
                    want[y, x] = img[oy + y, ox + x]
        assert got.shape == (H, W, 3) and np.array_equal(got, want), (h, w, H, W, oy, ox)
    # a non-zero offset places the content there (the documented contract; resize_and_pad only uses 0, 0)
    img = rng.uniform(size=(2, 3, 3)).astype(np.float32)
    got = np.asarray(tf.image.pad_to_bounding_box(torch.from_numpy(img), 1, 2, 5, 6))
    want = np.zeros((5, 6, 3), np.float32)
    want[1:3, 2:5] = img
    assert np.array_equal(got, want)
