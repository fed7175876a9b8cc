"""TEST INFRASTRUCTURE (CPU oracle; only tests/, smoke() and bench.py's cpu_baseline may import this).

numpy restatement of the pixel half of libjpeg's default decoder -- what `tf.image.decode_jpeg(x, channels=3)`
(model/dataloader.py:72-77) computes after entropy decoding: dequantisation, the "islow" integer inverse DCT (Loeffler /
Ligtenberg / Moschytz, CONST_BITS 13, PASS1_BITS 2), the h2v2 "fancy" triangle chroma upsampler, the 16-bit fixed-point
YCbCr -> RGB conversion.  Input: the quantised coefficients and the header that `merlot_jpeg_entropy_decode` (host C++)
produces.  PINNED against the host library itself: tests/test_jpeg.py decodes the same files with PIL (libjpeg-turbo) and
demands equality bit for bit."""
import numpy as np

C13 = dict(F_0_298=2446, F_0_390=3196, F_0_541=4433, F_0_765=6270, F_0_899=7373, F_1_175=9633, F_1_501=12299, F_1_847=15137,
           F_1_961=16069, F_2_053=16819, F_2_562=20995, F_3_072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct8(d, shift):
    """d: int64 [..., 8] -> one LL&M pass along the last axis."""
    K = C13
    z2, z3 = d[..., 2], d[..., 6]
    z1 = (z2 + z3) * K['F_0_541']
    tmp2 = z1 + z3 * (-K['F_1_847'])
    tmp3 = z1 + z2 * K['F_0_765']
    z2, z3 = d[..., 0], d[..., 4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = d[..., 7], d[..., 5], d[..., 3], d[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * K['F_1_175']
    tmp0, tmp1, tmp2, tmp3 = tmp0 * K['F_0_298'], tmp1 * K['F_2_053'], tmp2 * K['F_3_072'], tmp3 * K['F_1_501']
    z1, z2, z3, z4 = z1 * -K['F_0_899'], z2 * -K['F_2_562'], z3 * -K['F_1_961'] + z5, z4 * -K['F_0_390'] + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], -1)
    return _descale(out, shift)


def planes_from_coefficients(coef, info):
    """-> [Y, Cb, Cr] uint8 planes of whole blocks."""
    planes = []
    for c in range(3):
        bw, bh = info['blocks_w'][c], info['blocks_h'][c]
        q = coef[info['coef_offset'][c]:info['coef_offset'][c] + bw * bh * 64].astype(np.int64).reshape(bh, bw, 8, 8)
        q = q * np.asarray(info['quant'][c], np.int64).reshape(8, 8)
        cols = _idct8(q.transpose(0, 1, 3, 2), 13 - 2).transpose(0, 1, 3, 2)          # pass 1: down each column
        rows = _idct8(cols, 13 + 2 + 3)                                                # pass 2: along each row
        px = np.clip(rows + 128, 0, 255).astype(np.uint8)
        planes.append(px.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8))
    return planes


def _fancy_h2v2(p, W, H):
    dw, dh = (W + 1) // 2, (H + 1) // 2
    p = p[:dh, :dw].astype(np.int64)
    up = np.concatenate([p[:1], p[:-1]], 0)                                            # row above (first row repeats itself)
    dn = np.concatenate([p[1:], p[-1:]], 0)
    out = np.zeros((2 * dh, 2 * dw), np.int64)
    for v, other in ((0, up), (1, dn)):
        cs = 3 * p + other                                                             # column sums of the two nearest rows
        left = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        right = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        even = (cs * 3 + left + 8) >> 4
        odd = (cs * 3 + right + 7) >> 4
        even[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out[:H, :W]


def rgb_from_planes(planes, info):
    W, H = info['width'], info['height']
    Y = planes[0][:H, :W].astype(np.int64)
    if info['subsampling'] == 1:
        cb, cr = planes[1][:H, :W].astype(np.int64), planes[2][:H, :W].astype(np.int64)
    else:
        cb, cr = _fancy_h2v2(planes[1], W, H), _fancy_h2v2(planes[2], W, H)
    xb, xr = cb - 128, cr - 128
    r = Y + ((91881 * xr + 32768) >> 16)
    g = Y + ((-22554 * xb + 32768 - 46802 * xr) >> 16)
    b = Y + ((116130 * xb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode(coef, info):
    return rgb_from_planes(planes_from_coefficients(coef, info), info)
