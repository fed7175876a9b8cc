"""CPU: the host half of the JPEG decoder (merlot_jpeg_entropy_decode, C++) + the numpy restatement of libjpeg's pixel stages
(oracle/jpeg_oracle.py) against the host library itself: PIL / libjpeg-turbo decodes the same files, equality bit for bit.
This pins the oracle the GPU kernels are compared with (tests/test_jpeg_gpu.py)."""
import io

import numpy as np
import pytest

from merlot_amd import jpeg
from oracle import jpeg_oracle


def _image(h, w, seed):
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(yy * 3 + xx) % 256, (xx * 5 + seed * 17) % 256, ((yy - xx) * 2) % 256], -1).astype(np.float64)
    base += r.normal(0, 25, base.shape)                     # texture: exercises many AC coefficients
    base[h // 3:h // 3 + 9, :] = 255                         # saturated edges: range limiting
    base[:, w // 2:w // 2 + 5] = 0
    return np.clip(base, 0, 255).astype(np.uint8)


def _encode(arr, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format='JPEG', **kw)
    return b.getvalue()


def _pil(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))


CASES = [(64, 64, dict(quality=75)), (90, 120, dict(quality=75)), (37, 53, dict(quality=90)), (224, 224, dict(quality=95)),
         (41, 99, dict(quality=30)), (8, 8, dict(quality=75)), (17, 16, dict(quality=75, subsampling=0)),
         (100, 75, dict(quality=85, subsampling=0)), (123, 77, dict(quality=60, optimize=True)), (1, 1, dict(quality=75)),
         (240, 320, dict(quality=75)), (33, 47, dict(quality=100, subsampling=2))]


@pytest.mark.parametrize('h,w,kw', CASES)
def test_entropy_decode_plus_oracle_equals_libjpeg(h, w, kw):
    data = _encode(_image(h, w, h * 7 + w), **kw)
    got = jpeg.entropy_decode(data)
    assert got is not None
    coef, info = got
    i = jpeg.info_dict(info)
    assert (i['height'], i['width']) == (h, w)
    assert i['subsampling'] == (1 if kw.get('subsampling', 2) == 0 else 2)
    rgb = jpeg_oracle.decode(coef, i)
    assert np.array_equal(rgb, _pil(data))


def test_restart_intervals_and_unsupported_files():
    from PIL import Image
    arr = _image(70, 110, 3)
    try:
        data = _encode(arr, quality=80, restart_marker_blocks=2)
    except TypeError:
        data = None
    if data is not None and b'\xff\xdd' in data:
        coef, info = jpeg.entropy_decode(data)
        assert np.array_equal(jpeg_oracle.decode(coef, jpeg.info_dict(info)), _pil(data))
    assert jpeg.entropy_decode(_encode(arr, quality=75, progressive=True)) is None            # -> host library
    assert jpeg.entropy_decode(_encode(arr[:, :, 0], quality=75)) is None                     # grayscale
    assert jpeg.entropy_decode(_encode(arr, quality=75, subsampling=1)) is None               # 4:2:2
    # malformed / truncated files and decompression bombs are handed to the host decoder: None, never a raise and
    # never a coefficient buffer sized from an unchecked header
    good = _encode(arr, quality=75)
    assert jpeg.entropy_decode(good[:200]) is None                                            # truncated
    assert jpeg.entropy_decode(b'not a jpeg at all') is None
    bomb = bytearray(good)
    sof = bomb.index(b'\xff\xc0')
    bomb[sof + 5:sof + 9] = b'\xff\xff\xff\xff'                                               # SOF0 height / width = 65535
    assert jpeg.entropy_decode(bytes(bomb)) is None
