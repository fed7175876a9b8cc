"""`tf.image.decode_jpeg(x, channels=3)` (model/dataloader.py:72-77) split between host and GPU.

Host (C++, `merlot_jpeg_entropy_decode`): markers + Huffman decode -> quantised DCT coefficients.  GPU
(`merlot_jpeg_idct_rgb`): dequantise, inverse DCT, chroma upsampling, colour conversion -- bit for bit libjpeg's default
decoder.  `entropy_decode` returns None for files outside the supported subset (progressive, grayscale, CMYK, 4:2:2 ...):
the caller decodes those with the host library, as before."""
import numpy as np

from .lib import LIB

JPEG_MALFORMED, JPEG_UNSUPPORTED, JPEG_CAPACITY = -20, -21, -22
MAX_IMAGE_PIXELS = 89478485           # PIL.Image.MAX_IMAGE_PIXELS: the host decoder's own decompression-bomb threshold

# mirrors merlot_jpeg_info_t (include/merlot_hip.h), C layout
INFO_DTYPE = np.dtype([('width', np.int32), ('height', np.int32), ('subsampling', np.int32), ('blocks_w', np.int32, (3,)),
                       ('blocks_h', np.int32, (3,)), ('coef_offset', np.int64, (3,)), ('coef_count', np.int64),
                       ('coef_base', np.int64), ('dst_offset', np.int64), ('plane_offset', np.int64),
                       ('quant', np.uint16, (3, 64))], align=True)
assert INFO_DTYPE.itemsize == 480


def entropy_decode(data):
    """bytes of one JPEG file -> (coef int16 [coef_count], info INFO_DTYPE scalar array of shape (1,)), or None when the file is
    not baseline 8-bit YCbCr 4:4:4 / 4:2:0, is larger than MAX_IMAGE_PIXELS, or is malformed / truncated -- the caller then
    decodes it with the host library (PIL), which keeps its own error behaviour for such files."""
    buf = np.frombuffer(bytes(data), np.uint8)
    info = np.zeros(1, INFO_DTYPE)
    dll = LIB.load()
    rc = dll.merlot_jpeg_entropy_decode(buf.ctypes.data, buf.size, info.ctypes.data, None, 0)
    if rc == JPEG_UNSUPPORTED:
        return None
    if rc != 0:
        return None                                       # malformed / truncated: the host library decides (it may still cope)
    # decompression-bomb guard BEFORE the coefficient buffer is sized from the header (a crafted 65535 x 65535 SOF would ask
    # for ~13 GB inside a loader thread): above PIL's own limit the frame goes to the host decoder, which refuses it
    if int(info['width'][0]) * int(info['height'][0]) > MAX_IMAGE_PIXELS:
        return None
    coef = np.empty(int(info['coef_count'][0]), np.int16)
    rc = dll.merlot_jpeg_entropy_decode(buf.ctypes.data, buf.size, info.ctypes.data, coef.ctypes.data, coef.size)
    if rc != 0:                                           # unsupported or malformed scan data: host decoder
        return None
    return coef, info


def info_dict(info):
    i = info[0]
    return {'width': int(i['width']), 'height': int(i['height']), 'subsampling': int(i['subsampling']),
            'blocks_w': [int(v) for v in i['blocks_w']], 'blocks_h': [int(v) for v in i['blocks_h']],
            'coef_offset': [int(v) for v in i['coef_offset']], 'quant': np.asarray(i['quant'])}


def plane_bytes(info):
    i = info[0]
    return int(sum(int(i['blocks_w'][c]) * int(i['blocks_h'][c]) * 64 for c in range(3)))


def decode_batch_gpu(items, device):
    """items: [(coef, info)] -> (flat uint8 device tensor, [byte offset of image i], [(h, w)]): every image decoded to RGB
    [h, w, 3] at its 16-B aligned offset, with ONE upload of the coefficients and one launch pair."""
    import torch
    from . import ops
    n = len(items)
    infos = np.zeros(n, INFO_DTYPE)
    cbase = pbase = dbase = 0
    offs, shapes = [], []
    for k, (coef, info) in enumerate(items):
        infos[k] = info[0]
        infos[k]['coef_base'] = cbase
        infos[k]['plane_offset'] = pbase
        infos[k]['dst_offset'] = dbase
        offs.append(dbase)
        h, w = int(info[0]['height']), int(info[0]['width'])
        shapes.append((h, w))
        cbase += coef.size
        pbase += (plane_bytes(info) + 15) // 16 * 16
        dbase += (h * w * 3 + 15) // 16 * 16
    coef_all = torch.from_numpy(np.concatenate([c for c, _ in items]))
    return ops.jpeg_idct_rgb(coef_all.to(device, non_blocking=True), infos, pbase, dbase), offs, shapes
