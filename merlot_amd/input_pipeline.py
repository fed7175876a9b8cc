"""Input pipeline counterpart: TFRecords written by the reference's data/process.py:236-256 ->
the `features` dict `model_fn` consumes (model/modeling.py:685-703), without TensorFlow.

Mirror of model/dataloader.py:
  `_decode_record` (:33-54)        TFRecord framing + tf.train.Example wire format, parsed here (host, Python)
  `_dataset_parser` (:57-126)      per example: JPEG decode (host, PIL/libjpeg), then ONE GPU launch pair per batch for
                                   convert_image_dtype + random-scale resize (4 methods) + crop + pad + brightness /
                                   contrast augment + bf16 cast (csrc/image.hip, `merlot_image_frames`); text side on host
  `input_fn_builder` (:129-280)    file listing / per-rank sharding, shuffle buffer, batching (drop_remainder),
                                   `_process_example`: chunk shuffling, shuffled_idx_img (HIP `merlot_shuffled_idx`),
                                   frame flattening, optional transpose_input
All randomness is drawn on the host from one seeded numpy Generator per pipeline (`draw_*`), so a run is reproducible
and tests can inject the exact draws the reference graph consumed.  There is no CPU fallback for the frame kernels.
"""
import glob
import io
import os
import itertools
import queue
import struct
import threading

import numpy as np
import torch

from . import checkpoint as _ck
from . import ops
from .lib import LIB

START, NEXTCAPTION_START = 2, 5                         # utils/encode/encoder.py:18,21

# model/dataloader.py:20-32: key -> (kind, default); VarLen features default to an empty list
CHUNK_K2F = {
    'image/encoded': ('bytes', b''), 'image/format': ('bytes', b'jpeg'), 'image/key/sha256': ('bytes', b''),
    'image/height': ('int64', 1), 'image/width': ('int64', 1), 'youtube_id': ('bytes', b''),
    'tokenized_cleaned_asr': ('varlen', None), 'tokenized_raw_asr': ('varlen', None), 'is_eoc': ('int64', 1),
    'mean_time': ('float', 1.0), 'chunk_num': ('int64', 1),
}

JOB_DTYPE = np.dtype([('src_offset', '<i8'), ('src_h', '<i4'), ('src_w', '<i4'), ('scaled_h', '<i4'), ('scaled_w', '<i4'),
                      ('method', '<i4'), ('offset_y', '<i4'), ('offset_x', '<i4'), ('aug_kind', '<i4'),
                      ('factor', '<f4', (3,)), ('reserved', '<f4')])
assert JOB_DTYPE.itemsize == 56


class RecordError(ValueError):
    pass


# ---- TFRecord framing (tensorflow/core/lib/io/record_writer.cc): u64 length, masked crc32c(length), data, masked crc32c(data)
def read_tfrecords(path, verify=True):
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise RecordError(f'{path}: truncated record header')
            n, = struct.unpack('<Q', head[:8])
            if verify and _ck.mask_crc(_ck.crc32c(head[:8])) != struct.unpack('<I', head[8:])[0]:
                raise RecordError(f'{path}: corrupt record length')
            body = f.read(n + 4)
            if len(body) != n + 4:
                raise RecordError(f'{path}: truncated record')
            if verify and _ck.mask_crc(_ck.crc32c(body[:n])) != struct.unpack('<I', body[n:])[0]:
                raise RecordError(f'{path}: corrupt record data')
            yield body[:n]


def scan_tfrecords(path, verify=True):
    """(path, offset of the record data, length) of every record of a file: headers only, the data is not read"""
    size = os.path.getsize(path)
    with open(path, 'rb') as f:
        pos = 0
        while pos < size:
            head = f.read(12)
            if len(head) != 12:
                raise RecordError(f'{path}: truncated record header')
            n, = struct.unpack('<Q', head[:8])
            if verify and _ck.mask_crc(_ck.crc32c(head[:8])) != struct.unpack('<I', head[8:])[0]:
                raise RecordError(f'{path}: corrupt record length')
            if pos + 12 + n + 4 > size:
                raise RecordError(f'{path}: truncated record')
            yield path, pos + 12, n
            pos += 12 + n + 4
            f.seek(pos)


def read_record(ref, verify=True):
    """the data of one `scan_tfrecords` entry (checksum verified here, by whoever consumes the record)"""
    if isinstance(ref, (bytes, bytearray, memoryview)):
        return ref
    path, off, n = ref
    with open(path, 'rb') as f:
        f.seek(off)
        body = f.read(n + 4)
    if len(body) != n + 4:
        raise RecordError(f'{path}: truncated record')
    if verify and _ck.mask_crc(_ck.crc32c(body[:n])) != struct.unpack('<I', body[n:])[0]:
        raise RecordError(f'{path}: corrupt record data')
    return memoryview(body)[:n]


class TFRecordWriter(object):
    def __init__(self, path):
        self.f = open(path, 'wb')

    def write(self, data):
        head = struct.pack('<Q', len(data))
        self.f.write(head + struct.pack('<I', _ck.mask_crc(_ck.crc32c(head))))
        self.f.write(data + struct.pack('<I', _ck.mask_crc(_ck.crc32c(data))))

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- tf.train.Example (tensorflow/core/example/{example,feature}.proto) ------------------------------------------------
def parse_example(buf):
    """serialized tf.train.Example -> {key: ('bytes'|'float'|'int64', list)}; bytes values are zero-copy memoryviews of
    `buf` (a frame's JPEG is copied once, by the decoder)."""
    out = {}
    pf = lambda b: _ck._proto_fields(b, copy=False)
    for f, _, features in pf(memoryview(buf)):
        if f != 1:
            continue
        for f2, _, entry in pf(features):                           # map<string, Feature> entries
            if f2 != 1:
                continue
            key, feat = None, b''
            for f3, _, v in pf(entry):
                if f3 == 1:
                    key = bytes(v).decode('utf-8')
                elif f3 == 2:
                    feat = v
            kind, vals = None, []
            for f4, _, lst in pf(feat):
                if f4 == 1:
                    kind, vals = 'bytes', [v for f5, _, v in pf(lst) if f5 == 1]
                elif f4 == 2:
                    kind = 'float'
                    for f5, wt, v in pf(lst):
                        if f5 == 1 and wt == 2:                     # packed
                            vals.extend(struct.unpack(f'<{len(v) // 4}f', v))
                        elif f5 == 1:
                            vals.append(struct.unpack('<f', struct.pack('<I', v))[0])
                elif f4 == 3:
                    kind = 'int64'
                    for f5, wt, v in pf(lst):
                        if f5 == 1 and wt == 2:
                            pos = 0
                            while pos < len(v):
                                x, pos = _ck._get_varint(v, pos)
                                vals.append(_ck._signed(x))
                        elif f5 == 1:
                            vals.append(_ck._signed(v))
            if key is not None:
                out[key] = (kind, vals)
    return out


_KIND = {1: 'bytes', 2: 'float', 3: 'int64', 0: None}


def parse_example_native(buf):
    """`parse_example` through the native one-pass indexer (merlot_example_index in libmerlot_hip.so): same result,
    no per-field Python -- what `decode_record` uses."""
    mv = memoryview(buf)
    n = len(mv)
    arr = np.frombuffer(mv, np.uint8)
    cap_r, cap_i, cap_f = 256, 4096, 256
    while True:
        rows = np.empty((cap_r, 6), np.int64)
        ivals = np.empty(cap_i, np.int64)
        fvals = np.empty(cap_f, np.float32)
        cnt = int(LIB.query('merlot_example_index', arr.ctypes.data if n else None, n, rows.ctypes.data, cap_r,
                            ivals.ctypes.data, cap_i, fvals.ctypes.data, cap_f))
        if cnt == -2:
            cap_r, cap_i, cap_f = cap_r * 4, cap_i * 4, cap_f * 4
            continue
        if cnt < 0:
            raise RecordError('malformed tf.train.Example')
        break
    out = {}
    for ko, kl, kind, a, b, c in rows[:cnt].tolist():
        key = bytes(mv[ko:ko + kl]).decode('utf-8')
        if kind == 1:
            if c > 1:                                       # several bytes values in one feature: rare, general parser
                return parse_example(buf)
            vals = [mv[a:a + b]] if c == 1 else []
        elif kind == 2:
            vals = fvals[a:a + c].tolist()
        elif kind == 3:
            vals = ivals[a:a + c].tolist()
        else:
            vals = []
        out[key] = (_KIND[kind], vals)
    return out


def _ld(field, payload):
    out = bytearray([(field << 3) | 2])
    _ck._put_varint(out, len(payload))
    return bytes(out) + bytes(payload)


def encode_example(features):
    """{key: bytes | [bytes] | int | [int] | float | [float] (np.float32 scalars / lists count as float)} -> bytes,
    with the packed encodings protobuf emits for tf.train.Example."""
    entries = b''
    for key in sorted(features):
        v = features[key]
        v = list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v]
        if v and isinstance(v[0], (bytes, bytearray)):
            lst = _ld(1, b''.join(_ld(1, x) for x in v))
        elif v and isinstance(v[0], (float, np.floating)):
            lst = _ld(2, _ld(1, struct.pack(f'<{len(v)}f', *v)))
        else:
            packed = bytearray()
            for x in v:
                _ck._put_varint(packed, int(x))
            lst = _ld(3, _ld(1, packed) if v else b'')
        entries += _ld(1, _ld(1, key.encode('utf-8')) + _ld(2, lst))
    return _ld(1, entries)


def decode_record(record, num_chunks):
    """model/dataloader.py:33-54: -> list of per-chunk dicts (FixedLen features as scalars with the reference's
    defaults, VarLen as int lists)."""
    ex = parse_example_native(record)
    chunks = []
    for i in range(num_chunks):
        cur = {}
        for k, (kind, default) in CHUNK_K2F.items():
            got = ex.get(f'c{i:02d}/{k}')
            if kind == 'varlen':
                cur[k] = list(got[1]) if got is not None else []
            elif got is None or not got[1]:
                cur[k] = default
            else:
                if len(got[1]) != 1:
                    raise RecordError(f'c{i:02d}/{k}: expected one value, found {len(got[1])}')
                v = got[1][0]
                cur[k] = bytes(v) if (kind == 'bytes' and k != 'image/encoded') else v
        chunks.append(cur)
    return chunks


# ---- per-example parsing ---------------------------------------------------------------------------------------------
def encode_string(b, string_len):
    """utils/model_utils.py:628-637"""
    raw = np.frombuffer(bytes(b), np.uint8).astype(np.int32)[:string_len]
    out = np.zeros(string_len, np.int32)
    out[:raw.shape[0]] = raw
    return out


class JpegCoef(object):
    """a frame whose pixels do not exist on the host: the entropy-decoded DCT coefficients of its JPEG file (`data.gpu_jpeg_decode`);
    dequantisation, IDCT, upsampling and colour conversion run on the GPU at the frame's place in the batch's source buffer."""
    __slots__ = ('coef', 'info', 'shape', 'nbytes')

    def __init__(self, coef, info):
        self.coef, self.info = coef, info
        self.shape = (int(info[0]['height']), int(info[0]['width']), 3)
        self.nbytes = self.shape[0] * self.shape[1] * 3


def decode_jpeg_split(data):
    """`data.gpu_jpeg_decode`: the host does only the Huffman decode (merlot_jpeg_entropy_decode, C++, releases the GIL); files
    outside the GPU decoder's subset (progressive, grayscale, CMYK, 4:2:2) take the host library as before."""
    from . import jpeg
    got = jpeg.entropy_decode(data)
    return decode_jpeg(data) if got is None else JpegCoef(*got)


def decode_jpeg(data):
    """tf.image.decode_jpeg(x, channels=3) -> uint8 [h, w, 3] (host libjpeg through PIL)."""
    from PIL import Image
    img = Image.open(io.BytesIO(data))
    if img.mode != 'RGB':                                  # grayscale / CMYK sources; RGB frames are not copied again
        img = img.convert('RGB')
    return np.asarray(img, dtype=np.uint8)


def draw_example_noise(rng, num_chunks, config):
    """the random draws `_dataset_parser` consumes for one example (utils/model_utils.py:877, 906-907 scale and crop
    offsets, :834 `apply_with_random_selector`, :832-836 augment index + bernoulli, :778/:788 factors) + do_clean; a few
    array draws per example rather than ten scalar ones per frame."""
    lo, hi = config.get('random_scale_min', 0.95), config.get('random_scale_max', 1.05)
    d = 0.8 * 0.4                                        # max_{brightness,contrast}_delta at strength 0.4
    nc = num_chunks
    scale = rng.uniform(lo, hi, nc).astype(np.float32)
    u = rng.uniform(size=(nc, 2)).astype(np.float32)
    method = rng.integers(0, 4, nc)
    augment = config.get('augment_prob', 0.0) > 0.0
    kind = rng.integers(0, 2, nc) if augment else np.zeros(nc, np.int64)        # categorical over the two transforms
    do_aug = (rng.uniform(size=nc) < config['augment_prob']) if augment else np.zeros(nc, bool)
    factor = rng.uniform(1.0 - d, 1.0 + d, (nc, 3)).astype(np.float32) if augment else np.ones((nc, 3), np.float32)
    frames = [{'scale': scale[i], 'u_y': u[i, 0], 'u_x': u[i, 1], 'method': int(method[i]), 'do_augment': bool(do_aug[i]),
               'kind': int(kind[i]), 'factor': factor[i]} for i in range(nc)]
    return {'frames': frames, 'do_clean': bool(rng.uniform() < config.get('clean_asr_prob', 0.5))}


def resize_geometry(height, width, desired, scale_factor, u_y, u_x):
    """utils/model_utils.py:880-908 in fp32, as the graph computes it -> scaled_h, scaled_w, offset_y, offset_x."""
    F = np.float32
    dh, dw = desired
    h, w = F(height), F(width)
    scaled_y, scaled_x = int(F(scale_factor) * F(dh)), int(F(scale_factor) * F(dw))
    image_scale = min(F(scaled_x) / w, F(scaled_y) / h)
    image_scale = max(image_scale, F(64.0) / min(h, w))
    scaled_h, scaled_w = int(h * image_scale), int(w * image_scale)
    off_y = int(max(F(0.0), F(scaled_h - dh)) * F(u_y))
    off_x = int(max(F(0.0), F(scaled_w - dw)) * F(u_x))
    return scaled_h, scaled_w, off_y, off_x


def parse_example_host(record, config, noise):
    """host half of `_dataset_parser` (model/dataloader.py:57-126): everything except the frame arithmetic.
    -> dict(youtube_id, chunk_num, mean_time, input_ids, is_eoc, video_src_ids, frames_u8 [list of HWC uint8], jobs)."""
    num_chunks = config['num_chunks']
    desired = tuple(config['image_size'])
    chunks = decode_record(read_record(record), num_chunks)
    feats = {
        'youtube_id': np.stack([encode_string(c['youtube_id'], 64) for c in chunks], 0),
        'chunk_num': np.array([c['chunk_num'] for c in chunks], np.int32),
        'mean_time': np.array([c['mean_time'] for c in chunks], np.float32),
    }
    frames, jobs = [], np.zeros(num_chunks, JOB_DTYPE)
    for i, c in enumerate(chunks):
        img = decode_jpeg_split(c['image/encoded']) if config.get('gpu_jpeg_decode', False) else decode_jpeg(c['image/encoded'])
        n = noise['frames'][i]
        sh, sw, oy, ox = resize_geometry(img.shape[0], img.shape[1], desired, n['scale'], n['u_y'], n['u_x'])
        # utils/model_utils.py:829-830 binds the transform late: every switch_case branch runs the LAST transform
        # (contrast); the drawn index only matters with `augment_fix_selection: True` (not a reference key).
        kind = n['kind'] if config.get('augment_fix_selection', False) else 1
        jobs[i] = (0, img.shape[0], img.shape[1], sh, sw, n['method'], oy, ox, (1 + kind) if n['do_augment'] else 0,
                   n['factor'], 0.0)
        frames.append(img)
    feats['frames_u8'], feats['jobs'] = frames, jobs
    do_clean = noise['do_clean']
    key = 'tokenized_cleaned_asr' if do_clean else 'tokenized_raw_asr'
    Lc = config.get('chunk_text_len', 32)
    ids = np.zeros((num_chunks, Lc), np.int32)
    for i, c in enumerate(chunks):
        row = ([START if do_clean else NEXTCAPTION_START] + [int(t) for t in c[key]])[:Lc]
        ids[i, :len(row)] = row
    feats['input_ids'] = ids
    feats['is_eoc'] = np.array([bool(c['is_eoc']) for c in chunks[:-1]] + [True])
    feats['video_src_ids'] = np.cumsum(np.concatenate([[0], feats['is_eoc'][:-1].astype(np.int32)])).astype(np.int32)
    return feats


def frame_offsets(frames_u8):
    """16-B aligned byte offsets of the frames in one concatenated source buffer -> (offsets, total bytes)."""
    offs, off = [], 0
    for f in frames_u8:
        offs.append(off)
        off += (f.nbytes + 15) // 16 * 16
    return np.asarray(offs, np.int64), off


class Staging(object):
    """rotating page-locked host buffers for the decoded frames of a batch: worker threads copy into them in parallel,
    the consumer uploads with one non-blocking copy and leaves an event behind that guards the buffer's next use."""

    def __init__(self, n):
        self.bufs, self.events, self.i = [None] * n, [None] * n, 0

    def take(self, nbytes):
        i = self.i
        self.i = (i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
            self.events[i] = None
        if self.bufs[i] is None or self.bufs[i].numel() < nbytes:
            t = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8)
            self.bufs[i] = t.pin_memory() if torch.cuda.is_available() else t
        return i, self.bufs[i]


def pack_frames(frames_u8, staging=None, pool=None):
    """-> (flat uint8 host tensor holding every host-decoded frame at its offset, offsets, staging slot or None, jpeg pack or
    None).  Frames that are JpegCoef leave their bytes of `flat` unwritten; the jpeg pack = (coefficients of all of them
    concatenated [int16 host tensor], merlot_jpeg_info_t table with coef_base / dst_offset / plane_offset set, plane bytes)."""
    offs, total = frame_offsets(frames_u8)
    slot, flat = staging.take(total) if staging is not None else (None, torch.empty(total, dtype=torch.uint8))
    fnp = flat.numpy()
    host = [i for i, f in enumerate(frames_u8) if not isinstance(f, JpegCoef)]

    def put(i):
        f = frames_u8[i]
        fnp[offs[i]:offs[i] + f.nbytes] = f.reshape(-1)

    if pool is not None:
        list(pool.map(put, host))
    else:
        for i in host:
            put(i)
    jp = None
    gpu = [i for i, f in enumerate(frames_u8) if isinstance(f, JpegCoef)]
    if gpu:
        from . import jpeg
        infos = np.zeros(len(gpu), jpeg.INFO_DTYPE)
        cbase = pbase = 0
        for k, i in enumerate(gpu):
            f = frames_u8[i]
            infos[k] = f.info[0]
            infos[k]['coef_base'], infos[k]['plane_offset'], infos[k]['dst_offset'] = cbase, pbase, offs[i]
            cbase += f.coef.size
            pbase += (jpeg.plane_bytes(f.info) + 15) // 16 * 16
        coef = torch.empty(cbase, dtype=torch.int16)
        if torch.cuda.is_available():
            coef = coef.pin_memory()
        cnp, o = coef.numpy(), 0
        for i in gpu:
            c = frames_u8[i].coef
            cnp[o:o + c.size] = c
            o += c.size
        jp = (coef, infos, pbase)
    return flat[:total], offs, slot, jp


def frames_to_device(frames_u8, jobs, out_hw, device, packed=None, staging=None):
    """all frames of a batch -> bf16 [n, H, W, 3] on `device` with one upload and one `merlot_image_frames` call.
    jobs[i] describes output frame i and reads source frame i (or, with `packed=(flat, offsets, slot, jpeg pack)`, the frame at
    jobs['src_offset'][i], already set by the caller)."""
    jobs = jobs.copy()
    if packed is None:
        flat, offs, slot, jp = pack_frames(frames_u8)
        jobs['src_offset'] = offs
    else:
        flat, offs, slot, jp = packed
    src = flat.to(device, non_blocking=True)
    if jp is not None:                                      # the JPEG frames are decoded on the GPU straight into their slots of src
        coef, infos, plane_bytes = jp
        ops.jpeg_idct_rgb(coef.to(device, non_blocking=True), infos, plane_bytes, src.numel(), dst=src)
    if slot is not None and staging is not None and src.is_cuda:
        ev = torch.cuda.Event()
        ev.record()
        staging.events[slot] = ev
    jobs_host = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy())
    return ops.image_frames(src, jobs_host, jobs_host.to(device, non_blocking=True), len(jobs), out_hw[0], out_hw[1])


# ---- batch-level processing (model/dataloader.py:202-268) -------------------------------------------------------------
def draw_batch_noise(rng, batch_size, num_chunks, config):
    n = config['num_chunks_in_group']
    B = batch_size * num_chunks // n
    p = config.get('image_shuffle_prob', 0.5)
    out = {'u_chunks': rng.uniform(size=(batch_size, num_chunks)).astype(np.float32)}
    if p >= 1e-6:
        probs = np.array([1.0 - p, 1e-6] + [p / (n - 1)] * (n - 1), np.float64)
        out['num_shuffle'] = rng.choice(n + 1, size=B, p=probs / probs.sum()).astype(np.int32)
        out['u_sel'] = rng.uniform(size=(B, n)).astype(np.float32)
        out['u_perm'] = rng.uniform(size=(B, n)).astype(np.float32)
    return out


def shuffle_chunks_index(video_src_ids, u):
    """model/dataloader.py:203-213"""
    bsz, nchunk = video_src_ids.shape
    mapping = np.argsort(u, -1, kind='stable')
    new_id = np.take_along_axis(mapping, video_src_ids.astype(np.int64), 1)
    trg = new_id * nchunk + np.arange(nchunk, dtype=np.int64)[None]
    return np.argsort(trg, 1, kind='stable')


def collate(examples, config, noise, device, is_training=True, packed=None, staging=None):
    """`dataset.batch(batch_size)` + `_process_example` -> the features dict of model_fn, on `device`.
    packed = pack_frames(all frames in example order) when the loader thread already staged them."""
    bs = len(examples)
    nc = config['num_chunks']
    H, W = config['image_size']
    host = {k: np.stack([e[k] for e in examples], 0) for k in
            ('youtube_id', 'chunk_num', 'mean_time', 'input_ids', 'is_eoc', 'video_src_ids')}
    jobs = np.concatenate([e['jobs'] for e in examples])
    if packed is None:
        packed = pack_frames([f for e in examples for f in e['frames_u8']])
    jobs['src_offset'] = packed[1]
    order = np.arange(bs * nc).reshape(bs, nc)
    if is_training and config.get('shuffle_chunks', False):
        idx = shuffle_chunks_index(host['video_src_ids'], noise['u_chunks'])
        for k in host:
            host[k] = np.take_along_axis(host[k], idx.reshape(idx.shape + (1,) * (host[k].ndim - 2)), 1)
        order = np.take_along_axis(order, idx, 1)
    flat_order = order.reshape(-1)
    images = frames_to_device(None, jobs[flat_order], (H, W), device, packed=packed, staging=staging)   # output i <- source flat_order[i]
    feats = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in host.items()}
    feats['images'] = images.reshape(bs, nc, H, W, 3)
    if not is_training:                                     # `_process_example` is only mapped when training (:262-263)
        return feats
    n = config['num_chunks_in_group']
    B = bs * nc // n
    if config.get('image_shuffle_prob', 0.5) < 1e-6:
        feats['shuffled_idx_img'] = torch.arange(n, dtype=torch.int32, device=device).repeat(B)
    else:
        feats['shuffled_idx_img'] = ops.shuffled_idx(torch.from_numpy(noise['num_shuffle']).to(device),
                                                     torch.from_numpy(noise['u_sel']).to(device),
                                                     torch.from_numpy(noise['u_perm']).to(device), B, n, 16)
    feats['images'] = images                                # [bs * nc, H, W, 3]
    if 'mask' in noise:
        feats['noise'] = {k: (v.pin_memory() if torch.device(device).type == 'cuda' else v).to(device, non_blocking=True)
                          for k, v in noise['mask'].items()}
    if config.get('transpose_input', False):
        feats['images'] = images.permute(1, 2, 3, 0).contiguous()
    return feats


def _worker_parse(task):
    """loader worker PROCESS (never touches the GPU): one record -> host features; the decoded frames travel back in a
    shared-memory block (name returned), not through the result pipe."""
    from multiprocessing import resource_tracker, shared_memory
    record, config, noise = task
    if config.get('gpu_jpeg_decode', False):
        config = dict(config, gpu_jpeg_decode=False)         # worker processes hand back pixels (shared memory); the split decode is for the thread path
    feats = parse_example_host(record, config, noise)
    frames = feats.pop('frames_u8')
    offs, total = frame_offsets(frames)
    shm = shared_memory.SharedMemory(create=True, size=max(total, 1))
    try:
        resource_tracker.unregister(shm._name, 'shared_memory')      # the parent unlinks it after copying out
    except Exception:
        pass
    buf = np.ndarray((total,), np.uint8, buffer=shm.buf)
    for f, o in zip(frames, offs):
        buf[o:o + f.nbytes] = f.reshape(-1)
    feats['shm'] = (shm.name, [f.shape for f in frames], offs)
    del buf
    shm.close()
    return feats


def _spawn_pool(n):
    """loader processes by `spawn` (fork after HIP initialisation is not safe).  spawn re-imports the parent's
    `__main__` in every child; a training script without an `if __name__ == '__main__':` guard would then run itself
    again in each worker.  The workers only need this module, so `__main__` is hidden while they start."""
    import multiprocessing as mp
    import sys
    if mp.current_process().name != 'MainProcess':
        raise RuntimeError('InputPipeline(num_workers > 0) was started from inside a loader worker process')
    main = sys.modules.get('__main__')
    saved_file = main.__dict__.pop('__file__', None) if main is not None else None
    saved_spec = getattr(main, '__spec__', None) if main is not None else None
    try:
        if main is not None:
            main.__spec__ = None
        pool = mp.get_context('spawn').Pool(n)
        pool.map(_worker_ready, range(n), chunksize=1)        # all workers are up (and imported) before __main__ returns
    finally:
        if main is not None:
            main.__spec__ = saved_spec
            if saved_file is not None:
                main.__file__ = saved_file
    return pool


def _worker_ready(i):
    import time
    time.sleep(0.2)                                           # keeps one fast worker from answering every probe
    return os.getpid()


def _attach_frames(feats):
    """parent side: zero-copy views of a worker's frames + the handle to release afterwards"""
    from multiprocessing import shared_memory
    name, shapes, offs = feats.pop('shm')
    shm = shared_memory.SharedMemory(name=name)
    base = np.ndarray((shm.size,), np.uint8, buffer=shm.buf)
    feats['frames_u8'] = [base[o:o + int(np.prod(sh))].reshape(sh) for sh, o in zip(shapes, offs)]
    return shm


class InputPipeline(object):
    """`input_fn_builder(config, is_training)(params)` as a Python iterator of feature dicts.

    Files: `data.train_file` / `data.val_file` glob; with world_size > 1 each rank reads `files[rank::world_size]` (the
    multi-host branch, :160-166), otherwise all files, shuffled when training.  Records stream through a shuffle buffer
    of `shuffle_buffer_size` (256), are parsed by `num_threads` host threads (JPEG decode releases the GIL) and batched
    with drop_remainder; training repeats forever.  One background thread keeps `prefetch` batches ahead."""

    def __init__(self, config, is_training, batch_size, device, rank=0, world_size=1, seed=0, prefetch=2, num_workers=0):
        """num_workers = 0: records are parsed by `num_threads` threads of this process (libjpeg releases the GIL, the
        protobuf / bookkeeping Python does not: ~125 examples/s per process).  num_workers > 0: that many loader
        PROCESSES parse and decode, handing frames back through shared memory; this process only stages and uploads."""
        self.merged = dict(config.data)
        self.merged.update(config.model)
        self.is_training, self.batch_size, self.device = is_training, batch_size, device
        pattern = config.data['train_file'] if is_training else config.data['val_file']
        files = sorted(f for p in str(pattern).split(',') for f in glob.glob(p))
        if not files:
            raise RecordError(f'no input files match {pattern}')
        if world_size > 1:
            if len(files) // world_size < 1:
                raise RecordError(f'{len(files)} files cannot be sharded over {world_size} ranks')
            files = files[rank::world_size]
        self.files = files
        self.rng = np.random.default_rng([seed, rank])
        # host threads per process: `data.num_threads` (64 in merlot.yaml) bounded by this rank's share of the cores
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world_size))
        self.num_threads = max(1, min(int(config.data.get('num_threads', 64)), (os.cpu_count() or 1) // max(1, local_world)))
        self.buffer_size = int(config.data.get('shuffle_buffer_size', 256))
        self.prefetch = prefetch
        self.staging = Staging(prefetch + 2)
        self.num_workers = int(num_workers)
        self._procs = None
        self._stream = None

    def _records(self):
        """model/dataloader.py:139-150: shuffled file list -> `parallel_interleave(TFRecordDataset, cycle_length =
        min(num_threads, num_files), sloppy)` when training: ONE record at a time from each of cycle_length open files in
        turn (an exhausted file is replaced by the next one of the list), so a batch mixes many files before the shuffle
        buffer even sees it; evaluation reads the files one after another in order."""
        while True:
            files = list(self.files)
            if not self.is_training:
                for f in files:
                    yield from scan_tfrecords(f)          # (path, offset, length): whoever parses the record reads it
                return
            self.rng.shuffle(files)
            pending = iter(files)
            cycle = max(1, min(int(self.merged.get('num_threads', 64)), len(files)))   # the yaml's value, not the core-bounded one
            open_its = [scan_tfrecords(f) for f in itertools.islice(pending, cycle)]
            while open_its:
                nxt = []
                for it in open_its:
                    rec = next(it, None)
                    if rec is None:                          # this file is done: open the next one in its slot
                        f = next(pending, None)
                        if f is None:
                            continue
                        it = scan_tfrecords(f)
                        rec = next(it, None)
                        if rec is None:
                            continue
                    yield rec
                    nxt.append(it)
                open_its = nxt

    def _shuffled(self):
        if not self.is_training:
            yield from self._records()
            return
        buf = []
        for r in self._records():
            if len(buf) < self.buffer_size:
                buf.append(r)
                continue
            i = int(self.rng.integers(0, len(buf)))
            out, buf[i] = buf[i], r
            yield out

    def _host_batches(self):
        from concurrent.futures import ThreadPoolExecutor
        nc = self.merged['num_chunks']
        if self.num_workers > 0 and self._procs is None:
            self._procs = _spawn_pool(self.num_workers)
        with ThreadPoolExecutor(self.num_threads) as pool:
            batch = []
            for rec in self._shuffled():
                batch.append((rec, draw_example_noise(self.rng, nc, self.merged)))
                if len(batch) < self.batch_size:
                    continue
                shms = []
                if self._procs is not None:
                    examples = self._procs.map(_worker_parse, [(r, self.merged, n) for r, n in batch], chunksize=1)
                    shms = [_attach_frames(e) for e in examples]
                else:
                    examples = list(pool.map(lambda a: parse_example_host(a[0], self.merged, a[1]), batch))
                packed = pack_frames([f for e in examples for f in e['frames_u8']], self.staging, pool)
                for e in examples:
                    e['frames_u8'] = None                    # the staged copy is the only one kept
                for shm in shms:
                    shm.close()
                    shm.unlink()
                noise = draw_batch_noise(self.rng, self.batch_size, nc, self.merged)
                if self.is_training and 'vocab_size' in self.merged:
                    # the random draws of mask_inputs (model/modeling.py:445-481), made HERE in the loader thread from the
                    # rank-keyed generator: the training step neither draws on the host nor uploads synchronously
                    from .modeling import draw_mask_noise
                    n = self.merged['num_chunks_in_group']
                    g = torch.Generator().manual_seed(int(self.rng.integers(0, 2 ** 62)))
                    noise['mask'] = draw_mask_noise(self.batch_size * nc // n, n * self.merged.get('chunk_text_len', 32), self.merged,
                                                    self.merged['vocab_size'], g)
                yield examples, noise, packed
                batch = []

    def close(self):
        if self._procs is not None:
            self._procs.terminate()
            self._procs = None

    def __iter__(self):
        q = queue.Queue(self.prefetch)
        stop = object()

        def work():
            try:
                for item in self._host_batches():
                    q.put(item)
                q.put(stop)
            except BaseException as e:                    # surface loader errors in the consumer
                q.put(e)

        threading.Thread(target=work, daemon=True).start()
        while True:
            item = q.get()
            if item is stop:
                return
            if isinstance(item, BaseException):
                raise item
            examples, noise, packed = item
            if self._stream is None and torch.device(self.device).type == 'cuda':
                self._stream = torch.cuda.Stream(self.device)
            if self._stream is None:
                yield collate(examples, self.merged, noise, self.device, self.is_training, packed=packed, staging=self.staging)
                continue
            # upload + frame kernels on a side stream: they overlap the training step still running on the main stream
            main = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self._stream):
                feats = collate(examples, self.merged, noise, self.device, self.is_training, packed=packed, staging=self.staging)
            main.wait_stream(self._stream)
            for v in feats.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(main)
            yield feats
