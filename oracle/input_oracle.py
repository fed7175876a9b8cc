"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, fp32) of the reference's input pipeline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Two layers:
  * TF primitive kernels the pipeline calls, restated from the published tensorflow==1.15.5 algorithms
    (core/kernels/resize_{bilinear,nearest_neighbor,bicubic,area}_op.cc, legacy scaler, align_corners=True;
    `tf.image.convert_image_dtype`; `tf.image.pad_to_bounding_box`).  TensorFlow is absent (requirements.txt:70):
    PARITY UNPINNED for these kernels.  `oracle/tf_shim.py` uses them as its `tf.image.*` so that the reference's own
    `resize_and_pad` / `lightweight_image_augment` (utils/model_utils.py:758-940) run unmodified on top of them.
  * the reference's Python logic, restated op for op with explicit noise: `resize_and_pad` (utils/model_utils.py:860-940),
    `lightweight_image_augment` restricted to 'brightness,contrast' (:758-842, model/dataloader.py:89-92), the text side
    of `_dataset_parser` (model/dataloader.py:99-126) and `_process_example` (:202-257).  These are pinned by
    tests/test_input_pipeline.py, which runs the reference functions under the shim on the same draws.
"""
import numpy as np

F32 = np.float32


# ---- TF primitive kernels ----------------------------------------------------------------------------------------------
def convert_image_dtype_u8_to_f32(img_u8):
    """tf.image.convert_image_dtype(uint8 -> float32): cast, then multiply by float32(1 / 255)."""
    return img_u8.astype(F32) * F32(1.0 / 255.0)


def _scale(in_size, out_size):
    """CalculateResizeScale with align_corners=True (core/kernels/image_resizer_state.h)."""
    return F32(in_size - 1) / F32(out_size - 1) if out_size > 1 else F32(0.0)


def resize_bilinear(img, out_h, out_w):
    in_h, in_w = img.shape[:2]

    def weights(out_size, in_size):
        pos = np.arange(out_size, dtype=F32) * _scale(in_size, out_size)
        lo_f = np.floor(pos)
        lo = np.maximum(lo_f.astype(np.int64), 0)
        hi = np.minimum(np.ceil(pos).astype(np.int64), in_size - 1)
        return lo, hi, (pos - lo_f).astype(F32)

    y0, y1, ly = weights(out_h, in_h)
    x0, x1, lx = weights(out_w, in_w)
    lx = lx[None, :, None]
    ly = ly[:, None, None]
    tl, tr = img[y0][:, x0], img[y0][:, x1]
    bl, br = img[y1][:, x0], img[y1][:, x1]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    return (top + (bot - top) * ly).astype(F32)


def _roundf(x):
    """C roundf: halves away from zero (inputs are >= 0 here)."""
    return np.floor(x + F32(0.5))


def resize_nearest(img, out_h, out_w):
    in_h, in_w = img.shape[:2]
    ys = np.minimum(_roundf(np.arange(out_h, dtype=F32) * _scale(in_h, out_h)).astype(np.int64), in_h - 1)
    xs = np.minimum(_roundf(np.arange(out_w, dtype=F32) * _scale(in_w, out_w)).astype(np.int64), in_w - 1)
    return img[ys][:, xs].astype(F32)


def _cubic(out_size, in_size):
    """GetWeightsAndIndices<LegacyScaler, false>: Keys cubic with A = -0.75 tabulated at 1/1024."""
    A = F32(-0.75)
    pos = np.arange(out_size, dtype=F32) * _scale(in_size, out_size)
    loc = np.floor(pos)
    delta = (pos - loc).astype(F32)
    off = np.rint(delta * F32(1024.0)).astype(np.int64)               # lrintf: round half to even

    def near(i):
        x = (i.astype(F32) * F32(1.0 / 1024.0)).astype(F32)
        return (((A + F32(2)) * x - (A + F32(3))) * x * x + F32(1)).astype(F32)

    def far(i):
        x = (i.astype(F32) * F32(1.0 / 1024.0) + F32(1.0)).astype(F32)
        return (((A * x - F32(5) * A) * x + F32(8) * A) * x - F32(4) * A).astype(F32)

    loc = loc.astype(np.int64)
    idx = [np.clip(loc + d, 0, in_size - 1) for d in (-1, 0, 1, 2)]
    w = [far(off), near(off), near(1024 - off), far(1024 - off)]
    return idx, w


def resize_bicubic(img, out_h, out_w):
    in_h, in_w = img.shape[:2]
    yi, yw = _cubic(out_h, in_h)
    xi, xw = _cubic(out_w, in_w)
    cols = []
    for k in range(4):                                                   # along y first (cached per x index), then x
        g = [img[yi[r]][:, xi[k]] for r in range(4)]
        cols.append(g[0] * yw[0][:, None, None] + g[1] * yw[1][:, None, None] + g[2] * yw[2][:, None, None]
                    + g[3] * yw[3][:, None, None])
    out = (cols[0] * xw[0][None, :, None] + cols[1] * xw[1][None, :, None] + cols[2] * xw[2][None, :, None]
           + cols[3] * xw[3][None, :, None])
    return out.astype(F32)


def resize_area(img, out_h, out_w):
    in_h, in_w = img.shape[:2]
    hs, ws = _scale(in_h, out_h), _scale(in_w, out_w)

    def spans(out_size, in_size, s):
        res = []
        for o in range(out_size):
            a, b = F32(o) * s, F32(o + 1) * s
            start, end = int(np.floor(a)), int(np.ceil(b))
            idx, wt = [], []
            for i in range(start, end):
                if F32(i) < a:
                    w = s if F32(i + 1) > b else F32(i + 1) - a
                else:
                    w = b - F32(i) if F32(i + 1) > b else F32(1.0)
                idx.append(min(max(i, 0), in_size - 1))
                wt.append(F32(w))
            res.append((idx, wt))
        return res

    ysp, xsp = spans(out_h, in_h, hs), spans(out_w, in_w, ws)
    out = np.zeros((out_h, out_w, img.shape[2]), F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = F32(1.0) / (hs * ws)
        rows = np.zeros((in_h, out_w, img.shape[2]), F32)                # row sums over x, per source row
        for x, (idx, wt) in enumerate(xsp):
            acc = np.zeros((in_h, img.shape[2]), F32)
            for i, w in zip(idx, wt):
                acc = acc + img[:, i] * w
            rows[:, x] = acc
        for y, (idx, wt) in enumerate(ysp):
            acc = np.zeros((out_w, img.shape[2]), F32)
            for i, w in zip(idx, wt):
                acc = acc + rows[i] * w
            out[y] = acc * inv
    return out


RESIZE = {0: resize_bilinear, 1: resize_nearest, 2: resize_bicubic, 3: resize_area}


def resize_images(img, size, method=0):
    """tf.image.resize_images(img [h,w,c] f32, [new_h, new_w], method, align_corners=True)."""
    return RESIZE[int(method)](np.asarray(img, F32), int(size[0]), int(size[1]))


def pad_to_bounding_box(img, off_h, off_w, target_h, target_w):
    out = np.zeros((target_h, target_w, img.shape[2]), img.dtype)
    out[off_h:off_h + img.shape[0], off_w:off_w + img.shape[1]] = img
    return out


# ---- reference logic with explicit noise -------------------------------------------------------------------------------
def resize_geometry(height, width, desired, scale_factor, u_y, u_x):
    """utils/model_utils.py:880-908 (do_random_scale=True), all arithmetic in fp32 as the TF graph does it.
    -> scaled_h, scaled_w, offset_y, offset_x, image_scale"""
    dh, dw = desired
    h, w = F32(height), F32(width)
    scaled_y = int(F32(scale_factor) * F32(dh))                          # tf.cast(f32 -> int32) truncates
    scaled_x = int(F32(scale_factor) * F32(dw))
    image_scale = min(F32(scaled_x) / w, F32(scaled_y) / h)
    image_scale = max(image_scale, F32(64.0) / min(h, w))
    scaled_h = int(h * image_scale)
    scaled_w = int(w * image_scale)
    off_y = int(max(F32(0.0), F32(scaled_h - dh)) * F32(u_y))
    off_x = int(max(F32(0.0), F32(scaled_w - dw)) * F32(u_x))
    return scaled_h, scaled_w, off_y, off_x, image_scale


def resize_and_pad(img_f32, desired, scale_factor, u_y, u_x, method):
    """utils/model_utils.py:860-940 with resize_method='random', do_random_scale=True; -> (image, image_info)."""
    height, width = img_f32.shape[:2]
    sh, sw, oy, ox, image_scale = resize_geometry(height, width, desired, scale_factor, u_y, u_x)
    img = resize_images(img_f32, (sh, sw), method)
    img = img[oy:oy + desired[0], ox:ox + desired[1]]
    img = pad_to_bounding_box(img, 0, 0, desired[0], desired[1])
    info = np.array([min(sh, desired[0]), min(sw, desired[1]), F32(1.0) / image_scale, height, width], F32)
    return img, info


def augment(img, do_augment, kind, factor, fix_selection=False):
    """lightweight_image_augment(allowed_transforms='brightness,contrast') (utils/model_utils.py:758-842):
    brightness `x * f` | contrast `(x - mean) * f + mean`, then clip to [0, 1]; f is [3] (per channel).
    QUIRK kept (:829-830): `[lambda: t(x) for t in transforms]` binds `t` late, so EVERY branch of the switch_case calls
    the LAST transform -- with 'brightness,contrast' the drawn `augment_idx` (kind) is consumed but contrast always
    runs (found by executing the reference under the shim).  fix_selection=True gives the evidently intended choice."""
    if not do_augment:
        return img
    f = np.asarray(factor, F32).reshape(1, 1, 3)
    if fix_selection and kind == 0:
        out = img * f
    else:
        mean = img.mean(axis=(0, 1), keepdims=True, dtype=F32)
        out = (img - mean) * f + mean
    return np.clip(out, F32(0.0), F32(1.0)).astype(F32)


def frame(img_u8, desired, noise):
    """one frame through model/dataloader.py:72-93: decode output (uint8) -> f32 [H, W, 3] before the bf16 cast.
    noise: dict(scale, u_y, u_x, method, do_augment, kind, factor[3])."""
    x = convert_image_dtype_u8_to_f32(img_u8)
    x, _ = resize_and_pad(x, desired, noise['scale'], noise['u_y'], noise['u_x'], noise['method'])
    x = np.where(np.isfinite(x), x, F32(0.0))
    return augment(x, noise['do_augment'], noise['kind'], noise['factor'], noise.get('fix_selection', False))


def text_features(chunks, do_clean, num_chunks, len_per_chunk, START=2, NEXTCAPTION_START=None):
    """model/dataloader.py:99-126: chunks = list of dicts with 'tokenized_cleaned_asr', 'tokenized_raw_asr', 'is_eoc'.
    -> input_ids [num_chunks, len_per_chunk] int32, is_eoc [num_chunks] bool, video_src_ids [num_chunks] int32."""
    key = 'tokenized_cleaned_asr' if do_clean else 'tokenized_raw_asr'
    start = START if do_clean else NEXTCAPTION_START
    rows = [[start] + [int(t) for t in c[key]] for c in chunks]
    width = max(len(r) for r in rows)
    dense = np.zeros((num_chunks, width), np.int32)                     # ragged -> dense, default 0
    for i, r in enumerate(rows):
        dense[i, :len(r)] = r
    ids = np.zeros((num_chunks, len_per_chunk), np.int32)               # pad_to_fixed_size(truncate=True, axis=1)
    w = min(width, len_per_chunk)
    ids[:, :w] = dense[:, :w]
    is_eoc = np.array([bool(c['is_eoc']) for c in chunks[:-1]] + [True])
    delta = np.concatenate([[0], is_eoc[:-1].astype(np.int32)])
    return ids, is_eoc, np.cumsum(delta).astype(np.int32)


def encode_string(b, string_len):
    """utils/model_utils.py:628-637: bytes -> int32 [string_len], zero padded / truncated."""
    raw = np.frombuffer(b, np.uint8).astype(np.int32)[:string_len]
    out = np.zeros(string_len, np.int32)
    out[:raw.shape[0]] = raw
    return out


def shuffle_chunks_index(video_src_ids, u):
    """model/dataloader.py:203-213: the gather index that permutes whole source videos inside each example.
    video_src_ids [bsz, nchunk] int32, u = the random_uniform([bsz, nchunk]) draw."""
    bsz, nchunk = video_src_ids.shape
    mapping = np.argsort(u, -1, kind='stable')
    new_id = np.take_along_axis(mapping, video_src_ids, 1)
    trg = new_id * nchunk + np.arange(nchunk, dtype=np.int64)[None]
    return np.argsort(trg, 1, kind='stable')


def shuffled_idx_img(num_shuffle, u_sel, u_perm, n, shuffle_offset=16):
    """model/dataloader.py:233-246 (same as index_oracle.shuffled_idx; kept here so the pipeline restatement is whole)."""
    do = np.argsort(u_sel, 1, kind='stable') < np.asarray(num_shuffle)[:, None]
    return np.where(do, shuffle_offset + np.argsort(u_perm, 1, kind='stable'), np.arange(n)[None]).astype(np.int32).reshape(-1)
