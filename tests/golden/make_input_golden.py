"""Runs the reference's UNMODIFIED input-pipeline tensor functions (utils/model_utils.py: resize_and_pad,
lightweight_image_augment, encode_string, pad_to_fixed_size, sample_bernoulli) under oracle/tf_shim.py on synthetic
frames and stores inputs, the random draws they consumed, and their outputs.  BUILD container only.

    python tests/golden/make_input_golden.py               # writes tests/golden/ref_shim_input_pipeline.npz

Pins: the reference's resize geometry (random scale, the 64-pixel floor, crop offsets), the random choice of resize
method through switch/merge, crop + pad, augment selection / factors / clipping -- on top of the shim's `tf.image.*`
primitives, which are the restated TF-1.15 kernels of oracle/input_oracle.py (those stay unpinned, see its header).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
OUT = os.environ.get('MERLOT_GOLDEN_OUT', HERE)
sys.path.insert(0, ROOT)

from oracle import tf_shim                                   # noqa: E402
tf = tf_shim.install()
sys.path.insert(1, REF)
from utils import model_utils as ref_mu                      # noqa: E402  (the reference, unmodified)

from oracle import input_oracle as io_                       # noqa: E402


def synth_frame(h, w, seed):
    """a smooth scene with edges and some texture, uint8 [h, w, 3] (compresses well, exercises every resize kernel)"""
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        fy, fx, ph = r.uniform(0.01, 0.15, 3)
        img[..., c] = 0.5 + 0.35 * np.sin(fy * yy + ph * 10) * np.cos(fx * xx + c)
    for _ in range(4):                                                     # rectangles: hard edges
        y0, x0 = r.randint(0, h - 4), r.randint(0, w - 4)
        y1, x1 = y0 + r.randint(2, max(3, h // 2)), x0 + r.randint(2, max(3, w // 2))
        img[y0:y1, x0:x1] = r.uniform(0, 1, 3)
    img += r.normal(0, 0.02, img.shape)
    return (np.clip(img, 0, 1) * 255).astype(np.uint8)


def run_frame(img_u8, desired, seed, augment_prob, scale_min, scale_max):
    """model/dataloader.py:72-93 for one frame (decode_jpeg output given), reference functions called as written there"""
    tf_shim.STATE.reset(seed=seed)
    img = tf_shim._w(torch.from_numpy(io_.convert_image_dtype_u8_to_f32(img_u8)))
    img, info = ref_mu.resize_and_pad(img, desired, do_random_scale=True, random_scale_max=scale_max,
                                      random_scale_min=scale_min, resize_method='random')
    img = tf_shim._w(torch.where(torch.isfinite(img), img, torch.zeros_like(img)))      # :87
    if augment_prob > 0.0:
        img = ref_mu.lightweight_image_augment(img, augment_prob=augment_prob, allowed_transforms='brightness,contrast')
    draws = list(tf_shim.STATE.draws)
    kinds = [k for k, _ in draws]
    assert kinds[:4] == ['uniform'] * 4, kinds
    noise = {'scale': np.float32(draws[0][1]), 'u_y': np.float32(draws[1][1]), 'u_x': np.float32(draws[2][1]),
             'method': int(draws[3][1]), 'do_augment': False, 'kind': 0, 'factor': np.ones(3, np.float32)}
    if augment_prob > 0.0:
        assert kinds[4:6] == ['categorical'] * 2, kinds
        noise['kind'] = int(draws[4][1].reshape(-1)[0])
        noise['do_augment'] = bool(int(draws[5][1].reshape(-1)[0]))
        if noise['do_augment']:
            assert kinds[6:] == ['uniform'], kinds
            noise['factor'] = draws[6][1].numpy().reshape(3).astype(np.float32)
        else:
            assert len(kinds) == 6
    return img.numpy(), info.numpy(), noise


def main():
    cases = [((97, 131), (64, 64)), ((120, 200), (48, 80)), ((40, 300), (64, 64)), ((200, 90), (64, 48)),
             ((64, 64), (64, 64)), ((33, 47), (64, 64))]
    fx, seen, k = {}, set(), 0
    seed = 0
    while len(seen) < 12 or k < 18:                         # every (method, augment state) pair at least once
        (h, w), desired = cases[seed % len(cases)]
        img = synth_frame(h, w, seed)
        out, info, noise = run_frame(img, desired, seed=1000 + seed, augment_prob=0.7 if seed % 4 else 0.0,
                                     scale_min=0.8 if seed % 3 == 0 else 0.95, scale_max=1.3 if seed % 3 == 0 else 1.05)
        key = (noise['method'], (1 + noise['kind']) if noise['do_augment'] else 0)      # drawn index (unused, see oracle)
        seed += 1
        if key in seen and k >= 12 and len(seen) < 12:
            continue
        seen.add(key)
        mine = io_.frame(img, desired, noise)               # the restatement must reproduce the reference exactly
        err = float(np.abs(mine - out).max())
        geo = io_.resize_geometry(h, w, desired, noise['scale'], noise['u_y'], noise['u_x'])
        print(f'frame {k}: {h}x{w} -> {desired} method {noise["method"]} aug {key[1]} scaled {geo[0]}x{geo[1]} '
              f'offset ({geo[2]},{geo[3]}) restatement max-abs-err {err:.2e}')
        assert err <= 1e-6
        p = f'f{k:02d}/'
        fx[p + 'image_u8'], fx[p + 'desired'], fx[p + 'out'], fx[p + 'info'] = img, np.array(desired), out, info
        for n, v in noise.items():
            fx[p + 'noise/' + n] = np.asarray(v)
        k += 1
    fx['num_frames'] = np.int64(k)

    # encode_string / pad_to_fixed_size (utils/model_utils.py:522-575, 628-637)
    for i, s in enumerate([b'WAaKRUoY6Io', b'', b'x' * 80]):
        fx[f'encode_string/{i}/in'] = np.frombuffer(s, np.uint8)
        fx[f'encode_string/{i}/out'] = ref_mu.encode_string(s, 64).numpy().astype(np.int32)
        assert np.array_equal(fx[f'encode_string/{i}/out'], io_.encode_string(s, 64))
    r = np.random.RandomState(5)
    for i, (rows, width) in enumerate([(4, 7), (4, 32), (4, 45)]):
        a = r.randint(1, 1000, (rows, width)).astype(np.int32)
        out = ref_mu.pad_to_fixed_size(tf_shim._w(torch.from_numpy(a)), pad_value=0, output_shape=[rows, 32], truncate=True,
                                       axis=1).numpy()
        fx[f'pad_to_fixed_size/{i}/in'], fx[f'pad_to_fixed_size/{i}/out'] = a, out.astype(np.int32)
    # sample_bernoulli: which categorical outcome means True (utils/model_utils.py:742-745)
    tf_shim.STATE.reset(seed=3)
    outs = [bool(ref_mu.sample_bernoulli(0.5)) for _ in range(16)]
    drawn = [int(d[1].reshape(-1)[0]) for d in tf_shim.STATE.draws]
    assert outs == [d == 1 for d in drawn]
    fx['sample_bernoulli/draws'], fx['sample_bernoulli/outs'] = np.array(drawn), np.array(outs)
    np.savez_compressed(os.path.join(OUT, 'ref_shim_input_pipeline.npz'), **fx)
    print('wrote ref_shim_input_pipeline.npz', os.path.getsize(os.path.join(OUT, 'ref_shim_input_pipeline.npz')), 'bytes')


if __name__ == '__main__':
    main()
