"""Input-pipeline measurements: (1) the batched frame kernel `merlot_image_frames` alone, per resize
method and augment kind, against its algorithmic HBM bytes; (2) records -> features end to end with the host side
(record framing, protobuf, libjpeg decode in a thread pool) in the loop."""
import io
import os
import sys
import tempfile
import time

import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import input_pipeline as ip, ops
from merlot_amd.config import NeatConfig


def main():
    dev = torch.device('cuda', 0)
    N, SH, SW, OH, OW = 512, 384, 512, 224, 224
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (SH, SW, 3)).astype(np.uint8)
    frames = [img] * N
    names = ['bilinear', 'nearest', 'bicubic', 'area']


    def bench(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters


    print(f'(1) merlot_image_frames: {N} frames {SH}x{SW} uint8 -> {OH}x{OW} bf16 (random scale 0.95..1.05, random crop)')
    for method in range(4):
        for aug, aname in ((0, 'no augment'), (2, 'contrast')):
            jobs = np.zeros(N, ip.JOB_DTYPE)
            off = 0
            for i in range(N):
                sh, sw, oy, ox = ip.resize_geometry(SH, SW, (OH, OW), rng.uniform(0.95, 1.05), rng.uniform(), rng.uniform())
                jobs[i] = (off, SH, SW, sh, sw, method, oy, ox, aug, rng.uniform(0.7, 1.3, 3).astype(np.float32), 0.0)
                off += img.nbytes
            src = torch.from_numpy(np.concatenate([f.reshape(-1) for f in frames])).to(dev)
            jh = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy())
            jd = jh.to(dev)
            t = bench(lambda: ops.image_frames(src, jh, jd, N, OH, OW))
            # algorithmic bytes: the source window a frame's crop maps back to is read once, the bf16 frame is written once
            win = np.mean([min(j['scaled_h'], OH) / j['scaled_h'] * min(j['scaled_w'], OW) / j['scaled_w'] for j in jobs])
            byt = N * (SH * SW * 3 * win + OH * OW * 3 * 2)
            print(f'  {names[method]:9s} {aname:10s}: {t:8.1f} us  {N / t * 1e6 / 1e6:6.2f} M frames/s  {byt / t / 1e3:7.1f} GB/s algorithmic '
                  f'({byt / N / 1e3:.0f} KB/frame)', flush=True)

    print('(2) records -> features (16 chunks per example, JPEG 384x512 q90), host threads in the loop')
    from PIL import Image
    tmp = tempfile.mkdtemp(dir=os.environ.get('TMPDIR', '/tmp'))
    yy, xx = np.mgrid[0:SH, 0:SW]
    for f in range(4):
        with ip.TFRecordWriter(os.path.join(tmp, f'train{f:03d}.tfrecord')) as w:
            for e in range(8):
                feats = {}
                for i in range(16):
                    a = np.stack([(yy * 2 + 13 * e + xx) % 256, (xx * 3 + 7 * i) % 256, (yy + xx + f) % 256], -1).astype(np.uint8)
                    b = io.BytesIO()
                    Image.fromarray(a, mode='RGB').save(b, format='JPEG', quality=90)
                    c = {'image/encoded': b.getvalue(), 'image/height': SH, 'image/width': SW, 'youtube_id': b'vid',
                         'tokenized_cleaned_asr': [int(t) for t in rng.randint(100, 50000, 20)],
                         'tokenized_raw_asr': [int(t) for t in rng.randint(100, 50000, 25)], 'is_eoc': 0,
                         'mean_time': np.float32(i), 'chunk_num': i}
                    for k, v in c.items():
                        feats[f'c{i:02d}/{k}'] = v
                w.write(ip.encode_example(feats))
    cfg = NeatConfig.from_dict({
        'data': {'train_file': os.path.join(tmp, 'train*.tfrecord'), 'num_chunks': 16, 'chunk_text_len': 32, 'shuffle_buffer_size': 8,
                 'augment_prob': 0.5, 'num_threads': int(os.environ.get('THREADS', 64))},
        'model': {'image_size': [OH, OW], 'num_chunks_in_group': 4, 'image_shuffle_prob': 0.4, 'use_bfloat16': True},
        'optimizer': {}, 'device': {'output_dir': tmp}})
    for bs in (8, 32):
        pipe = ip.InputPipeline(cfg, True, batch_size=bs, device=dev, seed=0, prefetch=2, num_workers=int(os.environ.get('WORKERS', 0)))
        it = iter(pipe)
        next(it)
        torch.cuda.synchronize()
        t0 = time.time()
        nb = int(os.environ.get('NB', 6))
        for _ in range(nb):
            feats = next(it)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f'  batch {bs:3d} examples ({bs * 16} frames): {nb * bs * 16 / dt:8.0f} frames/s end to end on {os.cpu_count()} host cores '
              f'({pipe.num_threads} host threads, {pipe.num_workers} loader processes)', flush=True)
        pipe.close()


if __name__ == '__main__':
    main()
