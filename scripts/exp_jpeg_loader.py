"""Loader-only rate (records -> features on the GPU, no training step) of the THREAD path (num_workers = 0) with the host JPEG
decoder (PIL / libjpeg-turbo) and with `data.gpu_jpeg_decode` (host: Huffman decode in C++; GPU: IDCT + upsampling + colour),
plus the kernel time of the GPU half alone.  384 x 512 source frames, quality 90, 4:2:0, as scripts/train_from_records.py."""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from train_from_records import write_records  # noqa: E402


def main():
    from merlot_amd import NeatConfig, input_pipeline as ip, jpeg
    dev = torch.device('cuda', 0)
    tmp = tempfile.mkdtemp(dir=os.environ.get('TMPDIR', '/tmp'))
    write_records(tmp, files=4, per_file=8)
    for gpu in (False, True, False, True):
        config = NeatConfig.from_yaml(os.path.join(ROOT, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
        config.data.update(train_file=os.path.join(tmp, 'train*.tfrecord'), shuffle_buffer_size=16, augment_prob=0.5, gpu_jpeg_decode=gpu)
        pipe = ip.InputPipeline(config, True, batch_size=16, device=dev, seed=0, prefetch=3, num_workers=0)
        it = iter(pipe)
        for _ in range(2):
            next(it)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 6
        for _ in range(n):
            next(it)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f'gpu_jpeg_decode={gpu!s:5s}: {n * 16 * 16 / dt:7.0f} frames/s ({n * 16 / dt:5.1f} examples/s) on {pipe.num_threads} host threads', flush=True)
    # the GPU half alone
    import io
    import numpy as np
    from PIL import Image
    yy, xx = np.mgrid[0:384, 0:512]
    files = []
    for i in range(256):
        a = np.stack([(yy * 2 + 13 * i + xx) % 256, (xx * 3 + 7 * i) % 256, (yy + xx) % 256], -1).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(a).save(b, format='JPEG', quality=90)
        files.append(b.getvalue())
    t0 = time.time()
    items = [jpeg.entropy_decode(f) for f in files]
    t_h = time.time() - t0
    t0 = time.time()
    for f in files:
        Image.open(io.BytesIO(f)).convert('RGB').load()
    t_p = time.time() - t0
    jpeg.decode_batch_gpu(items, dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    jpeg.decode_batch_gpu(items, dev)
    e1.record()
    torch.cuda.synchronize()
    mb = 256 * 384 * 512 * 3 / 1e6
    print(f'256 frames 384x512: host Huffman decode {t_h * 1e3 / 256:.2f} ms/frame (1 thread) vs full PIL decode {t_p * 1e3 / 256:.2f} ms/frame; '
          f'GPU half incl. upload {e0.elapsed_time(e1):.2f} ms per 256 frames ({mb / e0.elapsed_time(e1):.1f} GB/s of RGB)', flush=True)


if __name__ == '__main__':
    main()
