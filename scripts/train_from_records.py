"""End to end on one GPU: TFRecords (as data/process.py writes them) -> InputPipeline (loader processes, shared-memory
frames, batched HIP frame kernel) -> Trainer.step (fwd + bwd + AdamW) at the bench configuration (512 segments/step).
Reports segments/s of the fed training loop next to the synthetic-input figure of bench.py."""
import io
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_records(tmp, files=4, per_file=8, chunks=16, h=384, w=512):
    from PIL import Image
    from merlot_amd import input_pipeline as ip
    rng = np.random.RandomState(0)
    yy, xx = np.mgrid[0:h, 0:w]
    for f in range(files):
        with ip.TFRecordWriter(os.path.join(tmp, f'train{f:03d}.tfrecord')) as wr:
            for e in range(per_file):
                feats = {}
                for i in range(chunks):
                    a = np.stack([(yy * 2 + 13 * e + xx) % 256, (xx * 3 + 7 * i) % 256, (yy + xx + 31 * f) % 256], -1).astype(np.uint8)
                    b = io.BytesIO()
                    Image.fromarray(a, mode='RGB').save(b, format='JPEG', quality=90)
                    n = int(rng.randint(8, 31))
                    c = {'image/encoded': b.getvalue(), 'image/height': h, 'image/width': w, 'youtube_id': f'v{f}_{e}'.encode(),
                         'tokenized_cleaned_asr': [int(t) for t in rng.randint(100, 50353, n)],
                         'tokenized_raw_asr': [int(t) for t in rng.randint(100, 50353, n)], 'is_eoc': int(i % 8 == 7),
                         'mean_time': np.float32(i * 3.5), 'chunk_num': i}
                    for k, v in c.items():
                        feats[f'c{i:02d}/{k}'] = v
                wr.write(ip.encode_example(feats))


def main():
    from merlot_amd import NeatConfig, input_pipeline as ip
    from merlot_amd.train import Trainer
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    tmp = tempfile.mkdtemp(dir=os.environ.get('TMPDIR', '/tmp'))
    t0 = time.time()
    write_records(tmp)
    print(f'wrote 32 examples x 16 frames in {time.time() - t0:.1f} s', flush=True)
    config = NeatConfig.from_yaml(os.path.join(ROOT, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
    config.data.update(train_file=os.path.join(tmp, 'train*.tfrecord'), shuffle_buffer_size=16, augment_prob=0.5)
    examples = int(os.environ.get('EXAMPLES', 32))
    workers = int(os.environ.get('WORKERS', 32))
    trainer = Trainer(config, dev, None, seed=0)
    pipe = ip.InputPipeline(config, True, batch_size=examples, device=dev, seed=0, prefetch=3, num_workers=workers)
    it = iter(pipe)
    seg = examples * config.data['num_chunks']
    for _ in range(3):
        out = trainer.step(next(it))
    torch.cuda.synchronize()
    steps = int(os.environ.get('STEPS', 8))
    t0 = time.time()
    for _ in range(steps):
        out = trainer.step(next(it))
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'records -> training step: {steps} steps of {seg} segments in {dt * 1e3:.0f} ms = {steps * seg / dt:.0f} segments/s '
          f'({dt / steps * 1e3:.1f} ms/step, {workers} loader processes, final loss {float(out["loss"]):.3f})', flush=True)
    pipe.close()


if __name__ == '__main__':
    main()
