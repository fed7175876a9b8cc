"""GPU: frame preprocessing kernel (csrc/image.hip) against the reference-run fixture and the oracle, and the whole
record -> features -> train-step path."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'ref_shim_input_pipeline.npz')
BF16 = torch.bfloat16


def _run(frames, jobs, out_hw):
    from merlot_amd import input_pipeline as ip
    return ip.frames_to_device(frames, jobs, out_hw, torch.device('cuda', 0)).cpu()


def _job(ip, img, desired, n, kind=None):
    sh, sw, oy, ox = ip.resize_geometry(img.shape[0], img.shape[1], desired, n['scale'], n['u_y'], n['u_x'])
    j = np.zeros(1, ip.JOB_DTYPE)
    k = (1 + (int(n['kind']) if kind is None else kind)) if n['do_augment'] else 0
    j[0] = (0, img.shape[0], img.shape[1], sh, sw, int(n['method']), oy, ox, k, np.asarray(n['factor'], np.float32), 0.0)
    return j


def _check(got, want_f32, exact):
    want = torch.from_numpy(want_f32).to(BF16)
    if exact:
        assert torch.equal(got, want)
        return
    d = (got.float() - want.float()).abs()
    assert float(d.max()) <= 2.0 ** -8                                   # one bf16 ulp below 1.0
    assert float((d > 0).float().mean()) < 5e-3


def test_frames_match_the_reference_run():
    """bit-exact against bf16(reference output) for un-augmented frames (every product and sum rounded like the TF
    kernel); contrast frames depend on the fp32 mean's summation order: <= 1 bf16 ulp on < 0.5 % of the values."""
    from merlot_amd import input_pipeline as ip
    fx = np.load(GOLD)
    frames, jobs, wants, augs, shapes = [], [], [], [], []
    for k in range(int(fx['num_frames'])):
        p = f'f{k:02d}/'
        n = {q: fx[p + 'noise/' + q] for q in ('scale', 'u_y', 'u_x', 'method', 'do_augment', 'kind', 'factor')}
        img, desired = fx[p + 'image_u8'], tuple(int(v) for v in fx[p + 'desired'])
        got = _run([img], _job(ip, img, desired, n, kind=1), desired)[0]   # reference quirk: contrast always
        _check(got, fx[p + 'out'], exact=not bool(n['do_augment']))
        if desired == (64, 64):
            frames.append(img); jobs.append(_job(ip, img, desired, n, kind=1)); wants.append(fx[p + 'out']); augs.append(bool(n['do_augment']))
    # the same frames as ONE batched launch (distinct sources, methods and augment kinds in one job table)
    got = _run(frames, np.concatenate(jobs), (64, 64))
    for i in range(len(frames)):
        _check(got[i], wants[i], exact=not augs[i])


@pytest.mark.parametrize('size,desired', [((384, 512), (224, 224)), ((360, 640), (192, 352)), ((50, 70), (224, 224))])
def test_frames_match_oracle_at_model_sizes(size, desired):
    from merlot_amd import input_pipeline as ip
    from oracle import input_oracle as io_
    r = np.random.RandomState(size[0])
    yy, xx = np.mgrid[0:size[0], 0:size[1]]
    img = np.stack([(yy * 2 + xx) % 256, (xx * 3) % 256, (yy * 5 + 7 * xx) % 256], -1).astype(np.uint8)
    img[10:30, 20:60] = r.randint(0, 256, (20, 40, 3))
    frames, jobs, wants, exact = [], [], [], []
    for method in range(4):
        for aug in (0, 1, 2):
            n = {'scale': np.float32(r.uniform(0.9, 1.2)), 'u_y': np.float32(r.uniform()), 'u_x': np.float32(r.uniform()),
                 'method': method, 'do_augment': aug > 0, 'kind': max(aug - 1, 0), 'factor': r.uniform(0.68, 1.32, 3).astype(np.float32),
                 'fix_selection': True}
            frames.append(img); jobs.append(_job(ip, img, desired, n)); wants.append(io_.frame(img, desired, n)); exact.append(aug != 2)
    got = _run(frames, np.concatenate(jobs), desired)
    for i in range(len(frames)):
        _check(got[i], wants[i], exact=exact[i])


def test_argument_validation():
    from merlot_amd import input_pipeline as ip, ops
    from merlot_amd.lib import MerlotHipError
    img = np.zeros((8, 8, 3), np.uint8)
    n = {'scale': 1.0, 'u_y': 0.0, 'u_x': 0.0, 'method': 0, 'do_augment': False, 'kind': 0, 'factor': np.ones(3, np.float32)}
    j = _job(ip, img, (16, 16), n)
    j['method'] = 7
    with pytest.raises(MerlotHipError, match='resize method'):
        _run([img], j, (16, 16))
    j = _job(ip, img, (16, 16), n)
    j['src_h'] = 100
    with pytest.raises(MerlotHipError, match='outside the source buffer'):
        _run([img], j, (16, 16))


def test_records_to_train_step(tmp_path):
    """TFRecords (as data/process.py writes them) -> InputPipeline -> features -> one optimizer step on the HIP path."""
    sys.path.insert(0, os.path.dirname(__file__))
    from common import tiny_config
    from test_input_pipeline import _write_records
    from merlot_amd import input_pipeline as ip
    from merlot_amd.config import NeatConfig
    from merlot_amd.train import Trainer
    for i in range(2):
        _write_records(str(tmp_path / f'train{i:03d}.tfrecord'), 4, 4, seed=20 + i)
    cfg = tiny_config()
    config = NeatConfig.from_dict({
        'data': {'train_file': str(tmp_path / 'train*.tfrecord'), 'num_chunks': 4, 'chunk_text_len': 32, 'shuffle_buffer_size': 4,
                 'augment_prob': 0.5, 'num_threads': 2},
        'model': cfg,
        'optimizer': {'type': 'adam_optimizer', 'learning_rate': 1e-4, 'num_train_steps': 100, 'num_warmup_steps': 10,
                      'weight_decay_rate': 0.1, 'beta_2': 0.98, 'use_bfloat16_adam': True},
        'device': {'output_dir': str(tmp_path / 'out')}})
    dev = torch.device('cuda', 0)
    trainer = Trainer(config, dev, seed=0)
    it = iter(ip.InputPipeline(config, True, batch_size=2, device=dev, seed=0))
    losses = []
    for _ in range(3):
        feats = next(it)
        assert feats['images'].shape == (8,) + tuple(cfg['image_size']) + (3,) and feats['images'].is_cuda
        losses.append(float(trainer.step(feats)['loss']))
    assert all(np.isfinite(losses))
    # and the Estimator-style checkpoint round trip on the GPU-resident state
    prefix = trainer.save()
    w0, m0 = trainer.store.master.clone(), trainer.opt.m.clone()
    trainer.store.master.zero_(); trainer.opt.m.zero_()
    assert trainer.restore() == 3 and torch.equal(trainer.store.master, w0) and torch.equal(trainer.opt.m, m0)
    assert os.path.exists(prefix + '.index')
