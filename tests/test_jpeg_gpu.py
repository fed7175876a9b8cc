"""GPU: merlot_jpeg_idct_rgb (dequantise + islow IDCT + fancy h2v2 upsampling + YCbCr->RGB) on a batch of different sizes and
subsamplings, against the numpy oracle AND against PIL / libjpeg-turbo directly -- bit for bit."""
import numpy as np
import pytest
import torch

from test_jpeg import CASES, _encode, _image, _pil

pytestmark = pytest.mark.gpu


def test_batch_decode_equals_libjpeg_bit_for_bit():
    from merlot_amd import jpeg
    from oracle import jpeg_oracle
    files = [_encode(_image(h, w, h * 7 + w), **kw) for h, w, kw in CASES]
    items = [jpeg.entropy_decode(f) for f in files]
    assert all(i is not None for i in items)
    flat, offs, shapes = jpeg.decode_batch_gpu(items, torch.device('cuda', 0))
    flat = flat.cpu().numpy()
    for f, (coef, info), off, (h, w) in zip(files, items, offs, shapes):
        got = flat[off:off + h * w * 3].reshape(h, w, 3)
        assert np.array_equal(got, jpeg_oracle.decode(coef, jpeg.info_dict(info)))
        assert np.array_equal(got, _pil(f))


def test_pipeline_frames_are_identical_with_gpu_jpeg_decode(tmp_path):
    """InputPipeline with data.gpu_jpeg_decode: the same batches as with the host decoder (PIL), bit for bit."""
    from test_input_pipeline import _write_records
    from merlot_amd import input_pipeline as ip
    from merlot_amd.config import NeatConfig
    import common
    for i in range(2):
        _write_records(str(tmp_path / f'train{i:03d}.tfrecord'), 4, 4, seed=50 + i)
    outs = {}
    for gpu in (False, True):
        cfg = NeatConfig.from_dict({
            'data': {'train_file': str(tmp_path / 'train*.tfrecord'), 'val_file': str(tmp_path / 'train000.tfrecord'), 'num_chunks': 4,
                     'chunk_text_len': 32, 'shuffle_buffer_size': 4, 'shuffle_chunks': True, 'augment_prob': 0.5, 'num_threads': 2,
                     'gpu_jpeg_decode': gpu},
            'model': dict(common.tiny_config(), image_size=[64, 64]), 'optimizer': {}, 'device': {'output_dir': str(tmp_path)}})
        it = iter(ip.InputPipeline(cfg, True, batch_size=2, device='cuda:0', seed=3))
        outs[gpu] = [next(it), next(it)]
        torch.cuda.synchronize()
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a['images'], b['images']) and torch.equal(a['input_ids'], b['input_ids'])
