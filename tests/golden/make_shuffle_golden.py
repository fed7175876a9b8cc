"""Runs the reference's OWN batch-level index work under oracle/tf_shim.py and stores the random draws it
consumed and the integers it produced.  BUILD container only (reads /root/reference).

    python tests/golden/make_shuffle_golden.py             # writes tests/golden/ref_shim_shuffle.npz

What is executed, unmodified:
  * `_process_example`, the closure `input_fn_builder(config, is_training=True)` builds at model/dataloader.py:213-271:
    `input_fn_builder` is called as written, its `input_fn(params)` is called as written on a RECORDING stand-in for
    `tf.data.Dataset` (list_files / repeat / apply / shuffle / map / batch / prefetch return the stand-in and `map`
    remembers its callable) and the last mapped callable -- the reference's own `_process_example` -- is then applied to
    synthetic `features`.  That runs :215-222 (`shuffle_chunks` re-ordering) and :224-257 (`shuffled_idx_img`) exactly
    as the reference wrote them, including `shuffle_offset = 16`.
  * the three `video_src_ids` statements at model/dataloader.py:121-125, lifted out of `_dataset_parser` by line number
    (AST nodes compiled as they are -- the rest of that function decodes JPEGs and tf.Examples).
The stand-ins added HERE (not in the shim) are control plane only: tf.data / tf.gfile / tf.io feature descriptors /
tf.contrib.slim.  The arithmetic primitives the executed lines use (argsort, random_uniform, random.categorical, gather
with batch_dims, where, cumsum, ...) are the shim's; their tie rules are anchored separately in tests/test_shim_anchors.py.
"""
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
OUT = os.environ.get('MERLOT_GOLDEN_OUT', HERE)
sys.path.insert(0, ROOT)

from oracle import tf_shim                                   # noqa: E402
tf = tf_shim.install()


class RecordingDataset(object):
    """control-plane stand-in for tf.data.Dataset: every transformation returns self; map() records its callable."""
    mapped = []

    def _same(self, *a, **k):
        return self
    shard = repeat = apply = shuffle = batch = unbatch = prefetch = _same

    def map(self, fn, num_parallel_calls=None):
        RecordingDataset.mapped.append(fn)
        return self

    @staticmethod
    def list_files(*a, **k):
        return RecordingDataset()


def add_control_plane_stubs():
    tf.string = 'string'
    tf.io.FixedLenFeature = lambda *a, **k: ('fixed', a, k)
    tf.io.VarLenFeature = lambda *a, **k: ('varlen', a, k)
    tf.contrib.slim = types.SimpleNamespace(tfexample_decoder=types.SimpleNamespace())
    tf.gfile = types.SimpleNamespace(Glob=lambda pattern: ['train000.tfrecord', 'train001.tfrecord'])
    tf.data = types.SimpleNamespace(
        Dataset=RecordingDataset, TFRecordDataset=lambda f: RecordingDataset(),
        experimental=types.SimpleNamespace(parallel_interleave=lambda *a, **k: None, AUTOTUNE=-1))
    tf.compat.v1.random = types.SimpleNamespace(set_random_seed=lambda s: None)
    if not hasattr(tf, 'squeeze'):
        tf.squeeze = lambda x, axis=None: tf_shim._w(tf_shim._t(x).squeeze(int(axis)) if axis is not None else tf_shim._t(x).squeeze())
    tf.cumsum = lambda x, axis=0: tf_shim._w(torch.cumsum(tf_shim._t(x), dim=int(axis)).to(tf_shim._t(x).dtype))


add_control_plane_stubs()
sys.path.insert(1, REF)
from model import dataloader as ref_dl                       # noqa: E402  (the reference, unmodified)
from utils.neat_config import NeatConfig as RefNeatConfig    # noqa: E402


def reference_process_example(model_cfg, data_cfg):
    config = RefNeatConfig.__new__(RefNeatConfig)            # the YAML loader is not what is under test
    config.model, config.data = dict(model_cfg), dict(data_cfg)
    RecordingDataset.mapped.clear()
    input_fn = ref_dl.input_fn_builder(config, is_training=True)
    input_fn({'batch_size': 2})
    fn = RecordingDataset.mapped[-1]
    assert fn.__name__ == '_process_example', fn
    return fn


def lifted_video_src_ids():
    """statements of model/dataloader.py:121-125 compiled as they stand -> f(chunk_list) -> (is_eoc, video_src_ids)"""
    src = open(os.path.join(REF, 'model', 'dataloader.py')).read()
    tree = ast.parse(src)
    parser = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == '_dataset_parser')
    stmts = sorted((n for n in ast.walk(parser) if isinstance(n, ast.Assign) and 121 <= n.lineno <= 125), key=lambda n: n.lineno)
    assert len(stmts) == 3, [ast.unparse(s) for s in stmts]
    text = [ast.unparse(s) for s in stmts]
    assert 'is_eoc' in text[0] and 'chunk_id_delta' in text[1] and 'cumsum' in text[2], text
    fn = ast.FunctionDef(name='lifted', args=ast.arguments(posonlyargs=[], args=[ast.arg('chunk_list'), ast.arg('features')],
                                                           kwonlyargs=[], kw_defaults=[], defaults=[]),
                         body=stmts + [ast.Return(ast.Name('features', ast.Load()))], decorator_list=[])
    mod = ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[]))
    ns = {'tf': tf}
    exec(compile(mod, 'dataloader.py:121-125', 'exec'), ns)
    return ns['lifted']


def main():
    fx = {}
    k = 0
    # ---- shuffled_idx_img, :224-257 -------------------------------------------------------------------------------
    # Found by executing it: with `shuffle_chunks: False` and image_shuffle_prob >= 1e-6 the reference raises
    # UnboundLocalError at :245 (its log line formats `k`, the loop variable of the shuffle_chunks branch at :219) -- both
    # released configs set `shuffle_chunks: True`, so the shuffled-index branch only ever runs AFTER the chunk re-ordering
    # has consumed one random_uniform([bsz, nchunk]).  The cases below run it that way and keep that draw.
    for n, nc, bs, p, seed in [(4, 16, 2, 0.4, 1), (4, 16, 3, 0.4, 2), (5, 5, 4, 0.5, 3), (4, 4, 8, 0.9, 4), (16, 16, 2, 0.4, 5),
                              (4, 16, 2, 0.0, 6), (2, 4, 6, 0.5, 7)]:
        fn = reference_process_example({'num_chunks_in_group': n, 'image_shuffle_prob': p},
                                       {'num_chunks': nc, 'train_file': 'x', 'shuffle_chunks': True})
        tf_shim.STATE.reset(seed=seed)
        feats = {key: tf_shim._w(torch.zeros(bs, nc, dtype=torch.int32)) for key in
                 ['youtube_id', 'chunk_num', 'mean_time', 'is_eoc', 'video_src_ids']}
        feats['images'] = tf_shim._w(torch.zeros(bs, nc, 2, 2, 3))
        feats['input_ids'] = tf_shim._w(torch.zeros(bs, nc, 32, dtype=torch.int32))
        out, _ = fn(dict(feats), {})
        draws = tf_shim.STATE.draws
        B = bs * nc // n
        sidx = out['shuffled_idx_img'].numpy().astype(np.int32)
        assert sidx.shape == (B * n,)
        pre = f'shuffle/{k}/'
        fx[pre + 'n'], fx[pre + 'B'], fx[pre + 'p'] = np.int32(n), np.int32(B), np.float64(p)
        fx[pre + 'out'] = sidx
        if p >= 1e-6:
            assert [d[0] for d in draws] == ['uniform', 'categorical', 'uniform', 'uniform'], [d[0] for d in draws]
            fx[pre + 'num_shuffle'] = draws[1][1].numpy().reshape(-1).astype(np.int32)
            fx[pre + 'u_select'] = draws[2][1].numpy().astype(np.float32)
            fx[pre + 'u_perm'] = draws[3][1].numpy().astype(np.float32)
        else:
            assert [d[0] for d in draws] == ['uniform']
        assert tuple(out['images'].shape) == (bs * nc, 2, 2, 3)        # :259-260
        k += 1
    # ... and the failure itself, so the restatement's caller can mirror it
    fn = reference_process_example({'num_chunks_in_group': 4, 'image_shuffle_prob': 0.4}, {'num_chunks': 4, 'train_file': 'x'})
    try:
        fn({'images': tf_shim._w(torch.zeros(1, 4, 2, 2, 3)), 'input_ids': tf_shim._w(torch.zeros(1, 4, 32, dtype=torch.int32)),
            'video_src_ids': tf_shim._w(torch.zeros(1, 4, dtype=torch.int32))}, {})
        raised = False
    except UnboundLocalError:
        raised = True
    fx['shuffle/unbound_k_without_shuffle_chunks'] = np.bool_(raised)
    fx['shuffle/count'] = np.int32(k)

    # ---- shuffle_chunks re-ordering, :215-222 -----------------------------------------------------------------------
    fn = reference_process_example({'num_chunks_in_group': 4, 'image_shuffle_prob': 0.0},
                                   {'num_chunks': 8, 'train_file': 'x', 'shuffle_chunks': True})
    tf_shim.STATE.reset(seed=11)
    bs, nc = 3, 8
    vsrc = torch.tensor([[0, 0, 0, 1, 1, 2, 2, 2], [0, 0, 0, 0, 0, 0, 0, 0], [0, 1, 2, 3, 3, 3, 4, 4]], dtype=torch.int32)
    feats = {key: tf_shim._w(torch.arange(bs * nc, dtype=torch.int32).reshape(bs, nc) + 100 * i)
             for i, key in enumerate(['youtube_id', 'chunk_num', 'mean_time', 'is_eoc'])}
    feats['images'] = tf_shim._w(torch.arange(bs * nc, dtype=torch.float32).reshape(bs, nc, 1, 1, 1).expand(bs, nc, 1, 1, 3).clone())
    feats['input_ids'] = tf_shim._w(torch.arange(bs * nc * 4, dtype=torch.int32).reshape(bs, nc, 4))
    feats['video_src_ids'] = tf_shim._w(vsrc)
    out, _ = fn(dict(feats), {})
    draws = tf_shim.STATE.draws
    assert [d[0] for d in draws] == ['uniform']
    fx['chunks/video_src_ids'] = vsrc.numpy()
    fx['chunks/u'] = draws[0][1].numpy().astype(np.float32)
    fx['chunks/chunk_num_in'] = feats['chunk_num'].numpy().astype(np.int32)
    fx['chunks/chunk_num_out'] = out['chunk_num'].numpy().astype(np.int32)
    fx['chunks/video_src_ids_out'] = out['video_src_ids'].numpy().astype(np.int32)
    fx['chunks/input_ids_out'] = out['input_ids'].numpy().astype(np.int32)

    # ---- video_src_ids, :121-125 ----------------------------------------------------------------------------------------
    lifted = lifted_video_src_ids()
    r = np.random.RandomState(0)
    rows_in, rows_out, rows_eoc = [], [], []
    for i in range(12):
        nc = 16
        eoc = (r.uniform(size=nc) < (0.0 if i == 0 else 0.25)).astype(np.int64)
        if i == 1:
            eoc[:] = 1
        chunk_list = [{'is_eoc': tf_shim._w(torch.tensor(int(e), dtype=torch.int64))} for e in eoc]
        f = lifted(chunk_list, {})
        rows_in.append(eoc)
        rows_eoc.append(f['is_eoc'].numpy().astype(np.uint8))
        rows_out.append(f['video_src_ids'].numpy().astype(np.int32))
    fx['vsrc/is_eoc_records'] = np.stack(rows_in)
    fx['vsrc/is_eoc'] = np.stack(rows_eoc)
    fx['vsrc/video_src_ids'] = np.stack(rows_out)

    # the restatement must reproduce all of it bit for bit before the fixture is written
    from oracle import index_oracle as ix
    for i in range(int(fx['shuffle/count'])):
        pre = f'shuffle/{i}/'
        n, B, p = int(fx[pre + 'n']), int(fx[pre + 'B']), float(fx[pre + 'p'])
        mine = ix.shuffled_idx_img(B, n, p, fx.get(pre + 'num_shuffle'), fx.get(pre + 'u_select'), fx.get(pre + 'u_perm'))
        assert np.array_equal(mine, fx[pre + 'out']), i
    for a, b in zip(fx['vsrc/is_eoc'], fx['vsrc/video_src_ids']):
        assert np.array_equal(ix.video_src_ids(a), b)
    from oracle import input_oracle as io_
    mine = io_.shuffle_chunks_index(fx['chunks/video_src_ids'], fx['chunks/u'])
    assert np.array_equal(np.take_along_axis(fx['chunks/chunk_num_in'], mine, 1), fx['chunks/chunk_num_out'])
    np.savez_compressed(os.path.join(OUT, 'ref_shim_shuffle.npz'), **fx)
    print('wrote ref_shim_shuffle.npz', os.path.getsize(os.path.join(OUT, 'ref_shim_shuffle.npz')), 'bytes;',
          int(fx['shuffle/count']), 'shuffle cases')


if __name__ == '__main__':
    main()
