"""CPU: input-pipeline counterpart: record / Example wire formats, the oracle against what the
reference's own functions computed under the shim (tests/golden/ref_shim_input_pipeline.npz), and the host pipeline with
the two HIP entry points emulated."""
import io
import itertools
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

from merlot_amd import input_pipeline as ip
from oracle import index_oracle, input_oracle as io_

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'ref_shim_input_pipeline.npz')


def _noise(fx, p):
    return {k: fx[p + 'noise/' + k] for k in ('scale', 'u_y', 'u_x', 'method', 'do_augment', 'kind', 'factor')}


def test_oracle_reproduces_reference_frames():
    fx = np.load(GOLD)
    combos = set()
    for k in range(int(fx['num_frames'])):
        p = f'f{k:02d}/'
        n = _noise(fx, p)
        img, desired = fx[p + 'image_u8'], tuple(int(v) for v in fx[p + 'desired'])
        out = io_.frame(img, desired, n)
        assert out.shape == desired + (3,) and out.dtype == np.float32
        assert np.abs(out - fx[p + 'out']).max() <= 1e-6, k              # fp32 reduction order of the contrast mean
        _, info = io_.resize_and_pad(io_.convert_image_dtype_u8_to_f32(img), desired, n['scale'], n['u_y'], n['u_x'], n['method'])
        assert np.allclose(info, fx[p + 'info'], rtol=1e-6)
        geo = ip.resize_geometry(img.shape[0], img.shape[1], desired, n['scale'], n['u_y'], n['u_x'])
        assert geo == io_.resize_geometry(img.shape[0], img.shape[1], desired, n['scale'], n['u_y'], n['u_x'])[:4]
        combos.add((int(n['method']), bool(n['do_augment'])))
    assert combos == {(m, a) for m in range(4) for a in (False, True)}
    # the late-binding quirk (utils/model_utils.py:829-830): a drawn index 0 ("brightness") still ran contrast
    quirk = [k for k in range(int(fx['num_frames'])) if fx[f'f{k:02d}/noise/do_augment'] and fx[f'f{k:02d}/noise/kind'] == 0]
    assert quirk
    p = f'f{quirk[0]:02d}/'
    fixed = io_.frame(fx[p + 'image_u8'], tuple(int(v) for v in fx[p + 'desired']), dict(_noise(fx, p), fix_selection=True))
    assert np.abs(fixed - fx[p + 'out']).max() > 1e-3


def test_small_reference_functions():
    fx = np.load(GOLD)
    for i in range(3):
        s = fx[f'encode_string/{i}/in'].tobytes()
        assert np.array_equal(io_.encode_string(s, 64), fx[f'encode_string/{i}/out'])
        assert np.array_equal(ip.encode_string(s, 64), fx[f'encode_string/{i}/out'])
        a, want = fx[f'pad_to_fixed_size/{i}/in'], fx[f'pad_to_fixed_size/{i}/out']
        chunks = [{'tokenized_cleaned_asr': list(r[1:]), 'tokenized_raw_asr': [], 'is_eoc': 0} for r in a]
        ids, _, _ = io_.text_features(chunks, True, len(a), 32, START=int(a[0, 0]))
        ids[:, 0] = a[:, 0]                                               # the fixture rows carry their own first token
        assert np.array_equal(ids, want)
    assert np.array_equal(fx['sample_bernoulli/outs'], fx['sample_bernoulli/draws'] == 1)


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='needs the reference checkout (build container only)')
def test_fixture_regenerates_from_the_reference(tmp_path):
    env = dict(os.environ, MERLOT_GOLDEN_OUT=str(tmp_path))
    script = os.path.join(os.path.dirname(__file__), 'golden', 'make_input_golden.py')
    subprocess.run([sys.executable, script], check=True, env=env, capture_output=True, timeout=600)
    new, old = np.load(str(tmp_path / 'ref_shim_input_pipeline.npz')), np.load(GOLD)
    assert set(new.files) == set(old.files)
    for k in old.files:
        assert np.array_equal(new[k], old[k]), k


def test_tf_resize_kernels_restated_sanely():
    """independent anchors for the unpinned TF kernels: identity at equal size, constants stay constant, bilinear ==
    torch align_corners=True, bicubic == torch's A=-0.75 cubic up to the 1/1024 weight table, nearest picks pixels."""
    import torch.nn.functional as F
    r = np.random.RandomState(0)
    img = r.uniform(0, 1, (37, 53, 3)).astype(np.float32)
    for m in range(4):
        assert np.allclose(io_.resize_images(img, (37, 53), m), img, atol=1e-6), m
        c = io_.resize_images(np.full((20, 30, 3), 0.25, np.float32), (33, 17), m)
        assert np.allclose(c, 0.25, atol=1e-6), m
    t = torch.from_numpy(img).permute(2, 0, 1)[None]
    for size in [(64, 64), (20, 91), (37, 106)]:
        bl = F.interpolate(t, size=size, mode='bilinear', align_corners=True)[0].permute(1, 2, 0).numpy()
        assert np.abs(io_.resize_images(img, size, 0) - bl).max() < 2e-6
        bc = F.interpolate(t, size=size, mode='bicubic', align_corners=True)[0].permute(1, 2, 0).numpy()
        assert np.abs(io_.resize_images(img, size, 2) - bc).max() < 4e-3
        nn = io_.resize_images(img, size, 1)
        assert set(np.unique(nn)) <= set(np.unique(img))
        assert np.array_equal(nn[0, 0], img[0, 0]) and np.array_equal(nn[-1, -1], img[-1, -1])   # align_corners
    # area: an exact 2x box average when (in-1)/(out-1) == 2 and the boxes stay inside the image
    a = io_.resize_images(img[:33, :33], (17, 17), 3)
    assert np.allclose(a[3, 5], img[6:8, 10:12].mean((0, 1)), atol=1e-6)


def _protobuf_example(feats):
    """tf.train.Example serialised by the real protobuf runtime from descriptors built here (example.proto/feature.proto)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name='merlot_test_example.proto', package='mt', syntax='proto3')

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=1, type_name=None, packed=None):
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, num, typ, label
        if type_name:
            f.type_name = type_name
        if packed is not None:
            f.options.packed = packed
        return f

    T = descriptor_pb2.FieldDescriptorProto
    field(msg('BytesList'), 'value', 1, T.TYPE_BYTES, 3)
    field(msg('FloatList'), 'value', 1, T.TYPE_FLOAT, 3, packed=True)
    field(msg('Int64List'), 'value', 1, T.TYPE_INT64, 3, packed=True)
    fe = msg('Feature')
    fe.oneof_decl.add().name = 'kind'
    for i, (n, tn) in enumerate([('bytes_list', '.mt.BytesList'), ('float_list', '.mt.FloatList'), ('int64_list', '.mt.Int64List')]):
        f = field(fe, n, i + 1, T.TYPE_MESSAGE, type_name=tn)
        f.oneof_index = 0
    fs = msg('Features')
    entry = fs.nested_type.add()
    entry.name = 'FeatureEntry'
    entry.options.map_entry = True
    field(entry, 'key', 1, T.TYPE_STRING)
    field(entry, 'value', 2, T.TYPE_MESSAGE, type_name='.mt.Feature')
    field(fs, 'feature', 1, T.TYPE_MESSAGE, 3, type_name='.mt.Features.FeatureEntry')
    field(msg('Example'), 'features', 1, T.TYPE_MESSAGE, type_name='.mt.Features')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Example = message_factory.GetMessageClass(pool.FindMessageTypeByName('mt.Example'))
    ex = Example()
    for k, (kind, vals) in feats.items():
        getattr(ex.features.feature[k], kind + '_list').value.extend(vals)
    return ex.SerializeToString(deterministic=True)


def test_example_wire_format_against_protobuf_runtime():
    feats = {'c00/image/encoded': ('bytes', [b'\xff\xd8jpegbytes\x00\x01']), 'c00/tokenized_raw_asr': ('int64', [5, 300, 70000, 2 ** 40]),
             'c00/mean_time': ('float', [12.5]), 'c00/is_eoc': ('int64', [1]), 'c01/tokenized_cleaned_asr': ('int64', []),
             'c01/youtube_id': ('bytes', [b'WAaKRUoY6Io']), 'neg': ('int64', [-3])}
    wire = _protobuf_example(feats)
    got = ip.parse_example(wire)
    for k, (kind, vals) in feats.items():
        assert got[k][1] == vals and (got[k][0] == kind or not vals), k
    mine = ip.encode_example({k: (np.float32(v[0]) if kind == 'float' else v) for k, (kind, v) in feats.items()
                              if k != 'c01/tokenized_cleaned_asr'})
    assert ip.parse_example(mine) == {k: v for k, v in got.items() if k != 'c01/tokenized_cleaned_asr'}
    # the native indexer agrees with the Python parser, value for value
    for w in (wire, mine, ip.encode_example({'a': [b'x', b'yy'], 'b': [1.5, 2.5], 'c': []})):
        nat, py = ip.parse_example_native(w), ip.parse_example(w)
        assert set(nat) == set(py)
        for k in py:
            assert nat[k][0] == py[k][0] or not py[k][1], k
            assert [bytes(v) if isinstance(v, memoryview) else v for v in nat[k][1]] == \
                   [bytes(v) if isinstance(v, memoryview) else v for v in py[k][1]], k
    for bad in (wire[:-3], b'\x0a\xff\xff\xff\xff\x0f', bytes([0x0a, 0x02, 0x0a, 0x05])):
        with pytest.raises((ip.RecordError, Exception)):
            ip.parse_example_native(bad)
    # unpacked repeated scalars (older writers) parse too
    unpacked = ip._ld(1, ip._ld(1, ip._ld(1, b'k') + ip._ld(2, ip._ld(3, bytes([0x08, 0x07, 0x08, 0x09])))))
    assert ip.parse_example(unpacked) == {'k': ('int64', [7, 9])} == ip.parse_example_native(unpacked)


def test_tfrecord_framing(tmp_path):
    path = str(tmp_path / 'a.tfrecord')
    recs = [b'', b'x', os.urandom(1000), b'hello world']
    with ip.TFRecordWriter(path) as w:
        for r in recs:
            w.write(r)
    assert list(ip.read_tfrecords(path)) == recs
    raw = open(path, 'rb').read()
    # hand-checked frame of the empty record: length 0, masked crc of eight zero bytes, masked crc of b''
    from merlot_amd import checkpoint as ck
    assert raw[:16] == struct.pack('<Q', 0) + struct.pack('<I', ck.mask_crc(ck.crc32c(bytes(8)))) + struct.pack('<I', ck.mask_crc(0))
    bad = bytearray(raw)
    bad[40] ^= 1
    open(path, 'wb').write(bytes(bad))
    with pytest.raises(ip.RecordError, match='corrupt'):
        list(ip.read_tfrecords(path))
    open(path, 'wb').write(raw[:-3])
    with pytest.raises(ip.RecordError, match='truncated'):
        list(ip.read_tfrecords(path))


def _jpeg(arr):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr, mode='RGB').save(b, format='JPEG', quality=90)
    return b.getvalue()


def _write_records(path, n_examples, num_chunks, seed, vocab=50000):
    """records as data/process.py:236-256 writes them"""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    r = np.random.RandomState(seed)
    truth = []
    with ip.TFRecordWriter(path) as w:
        for e in range(n_examples):
            feats, ex = {}, []
            for i in range(num_chunks):
                h, wd = int(r.randint(40, 90)), int(r.randint(50, 120))
                yy, xx = np.mgrid[0:h, 0:wd]
                img = np.stack([(yy * 3 + e * 20) % 256, (xx * 2 + i * 30) % 256, (yy + xx) % 256], -1).astype(np.uint8)
                enc = _jpeg(img)
                clean = [int(t) for t in r.randint(100, vocab, r.randint(0, 45))]
                raw = [int(t) for t in r.randint(100, vocab, r.randint(1, 20))]
                c = {'image/encoded': enc, 'image/height': h, 'image/width': wd, 'image/key/sha256': b'00', 'image/format': b'jpeg',
                     'youtube_id': f'vid{e}_{i // 3}'.encode(), 'tokenized_cleaned_asr': clean, 'tokenized_raw_asr': raw,
                     'is_eoc': int(i % 3 == 2), 'mean_time': np.float32(1.5 * i), 'chunk_num': i}
                ex.append(c)
                for k, v in c.items():
                    if k == 'is_eoc' and i == 1:
                        continue                                        # exercises the FixedLenFeature default (1)
                    feats[f'c{i:02d}/{k}'] = v
            w.write(ip.encode_example(feats))
            truth.append(ex)
    return truth


def test_parse_example_host_matches_oracle(tmp_path):
    path = str(tmp_path / 'train000.tfrecord')
    truth = _write_records(path, 3, 4, seed=0)
    cfg = {'num_chunks': 4, 'image_size': [64, 64], 'chunk_text_len': 32, 'augment_prob': 0.5}
    rng = np.random.default_rng(0)
    for rec, ex in zip(ip.read_tfrecords(path), truth):
        chunks = ip.decode_record(rec, 4)
        assert chunks[1]['is_eoc'] == 1 and chunks[0]['is_eoc'] == 0 and chunks[2]['is_eoc'] == 1
        assert chunks[0]['tokenized_raw_asr'] == ex[0]['tokenized_raw_asr'] and chunks[3]['mean_time'] == 4.5
        noise = ip.draw_example_noise(rng, 4, cfg)
        f = ip.parse_example_host(rec, cfg, noise)
        ids, is_eoc, vsrc = io_.text_features(chunks, noise['do_clean'], 4, 32, START=2, NEXTCAPTION_START=5)
        assert np.array_equal(f['input_ids'], ids) and np.array_equal(f['is_eoc'], is_eoc) and np.array_equal(f['video_src_ids'], vsrc)
        assert f['input_ids'][0, 0] == (2 if noise['do_clean'] else 5) and is_eoc[-1]
        assert np.array_equal(f['youtube_id'][0], io_.encode_string(ex[0]['youtube_id'], 64))
        for i in range(4):
            assert f['frames_u8'][i].shape == (ex[i]['image/height'], ex[i]['image/width'], 3)
            j, n = f['jobs'][i], noise['frames'][i]
            geo = io_.resize_geometry(j['src_h'], j['src_w'], (64, 64), n['scale'], n['u_y'], n['u_x'])
            assert (j['scaled_h'], j['scaled_w'], j['offset_y'], j['offset_x']) == geo[:4]
            assert j['aug_kind'] == (2 if n['do_augment'] else 0) and j['method'] == n['method']   # quirk: always contrast


def test_shuffle_chunks_keeps_videos_contiguous():
    r = np.random.RandomState(1)
    vs = np.array([[0, 0, 0, 1, 1, 2], [0, 1, 1, 1, 1, 1]], np.int32)
    u = r.uniform(size=vs.shape).astype(np.float32)
    idx = ip.shuffle_chunks_index(vs, u)
    assert np.array_equal(idx, io_.shuffle_chunks_index(vs, u))
    for b in range(2):
        assert sorted(idx[b]) == list(range(6))
        moved = vs[b][idx[b]]
        runs = [moved[0]] + [moved[i] for i in range(1, 6) if moved[i] != moved[i - 1]]
        assert len(runs) == len(set(runs))                               # each source video stays one block, in order
        for v in set(vs[b]):
            assert list(idx[b][moved == v]) == list(np.where(vs[b] == v)[0])


def test_pipeline_end_to_end_with_emulated_kernels(tmp_path, emu):
    from merlot_amd.config import NeatConfig
    for i in range(4):
        _write_records(str(tmp_path / f'train{i:03d}.tfrecord'), 3, 4, seed=10 + i)
    cfg = NeatConfig.from_dict({
        'data': {'train_file': str(tmp_path / 'train*.tfrecord'), 'val_file': str(tmp_path / 'train000.tfrecord'), 'num_chunks': 4,
                 'chunk_text_len': 32, 'shuffle_buffer_size': 4, 'shuffle_chunks': True, 'augment_prob': 0.6, 'num_threads': 2},
        'model': {'image_size': [64, 64], 'num_chunks_in_group': 4, 'image_shuffle_prob': 0.4, 'use_bfloat16': True},
        'optimizer': {}, 'device': {'output_dir': str(tmp_path)}})
    pipe = ip.InputPipeline(cfg, True, batch_size=2, device='cpu', seed=3)
    it = iter(pipe)
    b0, b1 = next(it), next(it)
    assert b0['images'].shape == (8, 64, 64, 3) and b0['images'].dtype == torch.bfloat16
    assert b0['input_ids'].shape == (2, 4, 32) and b0['input_ids'].dtype == torch.int32
    assert b0['shuffled_idx_img'].shape == (8,) and b0['video_src_ids'].shape == (2, 4)
    # (bicubic resizing overshoots [0, 1] by a little and the reference does not clip un-augmented frames)
    assert float(b0['images'].float().min()) >= -0.1 and float(b0['images'].float().max()) <= 1.1
    assert not torch.equal(b0['input_ids'], b1['input_ids'])
    again = next(iter(ip.InputPipeline(cfg, True, batch_size=2, device='cpu', seed=3)))
    assert all(torch.equal(again[k], b0[k]) for k in b0)                  # same seed, same batch
    other = next(iter(ip.InputPipeline(cfg, True, batch_size=2, device='cpu', seed=4)))
    assert not torch.equal(other['images'], b0['images'])
    # loader worker processes (frames come back through shared memory) produce the very same batches
    wp = ip.InputPipeline(cfg, True, batch_size=2, device='cpu', seed=3, num_workers=2)
    wit = iter(wp)
    w0, w1 = next(wit), next(wit)
    wp.close()
    assert all(torch.equal(w0[k], b0[k]) for k in b0) and all(torch.equal(w1[k], b1[k]) for k in b1)
    # per-rank file sharding (model/dataloader.py:160-166) and the eval path (no repeat, no _process_example)
    p0 = ip.InputPipeline(cfg, True, 2, 'cpu', rank=0, world_size=2)
    p1 = ip.InputPipeline(cfg, True, 2, 'cpu', rank=1, world_size=2)
    assert not set(p0.files) & set(p1.files) and len(p0.files) == len(p1.files) == 2
    with pytest.raises(ip.RecordError, match='sharded'):
        ip.InputPipeline(cfg, True, 2, 'cpu', rank=0, world_size=8)
    ev = list(ip.InputPipeline(cfg, False, batch_size=2, device='cpu'))
    assert len(ev) == 1 and ev[0]['images'].shape == (2, 4, 64, 64, 3) and 'shuffled_idx_img' not in ev[0]
    # shuffled_idx_img of the batch == the index oracle on the same draws
    rng = np.random.default_rng(7)
    nz = ip.draw_batch_noise(rng, 2, 4, dict(cfg.model, **cfg.data))
    want = index_oracle.shuffled_idx(nz['num_shuffle'], nz['u_sel'], nz['u_perm'], 16) if hasattr(index_oracle, 'shuffled_idx') \
        else io_.shuffled_idx_img(nz['num_shuffle'], nz['u_sel'], nz['u_perm'], 4)
    assert np.array_equal(io_.shuffled_idx_img(nz['num_shuffle'], nz['u_sel'], nz['u_perm'], 4), np.asarray(want).reshape(-1))


def test_example_parsers_agree_on_random_messages_and_survive_garbage():
    """property test: for random feature maps the native indexer, the Python parser and the encoder agree; on truncated or
    random bytes the native indexer either agrees with the Python parser or reports malformed input -- it never reads
    outside the buffer (the records come from disk)."""
    from hypothesis import given, settings, strategies as st
    keys = st.text(alphabet='abc/0123456789_', min_size=1, max_size=12)
    feature = st.one_of(st.lists(st.binary(max_size=40), min_size=1, max_size=1),
                        st.lists(st.integers(-2 ** 63, 2 ** 63 - 1), max_size=30),
                        st.lists(st.floats(width=32, allow_nan=False), min_size=1, max_size=10).map(lambda v: [np.float32(x) for x in v]))

    def norm(d):
        return {k: (kind, [bytes(v) if isinstance(v, memoryview) else v for v in vals]) for k, (kind, vals) in d.items()}

    @settings(max_examples=150, deadline=None, derandomize=True, database=None)
    @given(st.dictionaries(keys, feature, max_size=8), st.integers(0, 64), st.binary(max_size=64))
    def run(feats, cut, junk):
        wire = ip.encode_example(feats)
        nat, py = norm(ip.parse_example_native(wire)), norm(ip.parse_example(wire))
        assert set(nat) == set(py) == set(feats)
        for k, v in feats.items():
            want = [float(x) for x in v] if v and isinstance(v[0], np.floating) else list(v)
            assert nat[k][1] == py[k][1] == want, k
        for bad in (wire[:max(0, len(wire) - 1 - cut)], junk, wire + junk):
            try:
                ref = norm(ip.parse_example(bad))
            except Exception:
                ref = None
            try:
                got = norm(ip.parse_example_native(bad))
            except ip.RecordError:
                got = None
            if ref is not None and got is not None:
                assert {k: v[1] for k, v in got.items()} == {k: v[1] for k, v in ref.items()}

    run()


def test_train_loop_resumes_like_the_estimator(tmp_path, emu):
    """merlot_amd.train.train = model/train.py's estimator.train: records -> steps, a checkpoint every
    iterations_per_loop steps in the reference's bundle format, resume from output_dir on the next call."""
    from merlot_amd import checkpoint as ck, train as T
    from merlot_amd.config import NeatConfig
    from common import tiny_config
    for i in range(2):
        _write_records(str(tmp_path / f'train{i:03d}.tfrecord'), 3, 4, seed=40 + i, vocab=2000)
    out = str(tmp_path / 'out')
    config = NeatConfig.from_dict({
        'data': {'train_file': str(tmp_path / 'train*.tfrecord'), 'num_chunks': 4, 'chunk_text_len': 32, 'shuffle_buffer_size': 4,
                 'num_threads': 2},
        'model': dict(tiny_config(), vocab_size=2048),            # small word table: four checkpoints are written
        'optimizer': {'type': 'adam_optimizer', 'learning_rate': 1e-4, 'num_train_steps': 100, 'num_warmup_steps': 10,
                      'weight_decay_rate': 0.1, 'beta_2': 0.98, 'use_bfloat16_adam': True},
        'device': {'output_dir': out, 'train_batch_size': 2, 'iterations_per_loop': 2}})
    t1 = T.train(config, 'cpu', max_steps=3, log_every=0)
    assert t1.step_idx == 3 and t1.opt.step_count == 3
    assert ck.latest_checkpoint(out).endswith('model.ckpt-3')
    assert os.path.exists(os.path.join(out, 'model.ckpt-2.index'))             # the periodic one
    t2 = T.train(config, 'cpu', max_steps=4, log_every=0)                        # resumes at 3, runs ONE more step
    assert t2.step_idx == 4 and ck.latest_checkpoint(out).endswith('model.ckpt-4')
    assert int(ck.load_variable(out, 'global_step')) == 4
    assert not torch.equal(t1.store.master, t2.store.master)


def test_training_records_interleave_files_and_carry_mask_noise(tmp_path, emu):
    """model/dataloader.py:139-150 interleaves one record at a time from min(num_threads, num_files) open files before
    the shuffle buffer (a batch must not come from one or two files), and the masking noise is drawn in the loader, not in
    the training step."""
    from merlot_amd.config import NeatConfig
    for i in range(6):
        _write_records(str(tmp_path / f'train{i:03d}.tfrecord'), 4, 4, seed=30 + i)
    cfg = NeatConfig.from_dict({
        'data': {'train_file': str(tmp_path / 'train*.tfrecord'), 'val_file': str(tmp_path / 'train000.tfrecord'), 'num_chunks': 4,
                 'chunk_text_len': 32, 'shuffle_buffer_size': 1, 'shuffle_chunks': True, 'num_threads': 64},
        'model': dict(__import__('common').tiny_config(), image_size=[64, 64]), 'optimizer': {}, 'device': {'output_dir': str(tmp_path)}})
    pipe = ip.InputPipeline(cfg, True, batch_size=2, device='cpu', seed=5)
    recs = list(itertools.islice(pipe._records(), 30))
    files = [os.path.basename(r[0]) for r in recs]
    assert len(set(files[:6])) == 6                        # the first six records come from six different files
    assert all(len(set(files[i:i + 6])) == 6 for i in range(0, 24, 6))     # and so does every following round of the cycle
    assert len(recs) == 30 and len(set((r[0], r[1]) for r in recs[:24])) == 24   # one epoch = every record once (24), then it repeats
    ev = ip.InputPipeline(cfg, False, batch_size=2, device='cpu')
    assert [os.path.basename(r[0]) for r in ev._records()] == ['train000.tfrecord'] * 4       # evaluation: files in order
    b = next(iter(pipe))
    nz = b['noise']
    B, L = 2, 4 * 32
    assert set(nz) == {'gumbel', 'span_lower', 'span_upper', 'random_ids', 'option'}
    assert nz['gumbel'].shape == (B, L) and nz['random_ids'].shape == (B * L,) and nz['option'].dtype == torch.int32
    other = next(iter(ip.InputPipeline(cfg, True, batch_size=2, device='cpu', seed=5, rank=1, world_size=2)))
    assert not torch.equal(other['noise']['gumbel'], nz['gumbel'])          # rank-keyed: replicas draw independent noise
