"""CPU: TF tensor-bundle checkpoint reader / writer and the reference's init_checkpoint logic.

No bundle written by TensorFlow exists in this container, so the byte format is checked against (i) published CRC-32C
vectors (RFC 3720 B.4), (ii) hand-assembled protobuf / table bytes derived from tensor_bundle.proto and the leveldb
table format, not from the encoder under test, and (iii) round trips."""
import os
import struct

import numpy as np
import pytest
import torch

from merlot_amd import checkpoint as ck
from merlot_amd.lib import LIB
from merlot_amd.optimization import AdamOptimizer
from merlot_amd.params import ParamStore
from tests import common


def test_crc32c_known_answers():
    d = LIB.load()
    vecs = [(b'123456789', 0xE3069283), (bytes(32), 0x8A9136AA), (b'\xff' * 32, 0x62A8AB43),
            (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C), (b'', 0)]
    for data, want in vecs:
        assert ck.crc32c(data) == want
        assert d.merlot_crc32c(0, data, len(data), 1) == want            # table path
    x = np.random.RandomState(0).randint(0, 256, 100003).astype(np.uint8)
    whole = ck.crc32c(x)
    assert ck.crc32c(x[777:].copy(), ck.crc32c(x[:777].copy())) == whole
    assert d.merlot_crc32c(0, x.ctypes.data, x.nbytes, 1) == whole
    for c in [0, 1, 0xE3069283, 0xffffffff, 0x80000000]:
        assert ck.unmask_crc(ck.mask_crc(c)) == c
    assert ck.mask_crc(0) == 0xa282ead8 and ck.mask_crc(0x8000) == 0xa282ead9   # rotr 15 then + kMaskDelta


def test_entry_proto_bytes_are_the_schema():
    """BundleEntryProto{dtype=DT_FLOAT(1), shape{dim{size:2} dim{size:3}}, size=24, crc32c=0x01020304}, offset 0
    omitted (proto3), assembled by hand from tensor_bundle.proto / tensor_shape.proto field numbers."""
    want = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03, 0x28, 0x18,
                  0x35, 0x04, 0x03, 0x02, 0x01])
    assert ck._encode_entry(1, (2, 3), 0, 24, 0x01020304) == want
    e = ck._parse_entry(want)
    assert e == ck.Entry(1, (2, 3), 0, 0, 24, 0x01020304, False)
    # a field order / extra fields a newer writer may emit: shard_id, offset > 127 (2-byte varint), scalar shape
    other = bytes([0x08, 0x0e, 0x12, 0x00, 0x18, 0x00, 0x20, 0x80, 0x01, 0x28, 0x02, 0x35, 0, 0, 0, 0])
    assert ck._parse_entry(other) == ck.Entry(14, (), 0, 128, 2, 0, False)
    assert ck._encode_header(1) == bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])


def test_hand_assembled_table_is_read(tmp_path):
    """one data block with two prefix-compressed keys, built byte by byte from the leveldb table format."""
    v0, v1 = ck._encode_header(1), ck._encode_entry(3, (2,), 0, 8, ck.mask_crc(ck.crc32c(struct.pack('<ii', 7, -9))))
    v2 = ck._encode_entry(3, (1,), 8, 4, ck.mask_crc(ck.crc32c(struct.pack('<i', 5))))
    blk = bytes([0, 0, len(v0)]) + v0                                    # key ""
    blk += bytes([0, 3, len(v1)]) + b'a/b' + v1                          # key "a/b"
    blk += bytes([2, 1, len(v2)]) + b'c' + v2                            # shares "a/" -> key "a/c"
    blk += struct.pack('<II', 0, 1)                                      # one restart at 0
    img = blk + b'\x00' + struct.pack('<I', ck.mask_crc(ck.crc32c(blk + b'\x00')))
    meta_off = len(img)
    meta = struct.pack('<II', 0, 1)
    img += meta + b'\x00' + struct.pack('<I', ck.mask_crc(ck.crc32c(meta + b'\x00')))
    idx_off = len(img)
    idx = bytes([0, 3, 2]) + b'a/c' + bytes([0, len(blk)]) + struct.pack('<II', 0, 1)
    img += idx + b'\x00' + struct.pack('<I', ck.mask_crc(ck.crc32c(idx + b'\x00')))
    footer = bytes([meta_off, len(meta), idx_off, len(idx)])
    assert max(footer) < 128
    img += footer + bytes(40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    prefix = str(tmp_path / 'model.ckpt')
    open(prefix + '.index', 'wb').write(img)
    open(prefix + '.data-00000-of-00001', 'wb').write(struct.pack('<iii', 7, -9, 5))
    assert ck.list_variables(prefix) == [('a/b', [2]), ('a/c', [1])]
    assert ck.load_variable(prefix, 'a/b').tolist() == [7, -9] and ck.load_variable(prefix, 'a/c').tolist() == [5]
    # the writer produces the same data block for the same content (restart interval 16, no compression)
    ck.write_checkpoint(str(tmp_path / 'w.ckpt'), {'a/b': torch.tensor([7, -9], dtype=torch.int32),
                                                  'a/c': torch.tensor([5], dtype=torch.int32)})
    wimg = open(str(tmp_path / 'w.ckpt.index'), 'rb').read()
    assert wimg[:len(blk) + 5] == img[:len(blk) + 5]
    assert wimg[-8:] == img[-8:]


def test_round_trip_many_blocks_and_dtypes(tmp_path):
    g = torch.Generator().manual_seed(0)
    tensors = {f'scope{i // 7}/layer{i:02d}/kernel': torch.randn((i % 5 + 1, 3), generator=g) for i in range(200)}
    tensors['global_step'] = torch.tensor(1234567, dtype=torch.int64)
    tensors['m16'] = torch.randn((4, 5), generator=g).to(torch.bfloat16)
    tensors['flags'] = torch.tensor([True, False])
    tensors['empty'] = torch.zeros((0, 3))
    tensors['unicode/ünï'] = torch.arange(5, dtype=torch.int32)
    prefix = str(tmp_path / 'sub' / 'model.ckpt-7')
    ck.write_checkpoint(prefix, tensors, block_size=256)                 # forces ~100 data blocks
    r = ck.CheckpointReader(prefix)
    assert set(r.entries) == set(tensors)
    for k, v in tensors.items():
        got = r.get_tensor(k)
        assert got.dtype == v.dtype and got.shape == v.shape and torch.equal(got, v), k
    assert ck.latest_checkpoint(str(tmp_path / 'sub')) == prefix
    assert ck.CheckpointReader(str(tmp_path / 'sub')).prefix == prefix      # a directory resolves through `checkpoint`
    assert r.get_variable_to_dtype_map()['m16'] == 'bfloat16'
    assert [n for n, _ in ck.list_variables(prefix)] == sorted(tensors)
    assert sorted(tensors, key=lambda s: s.encode()) == list(r.entries)    # table order = bytewise key order


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'model.ckpt')
    ck.write_checkpoint(prefix, {'w': torch.arange(64, dtype=torch.float32), 'b': torch.ones(3)})
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    data[20] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    r = ck.CheckpointReader(prefix)
    with pytest.raises(ck.CheckpointError, match='checksum'):
        r.get_tensor('w')
    assert ck.CheckpointReader(prefix, verify=False).get_tensor('w').shape == (64,)
    idx = bytearray(open(prefix + '.index', 'rb').read())
    bad = bytearray(idx)
    bad[3] ^= 0x40
    open(prefix + '.index', 'wb').write(bytes(bad))
    with pytest.raises(ck.CheckpointError, match='checksum'):
        ck.CheckpointReader(prefix)
    bad = bytearray(idx)
    bad[-1] ^= 1
    open(prefix + '.index', 'wb').write(bytes(bad))
    with pytest.raises(ck.CheckpointError, match='magic'):
        ck.CheckpointReader(prefix)
    open(prefix + '.index', 'wb').write(bytes(idx[:20]))
    with pytest.raises(ck.CheckpointError):
        ck.CheckpointReader(prefix)
    with pytest.raises(ck.CheckpointError, match='not found'):
        ck.CheckpointReader(str(tmp_path / 'nothing'))


def _store_and_opt(seed, bf16_adam=True):
    cfg = dict(common.tiny_config(), vocab_size=1024)          # small word table: keeps the files at a few MB
    st = ParamStore(cfg, torch.device('cpu'), seed=seed)
    opt = AdamOptimizer(st, 1e-4, 1000, 100, weight_decay_rate=0.1, use_bfloat16_adam=bf16_adam)
    g = torch.Generator().manual_seed(seed + 100)
    for name in st.names():                              # slot contents incl. sign-encoded v patterns (negative values)
        st.view(opt.m, name).copy_(torch.randn(st.offsets[name][2], generator=g))
        st.view(opt.v, name).copy_(torch.randn(st.offsets[name][2], generator=g))
    opt.step_count = 4321
    return cfg, st, opt


@pytest.mark.parametrize('bf16_adam', [True, False])
def test_save_restore_is_exact_and_uses_reference_names(tmp_path, bf16_adam):
    from oracle import merlot_oracle as mo
    cfg, st, opt = _store_and_opt(1, bf16_adam)
    prefix = ck.save_checkpoint(str(tmp_path), st, opt)
    assert prefix.endswith('model.ckpt-4321')
    names = dict(ck.list_variables(prefix))
    shapes = mo.variable_shapes(cfg)                     # the reference's variable set (pinned by the shim fixtures)
    assert set(names) == set(shapes) | {k + s for k in shapes for s in ('/adam_m', '/adam_v')} | {'global_step'}
    for k, shp in shapes.items():
        assert names[k] == list(shp) and names[k + '/adam_m'] == list(shp)
    r = ck.CheckpointReader(prefix)
    want = 'bfloat16' if bf16_adam else 'float32'        # utils/optimization.py:372-383: slots in the state dtype
    k0 = 'encoder/layer00/query_layer/kernel'
    assert r.get_variable_to_dtype_map()[k0 + '/adam_v'] == want and r.get_variable_to_dtype_map()[k0] == 'float32'
    assert r.get_tensor('global_step').dtype == torch.int64
    _, st2, opt2 = _store_and_opt(2, bf16_adam)
    assert not torch.equal(st2.master, st.master)
    assert ck.restore_checkpoint(str(tmp_path), st2, opt2) == 4321
    assert torch.equal(st2.master, st.master) and opt2.step_count == 4321
    assert torch.equal(opt2.m.view(torch.int16 if bf16_adam else torch.int32), opt.m.view(torch.int16 if bf16_adam else torch.int32))
    assert torch.equal(opt2.v.view(torch.int16 if bf16_adam else torch.int32), opt.v.view(torch.int16 if bf16_adam else torch.int32))


def test_init_from_checkpoint_follows_the_assignment_map(tmp_path):
    cfg, st, opt = _store_and_opt(3)
    full = dict(st.export_tf_weights())
    kq, kk = 'encoder/layer00/query_layer/kernel', 'encoder/layer00/key_layer/kernel'
    ckpt = {k: v for k, v in full.items() if not k.startswith('vision_backbone/') and k != kk}
    ckpt[kq + '/adam_m'] = torch.full(full[kq].shape, 0.5).to(torch.bfloat16)
    ckpt['global_step'] = torch.tensor(99, dtype=torch.int64)
    ckpt['some/other/model/kernel'] = torch.zeros(3)
    prefix = str(tmp_path / 'model.ckpt')
    ck.write_checkpoint(prefix, ckpt)
    amap, init_names = ck.get_assignment_map_from_checkpoint([n + ':0' for n in ck.variable_names(st)], prefix)
    assert kq in amap and kk not in amap and 'global_step' not in amap and 'some/other/model/kernel' not in amap
    assert init_names[kq] == 1 and init_names[kq + ':0'] == 1
    amap2, _ = ck.get_assignment_map_from_checkpoint(['x/' + kq], prefix, reference_name_transform=lambda n: 'x/' + n)
    assert amap2 == {kq: 'x/' + kq}
    _, st2, opt2 = _store_and_opt(4)
    before = dict(st2.export_tf_weights())
    m_before = opt2.m.clone()
    init = ck.init_from_checkpoint(st2, prefix, opt2)
    after = dict(st2.export_tf_weights())
    for k in full:
        if k.startswith('vision_backbone/') or k == kk:
            assert torch.equal(after[k], before[k]) and k not in init, k        # untouched: keeps its initial value
        else:
            assert torch.equal(after[k], full[k]) and k in init, k
    assert opt2.step_count == 4321                                               # global_step is never taken
    H = cfg['hidden_size']
    mq = st2.view(opt2.m, 'encoder/layer00/qkv/kernel')
    assert torch.all(mq[:H] == 0.5) and torch.equal(mq[H:], st2.view(m_before, 'encoder/layer00/qkv/kernel')[H:])
    # inference graph (tf.trainable_variables() only, model/modeling.py:718): the slots are not read
    _, st3, opt3 = _store_and_opt(5)
    m3 = opt3.m.clone()
    ck.init_from_checkpoint(st3, prefix)
    assert torch.equal(opt3.m, m3)


def test_trainer_config_hook_names(tmp_path):
    """`init_checkpoint` under `model:` is the key the reference reads (model/modeling.py:724)."""
    import inspect
    from merlot_amd import train
    src = inspect.getsource(train.Trainer.__init__)
    assert "config.model.get('init_checkpoint'" in src


def test_table_round_trip_and_corruption_properties():
    """property test of the SSTable layer: any sorted key/value set survives write -> read at any block size; any single
    corrupted byte is either detected (CheckpointError) or leaves the content intact (it hit padding)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=120, deadline=None, derandomize=True, database=None)
    @given(st.dictionaries(st.binary(min_size=1, max_size=24), st.binary(max_size=60), max_size=40),
           st.integers(16, 400), st.integers(0, 10 ** 6), st.integers(0, 255))
    def run(kv, block_size, where, xor):
        items = [(b'', b'header')] + sorted(kv.items())
        img = ck._write_table(items, block_size)
        assert ck._read_table(img) == items
        if xor:
            bad = bytearray(img)
            bad[where % len(bad)] ^= xor
            try:
                got = ck._read_table(bytes(bad))
            except ck.CheckpointError:
                return
            assert got == items

    run()


def test_slot_dtype_mismatch_is_an_error_and_the_state_file_keeps_every_checkpoint(tmp_path):
    """a use_bfloat16_adam checkpoint (sign-encoded bf16 v) restored into an fp32 optimizer would give negative
    second moments; and the `checkpoint` state file lists every prefix still on disk (keep_checkpoint_max=None), written
    atomically."""
    cfg, st, opt = _store_and_opt(1, bf16_adam=True)
    p1 = ck.save_checkpoint(str(tmp_path), st, opt)
    opt.step_count = 5000
    p2 = ck.save_checkpoint(str(tmp_path), st, opt)
    state = open(tmp_path / 'checkpoint').read().splitlines()
    assert state[0] == 'model_checkpoint_path: "model.ckpt-5000"'
    assert state[1:] == ['all_model_checkpoint_paths: "model.ckpt-4321"', 'all_model_checkpoint_paths: "model.ckpt-5000"']
    assert not os.path.exists(tmp_path / 'checkpoint.tmp')
    assert ck.latest_checkpoint(str(tmp_path)) == p2 and p1 != p2
    _, st32, opt32 = _store_and_opt(2, bf16_adam=False)
    with pytest.raises(ck.CheckpointError, match='use_bfloat16_adam'):
        ck.restore_checkpoint(str(tmp_path), st32, opt32)
    with pytest.raises(ck.CheckpointError, match='use_bfloat16_adam'):
        ck.init_from_checkpoint(st32, p2, opt32)
    ck.init_from_checkpoint(st32, p2, None)                # weights alone are always fine
