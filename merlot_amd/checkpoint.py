"""Reference checkpoints without TensorFlow: reader and writer for the TF tensor-bundle format (`<prefix>.index` +
`<prefix>.data-00000-of-00001`), and the `init_checkpoint` / resume logic of the reference on top of them.

What it replaces:
  * `tf.train.list_variables` + `get_assignment_map_from_checkpoint` (utils/model_utils.py:388-413),
  * `tf.train.init_from_checkpoint` inside the scaffold (model/modeling.py:716-738; downstream/vcr/modeling.py:158-172),
  * the Estimator's own save / auto-resume of `model_dir` (`model.ckpt-<step>` + the `checkpoint` state file),
  * the files `download_checkpoint.py:24-31` fetches (`model.ckpt.{index,data-00000-of-00001}`; `.meta` is unused here).

Format restated from tensorflow==1.15.5 (requirements.txt:70; the library itself is absent here -- PARITY UNPINNED for
the byte format: no bundle written by TensorFlow exists in this container):
  `.index` is an SSTable (leveldb table format, tensorflow/core/lib/io/{table_builder,format,block}.cc): data blocks of
  prefix-compressed (shared, non_shared, value_len varint32; key tail; value) entries + uint32 restart array, each
  block followed by a 1-byte compression type and the masked CRC-32C of block+type; an index block mapping separator
  keys to BlockHandles (varint64 offset, size); a 48-byte footer = metaindex handle, index handle, padding, magic
  0xdb4775248b80fb57.  Key "" holds BundleHeaderProto {num_shards=1, endianness=2, version=3}; every other key is a
  variable name holding BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32, masked),
  slices=7} (tensorflow/core/protobuf/tensor_bundle.proto).  The data file is the raw little-endian tensor bytes.
"""
import collections
import os
import re
import struct

import numpy as np
import torch

from .lib import LIB

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DT = {1: ('float32', torch.float32), 2: ('float64', torch.float64), 3: ('int32', torch.int32), 4: ('uint8', torch.uint8),
      5: ('int16', torch.int16), 6: ('int8', torch.int8), 9: ('int64', torch.int64), 10: ('bool', torch.bool),
      14: ('bfloat16', torch.bfloat16), 19: ('float16', torch.float16)}
_DT_OF_TORCH = {v[1]: k for k, v in DT.items()}
DT_STRING = 7


class CheckpointError(ValueError):
    pass


# ---- checksums -------------------------------------------------------------------------------------------------------
def crc32c(data, crc=0):
    """CRC-32C of a bytes-like / contiguous numpy array (native: merlot_crc32c in libmerlot_hip.so)."""
    if isinstance(data, np.ndarray):
        return int(LIB.query('merlot_crc32c', crc, data.ctypes.data, data.nbytes, 0))
    if isinstance(data, memoryview):
        data = np.frombuffer(data, dtype=np.uint8)
        return int(LIB.query('merlot_crc32c', crc, data.ctypes.data, data.nbytes, 0))
    return int(LIB.query('merlot_crc32c', crc, bytes(data), len(data), 0))


def mask_crc(c):
    """crc32c::Mask (tensorflow/core/lib/hash/crc32c.h): rotate right 15, add a constant."""
    return ((((c >> 15) | (c << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


def unmask_crc(m):
    r = (m - _MASK_DELTA) & 0xffffffff
    return ((r >> 17) | (r << 15)) & 0xffffffff


# ---- varints and the two protos --------------------------------------------------------------------------------------
def _put_varint(out, v):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError('truncated varint')
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise CheckpointError('varint too long')


def _proto_fields(buf, copy=True):
    """-> [(field number, wire type, value)] of one protobuf message (varint / fixed64 / bytes / fixed32 only).
    copy=False on a memoryview returns length-delimited payloads as zero-copy slices (the TFRecord path: MB-sized
    Examples nested four levels deep)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n]) if copy else buf[pos:pos + n]
            if len(v) != n:
                raise CheckpointError('truncated proto field')
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError(f'unsupported protobuf wire type {wt}')
        out.append((f, wt, v))
    return out


def _signed(v):
    return v - (1 << 64) if v >> 63 else v


Entry = collections.namedtuple('Entry', 'dtype shape shard_id offset size crc32c sliced')


def _parse_entry(buf):
    dtype, shape, shard, off, size, crc, sliced = 0, [], 0, 0, 0, None, False
    for f, wt, v in _proto_fields(buf):
        if f == 1:
            dtype = v
        elif f == 2:                                    # TensorShapeProto {repeated Dim dim = 2 {int64 size = 1}}
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:
                    d = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1:
                            d = _signed(v3)
                    shape.append(d)
                elif f2 == 3 and v2:
                    raise CheckpointError('unknown-rank tensor in checkpoint')
        elif f == 3:
            shard = v
        elif f == 4:
            off = v
        elif f == 5:
            size = v
        elif f == 6:
            crc = v
        elif f == 7:
            sliced = True
    return Entry(dtype, tuple(shape), shard, off, size, crc, sliced)


def _encode_entry(dtype, shape, offset, size, crc_masked):
    out = bytearray()
    out.append(1 << 3)
    _put_varint(out, dtype)
    dims = bytearray()
    for d in shape:
        dim = bytearray()
        dim.append(1 << 3)
        _put_varint(dim, int(d))
        dims.append((2 << 3) | 2)
        _put_varint(dims, len(dim))
        dims += dim
    out.append((2 << 3) | 2)
    _put_varint(out, len(dims))
    out += dims
    if offset:                                           # proto3: zero-valued scalars are not serialised
        out.append(4 << 3)
        _put_varint(out, offset)
    if size:
        out.append(5 << 3)
        _put_varint(out, size)
    out.append((6 << 3) | 5)
    out += struct.pack('<I', crc_masked)
    return bytes(out)


def _encode_header(num_shards=1):
    out = bytearray()
    out.append(1 << 3)
    _put_varint(out, num_shards)
    version = bytes([1 << 3, 1])                         # VersionDef {producer = 1}; endianness LITTLE = 0 is omitted
    out.append((3 << 3) | 2)
    _put_varint(out, len(version))
    out += version
    return bytes(out)


# ---- SSTable ----------------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify):
    if offset + size + 5 > len(buf):
        raise CheckpointError('block handle points past the end of the index file')
    body = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        want = unmask_crc(struct.unpack_from('<I', buf, offset + size + 1)[0])
        if crc32c(bytes(buf[offset:offset + size + 1])) != want:
            raise CheckpointError(f'index block at {offset}: checksum mismatch')
    if ctype != 0:
        raise CheckpointError(f'index block at {offset}: compression type {ctype} (snappy) is not supported')
    return body


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError('block too small')
    nrestart = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * nrestart
    if limit < 0:
        raise CheckpointError('bad restart count')
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise CheckpointError('corrupt block entry')
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _read_table(buf, verify=True):
    """-> ordered [(key bytes, value bytes)] of an SSTable image."""
    if len(buf) < 48:
        raise CheckpointError('index file shorter than a table footer')
    footer = buf[-48:]
    if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
        raise CheckpointError('not a TF checkpoint index (bad table magic)')
    pos = 0
    _, pos = _get_varint(footer, pos)                    # metaindex handle
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


class _BlockBuilder(object):
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last, self.ri = bytearray(), [0], 0, b'', restart_interval

    def add(self, key, value):
        shared = 0
        if self.count < self.ri:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        _put_varint(self.buf, shared)
        _put_varint(self.buf, len(key) - shared)
        _put_varint(self.buf, len(value))
        self.buf += key[shared:]
        self.buf += value
        self.last, self.count = key, self.count + 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))
        return out


def _write_table(items, block_size=262144):
    """sorted [(key, value)] -> SSTable image (no compression, no filter; index keys = last key of each block)."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)
        out.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        h = bytearray()
        _put_varint(h, off)
        _put_varint(h, len(block))
        return bytes(h)

    index, bb = _BlockBuilder(restart_interval=1), _BlockBuilder()
    prev = None
    for k, v in items:
        if prev is not None and k <= prev:
            raise CheckpointError('table keys must be strictly increasing')
        bb.add(k, v)
        prev = k
        if bb.size() >= block_size:
            index.add(k, emit(bb.finish()))
            bb = _BlockBuilder()
    if not bb.empty():
        index.add(prev, emit(bb.finish()))
    meta = emit(_BlockBuilder().finish())
    idx = emit(index.finish())
    footer = bytearray(meta + idx)
    footer.extend(b'\x00' * (40 - len(footer)))
    footer.extend(struct.pack('<Q', TABLE_MAGIC))
    out.extend(footer)
    return bytes(out)


# ---- reader -----------------------------------------------------------------------------------------------------------
def resolve_prefix(path):
    """a checkpoint prefix, or a directory with a `checkpoint` state file (tf.train.latest_checkpoint)."""
    if os.path.isdir(path):
        latest = latest_checkpoint(path)
        if latest is None:
            raise CheckpointError(f'no checkpoint state in {path}')
        return latest
    return path


def latest_checkpoint(model_dir):
    state = os.path.join(model_dir, 'checkpoint')
    if not os.path.exists(state):
        return None
    for line in open(state):
        m = re.match(r'^model_checkpoint_path:\s*"(.*)"\s*$', line)
        if m:
            p = m.group(1)
            return p if os.path.isabs(p) else os.path.join(model_dir, p)
    return None


class CheckpointReader(object):
    """tf.train.load_checkpoint(prefix): `.get_variable_to_shape_map()`, `.get_tensor(name)`."""

    def __init__(self, prefix, verify=True):
        self.prefix = resolve_prefix(prefix)
        self.verify = verify
        idx = self.prefix + '.index'
        if not os.path.exists(idx):
            raise CheckpointError(f'{idx} not found')
        items = _read_table(open(idx, 'rb').read(), verify)
        if not items or items[0][0] != b'':
            raise CheckpointError('checkpoint index has no bundle header')
        self.num_shards, endian = 1, 0
        for f, _, v in _proto_fields(items[0][1]):
            if f == 1:
                self.num_shards = v
            elif f == 2:
                endian = v
        if endian != 0:
            raise CheckpointError('big-endian checkpoint')
        self.entries = collections.OrderedDict((k.decode('utf-8'), _parse_entry(v)) for k, v in items[1:])
        self._maps = {}

    def _shard(self, i):
        m = self._maps.get(i)
        if m is None:
            path = f'{self.prefix}.data-{i:05d}-of-{self.num_shards:05d}'
            if not os.path.exists(path):
                raise CheckpointError(f'{path} not found')
            m = np.memmap(path, dtype=np.uint8, mode='r') if os.path.getsize(path) else np.zeros(0, np.uint8)
            self._maps[i] = m
        return m

    def get_variable_to_shape_map(self):
        return {k: list(e.shape) for k, e in self.entries.items()}

    def get_variable_to_dtype_map(self):
        return {k: (DT[e.dtype][0] if e.dtype in DT else f'DT_{e.dtype}') for k, e in self.entries.items()}

    def has_tensor(self, name):
        return name in self.entries

    def get_tensor(self, name):
        """-> torch tensor (CPU) in the stored dtype (bf16 stays bf16)."""
        e = self.entries.get(name)
        if e is None:
            raise KeyError(f'{name} not in checkpoint {self.prefix}')
        if e.sliced:
            raise CheckpointError(f'{name}: partitioned (sliced) variables are not supported')
        if e.dtype not in DT:
            raise CheckpointError(f'{name}: unsupported dtype enum {e.dtype}' + (' (DT_STRING)' if e.dtype == DT_STRING else ''))
        tdt = DT[e.dtype][1]
        n = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        if n * torch.empty((), dtype=tdt).element_size() != e.size:
            raise CheckpointError(f'{name}: {e.size} bytes stored for shape {e.shape} {DT[e.dtype][0]}')
        data = self._shard(e.shard_id)
        if e.offset + e.size > data.shape[0]:
            raise CheckpointError(f'{name}: data file too short')
        raw = np.array(data[e.offset:e.offset + e.size])            # copy out of the map
        if self.verify and e.crc32c is not None and mask_crc(crc32c(raw)) != e.crc32c:
            raise CheckpointError(f'{name}: tensor checksum mismatch')
        if e.size == 0:
            return torch.empty(e.shape, dtype=tdt)
        return torch.frombuffer(raw, dtype=tdt).reshape(e.shape)


def list_variables(ckpt):
    """tf.train.list_variables: [(name, shape)] sorted by name."""
    r = CheckpointReader(ckpt, verify=False)
    return sorted((k, list(e.shape)) for k, e in r.entries.items())


def load_variable(ckpt, name):
    return CheckpointReader(ckpt).get_tensor(name)


# ---- writer -----------------------------------------------------------------------------------------------------------
def write_checkpoint(prefix, tensors, block_size=262144, update_state=True):
    """{name: torch tensor / ndarray / python scalar} -> `<prefix>.index` + `<prefix>.data-00000-of-00001`
    (+ the `checkpoint` state file of the directory, as tf.train.Saver keeps it)."""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    items = [(b'', _encode_header(1))]
    off = 0
    with open(prefix + '.data-00000-of-00001.tmp', 'wb') as f:
        for name in sorted(tensors, key=lambda s: s.encode('utf-8')):
            t = tensors[name]
            t = t.detach().cpu().contiguous() if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t))
            if t.dtype not in _DT_OF_TORCH:
                raise CheckpointError(f'{name}: dtype {t.dtype} has no TF equivalent here')
            raw = t.reshape(-1).view(torch.uint8).numpy() if t.numel() else np.zeros(0, np.uint8)
            f.write(raw.tobytes())
            items.append((name.encode('utf-8'),
                          _encode_entry(_DT_OF_TORCH[t.dtype], tuple(t.shape), off, raw.nbytes, mask_crc(crc32c(raw)))))
            off += raw.nbytes
    with open(prefix + '.index.tmp', 'wb') as f:
        f.write(_write_table(items, block_size))
    os.replace(prefix + '.data-00000-of-00001.tmp', prefix + '.data-00000-of-00001')
    os.replace(prefix + '.index.tmp', prefix + '.index')
    if update_state:
        # the `checkpoint` state file of tf.train.Saver: newest prefix + every prefix still on disk (the reference's
        # RunConfig keeps them all: keep_checkpoint_max=None, model/train.py), written atomically
        base = os.path.basename(prefix)
        state = os.path.join(d or '.', 'checkpoint')
        older = []
        if os.path.exists(state):
            for line in open(state):
                m = re.match(r'^all_model_checkpoint_paths: "(.*)"\s*$', line)
                if m and m.group(1) != base and os.path.exists(os.path.join(d or '.', m.group(1) + '.index')):
                    older.append(m.group(1))
        with open(state + '.tmp', 'w') as f:
            f.write(f'model_checkpoint_path: "{base}"\n')
            for b in older + [base]:
                f.write(f'all_model_checkpoint_paths: "{b}"\n')
        os.replace(state + '.tmp', state)


# ---- the reference's use of checkpoints -------------------------------------------------------------------------------
def get_assignment_map_from_checkpoint(tvar_names, init_checkpoint, reference_name_transform=None):
    """utils/model_utils.py:388-413 on names: -> (assignment_map {ckpt name: variable name}, initialized_variable_names)."""
    name_to_variable = collections.OrderedDict()
    for name in tvar_names:
        m = re.match("^(.*):\\d+$", name)
        if m is not None:
            name = m.group(1)
        name_to_variable[name] = name
    assignment_map = collections.OrderedDict()
    initialized_variable_names = {}
    for name, _ in list_variables(init_checkpoint):
        rhs_name = name if reference_name_transform is None else reference_name_transform(name)
        if rhs_name not in name_to_variable:
            continue
        assignment_map[name] = rhs_name
        initialized_variable_names[rhs_name] = 1
        initialized_variable_names[rhs_name + ":0"] = 1
    return assignment_map, initialized_variable_names


def variable_names(store, optimizer=None):
    """the TF variable names this build's state corresponds to: trainable variables (tf.trainable_variables()) and,
    with an optimizer, its `<name>/adam_m`, `<name>/adam_v` slots (utils/optimization.py:372-383) -- i.e. the
    GLOBAL_VARIABLES minus global_step that model/modeling.py:716 initialises when training."""
    names = [t for n in store.names() for t in store.tf_names(n)]
    if optimizer is not None:
        names = names + [t + s for t in names for s in ('/adam_m', '/adam_v')]
    return names


def _check_slot_dtypes(sub, flat, slot, where):
    """Adam slots are stored in the optimizer's own dtype (bf16 m; bf16 v with the update's sign bit folded in under
    use_bfloat16_adam, utils/optimization.py:267-288).  Loading them into an optimizer that keeps the other encoding would
    turn the sign-encoded v into negative second moments (NaN at the first sqrt) or silently drop the encoding."""
    for name, t in sub.items():
        have = t.dtype if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t)).dtype
        if have != flat.dtype:
            raise CheckpointError(f'{where}: {name} is stored as {have} but this optimizer keeps its {slot[1:]} slots as '
                                  f'{flat.dtype} (optimizer.use_bfloat16_adam differs from the run that wrote the checkpoint)')


def init_from_checkpoint(store, init_checkpoint, optimizer=None, reference_name_transform=None):
    """model/modeling.py:724-738: every variable of the model (and, when training, of the optimizer) that the
    checkpoint also holds under the same name is overwritten; the rest keep their initial values; `global_step` is
    never taken (:716).  -> initialized_variable_names."""
    reader = CheckpointReader(init_checkpoint)
    amap, initialized = get_assignment_map_from_checkpoint(variable_names(store, optimizer), reader.prefix,
                                                           reference_name_transform)
    loaded = {dst: reader.get_tensor(src) for src, dst in amap.items()}
    store.load_tf_weights({k: v for k, v in loaded.items() if not k.endswith(('/adam_m', '/adam_v'))}, strict=False)
    if optimizer is not None:
        for slot, flat in (('/adam_m', optimizer.m), ('/adam_v', optimizer.v)):
            sub = {k: v for k, v in loaded.items() if k.endswith(slot)}
            if sub:
                _check_slot_dtypes(sub, flat, slot, reader.prefix)
                store.load_tf_weights(sub, strict=False, getter=lambda n, flat=flat: store.view(flat, n), suffix=slot)
    return initialized


def save_checkpoint(model_dir, store, optimizer=None, global_step=None):
    """what the reference's Estimator writes to `output_dir`: `model.ckpt-<global_step>` holding every variable, the
    optimizer slots in their stored dtype (bf16 m, sign-encoded bf16 v under use_bfloat16_adam) and `global_step`."""
    step = int(optimizer.step_count if (global_step is None and optimizer is not None) else (global_step or 0))
    tensors = dict(store.export_tf_weights())
    if optimizer is not None:
        for slot, flat in (('/adam_m', optimizer.m), ('/adam_v', optimizer.v)):
            for k, v in store._export(lambda n, flat=flat: store.view(flat, n), keep_dtype=True).items():
                tensors[k + slot] = v
    tensors['global_step'] = torch.tensor(step, dtype=torch.int64)
    prefix = os.path.join(model_dir, f'model.ckpt-{step}')
    write_checkpoint(prefix, tensors)
    return prefix


def restore_checkpoint(model_dir_or_prefix, store, optimizer=None):
    """Estimator auto-resume: ALL state comes from the checkpoint (missing variables are an error), including
    global_step, which positions the learning-rate schedule (utils/optimization.py:94-115).  -> global_step."""
    reader = CheckpointReader(model_dir_or_prefix)
    names = variable_names(store, optimizer)
    missing = [n for n in names if not reader.has_tensor(n)]
    if missing:
        raise CheckpointError(f'{reader.prefix} lacks {len(missing)} variables, e.g. {missing[:3]}')
    loaded = {n: reader.get_tensor(n) for n in names}
    store.load_tf_weights({k: v for k, v in loaded.items() if not k.endswith(('/adam_m', '/adam_v'))}, strict=True)
    step = int(reader.get_tensor('global_step')) if reader.has_tensor('global_step') else 0
    if optimizer is not None:
        for slot, flat in (('/adam_m', optimizer.m), ('/adam_v', optimizer.v)):
            sub = {k: v for k, v in loaded.items() if k.endswith(slot)}
            _check_slot_dtypes(sub, flat, slot, reader.prefix)
            store.load_tf_weights(sub, strict=True, getter=lambda n, flat=flat: store.view(flat, n), suffix=slot)
        optimizer.step_count = step
    return step
