"""TEST INFRASTRUCTURE ONLY — an eager, torch-CPU-backed stand-in for the ~120 `tensorflow` 1.15 symbols that the
reference's hot-path modules use (SURVEY.md §8c "optional stronger oracle").

Purpose: `tests/golden/make_reference_golden.py` installs this module as `tensorflow` in `sys.modules`, imports the
UNMODIFIED reference modules from /root/reference (`model/modeling.py`, `utils/transformer.py`,
`utils/vision_transformer.py`, `utils/model_utils.py`, `utils/optimization.py`) and runs them.  What executes is the
reference's own control flow: variable scoping and naming, every reshape / tile / transpose / concat order, the
masking logic, loss assembly and the optimizer update.  What is OURS is the semantics of each primitive below
(`tf.reshape` = row-major reshape, `tf.layers.dense` = x·kernel + bias, `tf.math.top_k` = stable descending sort …),
written from TensorFlow 1.15's documented behaviour.  A fixture produced this way therefore pins the restatement in
`oracle/merlot_oracle.py` against the reference's program structure, not against TensorFlow's kernels; DESIGN.md §5
says so.  Nothing under merlot_amd/ imports this file, and it never travels to the GPU box in use (the fixtures do).

Randomness: every random op draws from one seeded torch generator and is logged in `STATE.draws` in call order, so
the same draws can be handed to the oracle / the HIP path as explicit noise.
"""
import contextlib
import math
import sys
import threading
import types

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------------------------
# tensor type: torch.Tensor + the TF-1 shape protocol (`x.shape.as_list()`, `x.shape[-1].value`, `x.shape.ndims`)
# ----------------------------------------------------------------------------------------------------------------
class Dim(int):
    @property
    def value(self):
        return int(self)


class TensorShape(tuple):
    def as_list(self):
        return [int(d) for d in self]

    @property
    def ndims(self):
        return len(self)

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return TensorShape(r) if isinstance(i, slice) else r


class T(torch.Tensor):
    @property
    def shape(self):
        return TensorShape(Dim(d) for d in self.size())

    def get_shape(self):
        return self.shape

    @property
    def name(self):
        return getattr(self, '_tf_name', None)

    @name.setter
    def name(self, value):
        self._tf_name = value

    def __getitem__(self, idx):
        """numpy/TF indexing incl. `[:, ::-1]` (torch rejects negative steps)."""
        if not isinstance(idx, tuple):
            idx = (idx,)
        flips, out, dim = [], [], 0
        n_real = sum(1 for i in idx if i is not None and i is not Ellipsis)
        for i in idx:
            if i is None:
                out.append(None)
                continue
            if i is Ellipsis:
                dim += self.dim() - n_real
                out.append(i)
                continue
            if isinstance(i, slice) and i.step is not None and i.step < 0:
                assert i.step == -1 and i.start is None and i.stop is None, "only [::-1] is supported"
                flips.append(dim)
                i = slice(None)
            if isinstance(i, torch.Tensor) and i.dtype in (torch.int32, torch.int16, torch.uint8):
                i = i.long()
            out.append(i)
            dim += 1
        src = torch.flip(self, flips) if flips else self
        return torch.Tensor.__getitem__(src, tuple(out))

    # tf tensors are immutable: `x += y` rebinds, it never writes in place (autograd needs the old value)
    def __iadd__(self, other):
        return self + other

    def __isub__(self, other):
        return self - other

    def __imul__(self, other):
        return self * other

    def __itruediv__(self, other):
        return self / other

    def assign(self, value):
        if STATE.num_shards > 1 and STATE.replica != 0:
            return self                 # simulated replicas share ONE copy of every variable: replica 0 writes it
        with torch.no_grad():
            self.data.copy_(_t(value).to(self.dtype).reshape(self.size()))
        return self

    def numpy_(self):
        return self.detach().cpu().numpy()

    def set_shape(self, shape):
        assert [int(d) for d in shape] == list(self.size()), (shape, self.size())


def _w(x):
    return x.as_subclass(T) if isinstance(x, torch.Tensor) and not isinstance(x, T) else x


def _t(x, dtype=None):
    """anything -> T (python scalars / lists / numpy become fp32 or int32 like tf.convert_to_tensor)."""
    if isinstance(x, torch.Tensor):
        out = x
    elif isinstance(x, (bool, np.bool_)):
        out = torch.tensor(bool(x))
    elif isinstance(x, (int, np.integer)):
        out = torch.tensor(int(x), dtype=torch.int32)
    elif isinstance(x, (float, np.floating)):
        out = torch.tensor(float(x), dtype=torch.float32)
    elif isinstance(x, (list, tuple)) and any(isinstance(e, torch.Tensor) for e in _flatten(x)):
        out = torch.stack([_t(e) for e in x], 0)
    else:
        arr = np.asarray(x)
        if arr.dtype == np.float64:
            arr = arr.astype(np.float32)
        elif arr.dtype == np.int64:
            arr = arr.astype(np.int32)
        out = torch.from_numpy(arr)
    if dtype is not None and out.dtype != dtype:
        out = out.to(dtype)
    return _w(out)


def _flatten(x):
    for e in x:
        if isinstance(e, (list, tuple)):
            yield from _flatten(e)
        else:
            yield e


def _shape(s):
    if isinstance(s, torch.Tensor):
        return [int(v) for v in s.tolist()]
    if isinstance(s, (int, np.integer)):
        return [int(s)]
    return [int(v) for v in s]


def _axes(axis):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple)):
        return [int(a) for a in axis]
    return int(axis)


# ----------------------------------------------------------------------------------------------------------------
# global state: variables, scopes, random draws, simulated replicas
# ----------------------------------------------------------------------------------------------------------------
class _State(object):
    def __init__(self):
        self.reset()

    def reset(self, seed=0, injected=None, num_shards=1):
        self.vars = {}                  # name -> T (leaf, requires_grad for trainable)
        self.trainable = []             # creation order, like tf.trainable_variables()
        self.created_by_initializer = []
        self.injected = dict(injected or {})
        self.seed = seed
        self._gens = {}                 # one generator and one draw log per simulated replica
        self._draws = {}                # replica -> [(kind, tensor)] in call order
        self.default_names = {}
        self.num_shards = num_shards
        self.local = threading.local()
        self.barrier = threading.Barrier(num_shards) if num_shards > 1 else None
        self.deposit = {}
        self.lock = threading.Lock()
        self.gradients_log = None

    @property
    def gen(self):
        r = self.replica
        if r not in self._gens:
            self._gens[r] = torch.Generator().manual_seed(self.seed + 1000 * r)
        return self._gens[r]

    @property
    def draws(self):
        return self._draws.setdefault(self.replica, [])

    def draws_of(self, replica):
        return self._draws.get(replica, [])

    # per-thread (= per simulated replica) scope stack
    @property
    def scopes(self):
        if not hasattr(self.local, 'scopes'):
            self.local.scopes = []
        return self.local.scopes

    @property
    def replica(self):
        return getattr(self.local, 'replica', 0)

    def scope_name(self):
        return '/'.join(s['name'] for s in self.scopes if s['name'])


STATE = _State()


class _VarScope(object):
    def __init__(self, name):
        self.name = name


@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None, custom_getter=None, reuse=None, **_):
    name = name_or_scope.name if isinstance(name_or_scope, _VarScope) else name_or_scope
    if name is None:
        # tf uniquifies default_name within the parent scope: LayerNorm, LayerNorm_1, ...
        key = (STATE.replica, STATE.scope_name(), default_name)
        k = STATE.default_names.get(key, 0)
        STATE.default_names[key] = k + 1
        name = default_name if k == 0 else f'{default_name}_{k}'
    STATE.scopes.append({'name': name, 'getter': custom_getter})
    try:
        yield _VarScope(STATE.scope_name())
    finally:
        STATE.scopes.pop()


def get_variable_scope():
    return _VarScope(STATE.scope_name())


def _raw_get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **_):
    dtype = dtype or torch.float32
    with STATE.lock:
        if name in STATE.vars:
            v = STATE.vars[name]
            if shape is not None and list(v.size()) != _shape(shape):
                raise ValueError(f"variable {name}: shape {list(v.size())} vs requested {_shape(shape)}")
            return v
        if name in STATE.injected:
            val = torch.as_tensor(np.asarray(STATE.injected[name])).to(dtype).clone()
            if shape is not None and list(val.size()) != _shape(shape):
                raise ValueError(f"injected {name}: shape {list(val.size())} vs requested {_shape(shape)}")
        else:
            if initializer is None:
                initializer = glorot_uniform_initializer()
            val = initializer(_shape(shape), dtype)
            STATE.created_by_initializer.append(name)
        v = _w(val.detach().clone())
        if trainable and v.is_floating_point():
            v.requires_grad_(True)
        v.name = name + ':0'
        v.trainable = trainable
        STATE.vars[name] = v
        if trainable:
            STATE.trainable.append(v)
        return v


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **kw):
    dtype = dtype or torch.float32
    full = '/'.join([p for p in (STATE.scope_name(), name) if p])
    getters = [s['getter'] for s in STATE.scopes if s['getter'] is not None]

    def base(name_, shape=None, dtype=None, initializer=None, trainable=True, **kw_):
        return _raw_get_variable(name_, shape=shape, dtype=dtype, initializer=initializer, trainable=trainable)

    getter = base
    for g in getters:                   # outermost first -> innermost ends up outermost wrapper, as in TF
        getter = (lambda inner, cg: (lambda n, **k: cg(inner, n, **k)))(getter, g)
    return getter(full, shape=shape, dtype=dtype, initializer=initializer, trainable=trainable)


def trainable_variables():
    return list(STATE.trainable)


# initializers -------------------------------------------------------------------------------------------------
def truncated_normal_initializer(mean=0.0, stddev=1.0, **_):
    def init(shape, dtype=torch.float32):
        x = torch.empty(shape, dtype=torch.float32)
        torch.nn.init.trunc_normal_(x, mean=mean, std=stddev, a=mean - 2 * stddev, b=mean + 2 * stddev,
                                    generator=STATE.gen)
        return x.to(dtype)
    return init


def constant_initializer(value=0.0, **_):
    return lambda shape, dtype=torch.float32: torch.full(shape, float(value), dtype=dtype)


def zeros_initializer(**_):
    return lambda shape, dtype=torch.float32: torch.zeros(shape, dtype=dtype)


def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def variance_scaling_initializer(scale=1.0, mode='fan_in', distribution='truncated_normal', **_):
    def init(shape, dtype=torch.float32):
        fan_in, fan_out = _fans(shape)
        n = {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': (fan_in + fan_out) / 2.0}[mode]
        std = math.sqrt(scale / max(1.0, n)) / .87962566103423978
        return truncated_normal_initializer(stddev=std)(shape, dtype)
    return init


def glorot_uniform_initializer(**_):
    def init(shape, dtype=torch.float32):
        fan_in, fan_out = _fans(shape)
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return ((torch.rand(shape, generator=STATE.gen) * 2 - 1) * lim).to(dtype)
    return init


# ----------------------------------------------------------------------------------------------------------------
# ops
# ----------------------------------------------------------------------------------------------------------------
def cast(x, dtype=None, **_):
    x = _t(x)
    if dtype in (torch.int32, torch.int64) and x.is_floating_point():
        return _w(torch.trunc(x).to(dtype))
    return _w(x.to(dtype))


def reshape(tensor, shape, name=None):
    return _w(_t(tensor).reshape(_shape(shape)))


def transpose(a, perm=None, **_):
    a = _t(a)
    return _w(a.permute(*[int(p) for p in perm]) if perm is not None else a.permute(*reversed(range(a.dim()))))


def tile(input, multiples, name=None):
    return _w(_t(input).repeat(*_shape(multiples)))


def concat(values, axis, name=None):
    return _w(torch.cat([_t(v) for v in values], dim=int(axis)))


def stack(values, axis=0, name=None):
    return _w(torch.stack([_t(v) for v in values], dim=int(axis)))


def unstack(value, num=None, axis=0, name=None):
    return [_w(v) for v in torch.unbind(_t(value), dim=int(axis))]


def tf_slice(input_, begin, size, name=None):
    x = _t(input_)
    idx = tuple(slice(int(b), None if int(s) == -1 else int(b) + int(s)) for b, s in zip(begin, size))
    return x[idx]


def tf_range(start, limit=None, delta=1, dtype=None, name=None):
    if limit is None:
        start, limit = 0, start
    return _w(torch.arange(int(start), int(limit), int(delta), dtype=dtype or torch.int32))


def ones(shape, dtype=torch.float32, name=None):
    return _w(torch.ones(_shape(shape), dtype=dtype))


def zeros(shape, dtype=torch.float32, name=None):
    return _w(torch.zeros(_shape(shape), dtype=dtype))


def zeros_like(x, dtype=None, **_):
    return _w(torch.zeros_like(_t(x), dtype=dtype))


def ones_like(x, dtype=None, **_):
    return _w(torch.ones_like(_t(x), dtype=dtype))


def fill(dims, value, name=None):
    v = _t(value)
    return _w(torch.full(_shape(dims), v.item(), dtype=v.dtype))


def constant(value, dtype=None, shape=None, name=None):
    out = _t(value, dtype)
    if shape is not None:
        out = out.expand(_shape(shape)).clone() if out.dim() == 0 else out.reshape(_shape(shape))
    return _w(out)


def identity(x, name=None):
    return _t(x)


def tf_shape(x, **_):
    return _w(torch.tensor(list(_t(x).size()), dtype=torch.int32))


def matmul(a, b, transpose_a=False, transpose_b=False, **_):
    a, b = _t(a), _t(b)
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return _w(torch.matmul(a, b))


def _binary(fn):
    def op(x, y, name=None):
        x, y = _t(x), _t(y)
        return _w(fn(x, y))
    return op


multiply = _binary(torch.mul)
add = _binary(torch.add)
subtract = _binary(torch.sub)
equal = _binary(torch.eq)
not_equal = _binary(torch.ne)
less = _binary(torch.lt)
less_equal = _binary(torch.le)
greater = _binary(torch.gt)
greater_equal = _binary(torch.ge)
logical_and = _binary(torch.logical_and)
logical_or = _binary(torch.logical_or)
minimum = _binary(torch.minimum)
maximum = _binary(torch.maximum)
floor_div = _binary(lambda x, y: torch.div(x, y, rounding_mode='floor'))
mod = _binary(torch.remainder)
tf_pow = _binary(lambda x, y: torch.pow(x.float() if not x.is_floating_point() else x, y))


def logical_not(x, name=None):
    return _w(torch.logical_not(_t(x)))


def _unary(fn):
    def op(x, name=None):
        return _w(fn(_t(x)))
    return op


log = _unary(torch.log)
log1p = _unary(torch.log1p)
exp = _unary(torch.exp)
sqrt = _unary(torch.sqrt)
rsqrt = _unary(torch.rsqrt)
square = _unary(torch.square)
tf_abs = _unary(torch.abs)
sign = _unary(torch.sign)
erf = _unary(torch.erf)
lgamma = _unary(torch.lgamma)
tanh = _unary(torch.tanh)
relu = _unary(torch.relu)


def _reduce(fn, boolean=False):
    def op(input_tensor, axis=None, keepdims=False, keep_dims=None, name=None, **_):
        x = _t(input_tensor)
        kd = bool(keepdims if keep_dims is None else keep_dims)
        ax = _axes(axis)
        if ax is None:
            ax = list(range(x.dim()))
        return _w(fn(x, ax, kd))
    return op


reduce_sum = _reduce(lambda x, a, k: torch.sum(x, dim=a, keepdim=k))
reduce_mean = _reduce(lambda x, a, k: torch.mean(x, dim=a, keepdim=k))
reduce_max = _reduce(lambda x, a, k: torch.amax(x, dim=a, keepdim=k))
reduce_min = _reduce(lambda x, a, k: torch.amin(x, dim=a, keepdim=k))
reduce_any = _reduce(lambda x, a, k: torch.sum(x.to(torch.int32), dim=a, keepdim=k) > 0)
reduce_all = _reduce(lambda x, a, k: torch.sum((~x).to(torch.int32), dim=a, keepdim=k) == 0)


def add_n(inputs, name=None):
    out = _t(inputs[0])
    for v in inputs[1:]:
        out = out + _t(v)
    return _w(out)


def argmax(input, axis=None, output_type=torch.int64, **_):
    return _w(torch.argmax(_t(input), dim=int(axis if axis is not None else 0)).to(output_type))


def top_k(input, k=1, sorted=True, name=None):
    """values descending; equal values keep the lower index first (TF's documented tie rule)."""
    x = _t(input)
    vals, idx = torch.sort(x, dim=-1, descending=True, stable=True)
    return _w(vals[..., :k]), _w(idx[..., :k].to(torch.int32))


def tf_sort(values, axis=-1, direction='ASCENDING', name=None):
    out, _ = torch.sort(_t(values), dim=int(axis), descending=(direction == 'DESCENDING'), stable=True)
    return _w(out)


def argsort(values, axis=-1, direction='ASCENDING', stable=False, name=None):
    return _w(torch.argsort(_t(values), dim=int(axis), descending=(direction == 'DESCENDING'), stable=True)
              .to(torch.int32))


def one_hot(indices, depth, on_value=None, off_value=None, axis=None, dtype=None, name=None):
    idx = _t(indices).long()
    return _w(F.one_hot(idx, int(depth)).to(dtype or torch.float32))


def gather(params, indices, validate_indices=None, name=None, axis=None, batch_dims=0):
    p, idx = _t(params), _t(indices).long()
    if batch_dims == 0:
        ax = int(axis or 0)
        out = torch.index_select(p, ax, idx.reshape(-1))
        return _w(out.reshape(list(p.shape[:ax]) + list(idx.shape) + list(p.shape[ax + 1:])))
    assert batch_dims == 1 and p.dim() >= 2, "only batch_dims in (0, 1)"
    # out[b, i...] = params[b, indices[b, i...]]
    b = p.size(0)
    flat = idx.reshape(b, -1)
    rest = p.shape[2:]
    g = torch.gather(p, 1, flat.reshape(b, -1, *([1] * len(rest))).expand(b, flat.size(1), *rest))
    return _w(g.reshape(list(idx.shape) + list(rest)))


def embedding_lookup(params, ids, **_):
    return gather(params, ids)


def scatter_nd(indices, updates, shape, name=None):
    upd = _t(updates)
    out = torch.zeros(_shape(shape), dtype=upd.dtype)
    idx = [[int(v) for v in row] for row in indices]
    pieces = list(torch.unbind(out, 0))
    for row, u in zip(idx, torch.unbind(upd, 0)):
        assert len(row) == 1
        pieces[row[0]] = pieces[row[0]] + u
    return _w(torch.stack(pieces, 0))


def where_v2(condition, x=None, y=None, name=None):
    return _w(torch.where(_t(condition), _t(x), _t(y)))


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    return _w(torch.clamp(_t(t), clip_value_min, clip_value_max))


def softmax(logits, axis=-1, name=None):
    return _w(torch.softmax(_t(logits), dim=int(axis)))


def log_softmax(logits, axis=-1, name=None):
    return _w(torch.log_softmax(_t(logits), dim=int(axis)))


def moments(x, axes, keep_dims=False, keepdims=None, **_):
    x = _t(x)
    kd = bool(keep_dims if keepdims is None else keepdims)
    ax = _axes(axes)
    mean = torch.mean(x, dim=ax, keepdim=True)
    var = torch.mean(torch.square(x - mean), dim=ax, keepdim=True)
    if not kd:
        mean, var = mean.squeeze(ax), var.squeeze(ax)
    return _w(mean), _w(var)


def l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
    x = _t(x)
    ax = _axes(axis if axis is not None else dim)
    sq = torch.sum(torch.square(x), dim=ax, keepdim=True)
    return _w(x * torch.rsqrt(torch.clamp(sq, min=epsilon)))


def l2_loss(t, name=None):
    return _w(torch.sum(torch.square(_t(t))) / 2)


def bias_add(value, bias, **_):
    return _w(_t(value) + _t(bias))


def tf_dropout(x, keep_prob=None, noise_shape=None, seed=None, name=None, rate=None):
    x = _t(x)
    rate = (1.0 - keep_prob) if rate is None else rate
    u = torch.rand(x.shape, generator=STATE.gen)
    STATE.draws.append(('dropout', u))
    keep = (u >= rate).to(x.dtype)
    return _w(x * keep / (1.0 - rate))


def avg_pool2d(value, ksize, strides, padding, data_format='NHWC', name=None):
    assert data_format == 'NHWC'
    x = _t(value).permute(0, 3, 1, 2)
    if padding == 'SAME':       # TF averages over the valid elements only; with even sizes and k == s nothing is padded
        assert int(x.shape[2]) % int(strides) == 0 and int(x.shape[3]) % int(strides) == 0 and ksize == strides
    return _w(F.avg_pool2d(x, kernel_size=ksize, stride=strides).permute(0, 2, 3, 1))


def tf_pad(tensor, paddings, mode='CONSTANT', name=None, constant_values=0):
    x = _t(tensor)
    flat = []
    for lo, hi_ in reversed([list(p) for p in paddings]):
        flat += [int(lo), int(hi_)]
    return _w(F.pad(x, flat, value=constant_values))


def nn_conv2d(input=None, filter=None, strides=None, padding='SAME', data_format='NHWC', name=None, filters=None, **_):
    """tf.nn.conv2d, NHWC x HWIO.  SAME = TF's rule: total pad = max((ceil(in/s)-1)*s + k - in, 0), extra on the high side."""
    assert data_format == 'NHWC'
    x, w = _t(input), _t(filter if filter is not None else filters)
    sh, sw = (strides[1], strides[2]) if len(strides) == 4 else (strides[0], strides[-1])
    kh, kw = int(w.shape[0]), int(w.shape[1])
    xc = x.permute(0, 3, 1, 2)
    if padding.upper() == 'SAME':
        ih, iw = int(x.shape[1]), int(x.shape[2])
        ph = max((-(-ih // sh) - 1) * sh + kh - ih, 0)
        pw = max((-(-iw // sw) - 1) * sw + kw - iw, 0)
        xc = F.pad(xc, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    out = F.conv2d(xc, w.permute(3, 2, 0, 1), None, stride=(sh, sw))
    return _w(out.permute(0, 2, 3, 1))


class Conv2DLayer(object):
    """tf.layers.Conv2D as utils/vision_transformer.py:39-50 uses it: construct, .build(shape), read .weights[0] and
    .padding.  The layer's variable scope is `variable_scope(None, default_name='conv2d')` at build time."""
    def __init__(self, filters, kernel_size, strides=1, padding='valid', data_format='channels_last', use_bias=True,
                 kernel_initializer=None, name=None, **_):
        self.filters, self.use_bias, self.kernel_initializer, self.name = int(filters), use_bias, kernel_initializer, name
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.padding = padding
        self.weights = []

    def build(self, input_shape):
        cin = int(input_shape[-1])
        with variable_scope(self.name, default_name='conv2d'):
            self.weights = [get_variable('kernel', [self.kernel_size[0], self.kernel_size[1], cin, self.filters],
                                         initializer=self.kernel_initializer)]
            if self.use_bias:
                self.weights.append(get_variable('bias', [self.filters], initializer=zeros_initializer()))


def sufficient_statistics(x, axes, shift=None, keep_dims=False, name=None, keepdims=None):
    x = _t(x)
    kd = bool(keep_dims if keepdims is None else keepdims)
    ax = _axes(axes)
    count = 1
    for a in ax:
        count *= int(x.shape[a])
    m_ss = torch.sum(x, dim=ax, keepdim=kd)
    v_ss = torch.sum(torch.square(x), dim=ax, keepdim=kd)
    return _w(torch.tensor(float(count))), _w(m_ss), _w(v_ss), None


def normalize_moments(counts, mean_ss, variance_ss, shift, name=None):
    divisor = 1.0 / _t(counts)
    mean = _t(mean_ss) * divisor
    variance = _t(variance_ss) * divisor - torch.square(mean)
    return _w(mean), _w(variance)


def _activation(x, activation):
    return x if activation is None else activation(x)


def dense(inputs, units, activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None,
          name=None, reuse=None, **_):
    x = _t(inputs)
    with variable_scope(name, default_name='dense'):
        kernel = get_variable('kernel', [int(x.shape[-1]), int(units)], dtype=x.dtype,
                              initializer=kernel_initializer)
        out = torch.matmul(x, kernel)
        if use_bias:
            out = out + get_variable('bias', [int(units)], dtype=x.dtype,
                                     initializer=bias_initializer or zeros_initializer())
    return _activation(_w(out), activation)


def conv2d_layer(inputs, filters, kernel_size, strides=(1, 1), padding='valid', data_format='channels_last',
                 activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None, name=None, **_):
    assert data_format == 'channels_last'
    x = _t(inputs)
    ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
    st = (strides, strides) if isinstance(strides, int) else tuple(strides)
    with variable_scope(name, default_name='conv2d'):
        kernel = get_variable('kernel', [ks[0], ks[1], int(x.shape[-1]), int(filters)], dtype=x.dtype,
                              initializer=kernel_initializer)            # HWIO
        bias = get_variable('bias', [int(filters)], dtype=x.dtype,
                            initializer=bias_initializer or zeros_initializer()) if use_bias else None
    pad = 0
    if padding.upper() == 'SAME':
        assert st == (1, 1) and ks[0] % 2 == 1 and ks[1] % 2 == 1, "SAME only for odd kernels at stride 1"
        pad = (ks[0] // 2, ks[1] // 2)
    out = F.conv2d(x.permute(0, 3, 1, 2), kernel.permute(3, 2, 0, 1), bias, stride=st, padding=pad)
    return _activation(_w(out.permute(0, 2, 3, 1)), activation)


# random -------------------------------------------------------------------------------------------------------
def random_uniform(shape=(), minval=0, maxval=None, dtype=torch.float32, seed=None, name=None):
    shape = _shape(shape)
    if dtype in (torch.int32, torch.int64):
        out = torch.randint(int(minval), int(maxval), shape, generator=STATE.gen).to(dtype)
    else:
        maxval = 1.0 if maxval is None else maxval
        u = torch.rand(shape, generator=STATE.gen, dtype=torch.float64)
        u = u.clamp_(1e-12, 1 - 1e-12).to(dtype)      # TF draws from [min, max); keep log(-log(u)) finite
        out = u * (maxval - minval) + minval
    STATE.draws.append(('uniform', out.clone()))
    return _w(out)


def random_categorical(logits, num_samples, dtype=None, seed=None, name=None):
    lg = _t(logits).double()
    p = torch.softmax(lg, dim=-1)
    out = torch.multinomial(p, int(num_samples), replacement=True, generator=STATE.gen).to(dtype or torch.int64)
    STATE.draws.append(('categorical', out.clone()))
    return _w(out)


# input pipeline (utils/model_utils.py:742-940 run on these) ------------------------------------------------------
class _Dead(object):
    """the untaken output of control_flow_ops.switch: ops that receive it produce it"""
    def __repr__(self):
        return '<dead tensor>'


DEAD = _Dead()


def cf_switch(data, pred, dtype=None, name=None):
    taken = bool(_t(pred))
    return (DEAD if taken else data, data if taken else DEAD)


def cf_merge(inputs, name=None):
    for i, x in enumerate(inputs):
        if x is not DEAD:
            return x, i
    raise RuntimeError('merge of dead tensors only')


def tf_cond(pred, true_fn=None, false_fn=None, name=None):
    return true_fn() if bool(_t(pred)) else false_fn()


def switch_case(branch_index, branch_fns, default=None, name=None):
    return branch_fns[int(_t(branch_index))]()


def image_resize_images(images, size, method=0, align_corners=False, preserve_aspect_ratio=False, name=None):
    """tf.image.resize_images on one [h, w, c] fp32 image: the TF-1.15 kernels restated in oracle/input_oracle.py."""
    from . import input_oracle
    if images is DEAD:
        return DEAD
    assert align_corners and not preserve_aspect_ratio, "only the align_corners=True form the reference uses"
    x = _t(images)
    assert x.dim() == 3 and x.dtype == torch.float32
    out = input_oracle.resize_images(x.detach().numpy(), [int(_t(v)) for v in size], int(method))
    return _w(torch.from_numpy(np.ascontiguousarray(out)))


def image_pad_to_bounding_box(image, offset_height, offset_width, target_height, target_width):
    from . import input_oracle
    out = input_oracle.pad_to_bounding_box(_t(image).detach().numpy(), int(offset_height), int(offset_width),
                                           int(target_height), int(target_width))
    return _w(torch.from_numpy(out))


def io_decode_raw(input_bytes, out_type=torch.uint8, **_):
    assert out_type == torch.uint8
    return _w(torch.from_numpy(np.frombuffer(bytes(input_bytes), np.uint8).copy()))


# gradients / misc graph-mode API ------------------------------------------------------------------------------
def gradients(ys, xs, **_):
    ys = ys if isinstance(ys, (list, tuple)) else [ys]
    total = ys[0]
    for y in ys[1:]:
        total = total + y
    with STATE.lock:
        gs = torch.autograd.grad(total, list(xs), allow_unused=True, retain_graph=True)
    gs = [None if g is None else _w(g) for g in gs]
    if STATE.gradients_log is None:
        STATE.gradients_log = {}
    STATE.gradients_log[STATE.replica] = {v.name[:-2]: g for v, g in zip(xs, gs)}
    return gs


def clip_by_global_norm(t_list, clip_norm, use_norm=None, name=None):
    if use_norm is None:
        use_norm = torch.sqrt(sum(torch.sum(torch.square(t)) for t in t_list if t is not None))
    scale = clip_norm * torch.minimum(1.0 / use_norm, torch.tensor(1.0 / clip_norm))
    return [None if t is None else _w(t * scale) for t in t_list], _w(torch.as_tensor(use_norm))


def group(*inputs, **_):
    return None


def polynomial_decay(learning_rate, global_step, decay_steps, end_learning_rate=0.0001, power=1.0, cycle=False,
                     name=None):
    assert not cycle
    lr = _t(learning_rate).float()
    gs = torch.minimum(_t(global_step).float(), torch.tensor(float(decay_steps)))
    p = gs / float(decay_steps)
    return _w((lr - end_learning_rate) * torch.pow(1.0 - p, power) + end_learning_rate)


def get_or_create_global_step():
    return _raw_get_variable('global_step', shape=[], dtype=torch.int64,
                             initializer=lambda s, d: torch.zeros(s, dtype=d), trainable=False)


class Optimizer(object):
    def __init__(self, use_locking=False, name=None):
        self._name = name


class CrossShardOptimizer(object):
    """tf.contrib.tpu.CrossShardOptimizer: cross_replica_sum of every gradient, then the wrapped optimizer."""
    def __init__(self, opt, **_):
        self._opt = opt

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        summed = [(None if g is None else cross_replica_sum(g), v) for g, v in grads_and_vars]
        return self._opt.apply_gradients(summed, global_step=global_step, name=name)


# simulated replicas (threads) -----------------------------------------------------------------------------------
def cross_replica_sum(x, group_assignment=None, name=None):
    """Every simulated replica (one thread each, see run_replicas) deposits its tensor; all get the sum.  The
    autograd graph spans the replicas, so d(sum of replica losses)/dθ carries the cross-replica terms exactly like
    the psum autodiff of the reference."""
    if STATE.num_shards <= 1:
        return _t(x)
    key = getattr(STATE.local, 'psum_count', 0)
    STATE.local.psum_count = key + 1
    with STATE.lock:
        STATE.deposit.setdefault(key, {})[STATE.replica] = _t(x)
    STATE.barrier.wait()
    parts = STATE.deposit[key]
    out = parts[0]
    for r in range(1, STATE.num_shards):
        out = out + parts[r]
    return _w(out)


def replica_id():
    return _w(torch.tensor(STATE.replica, dtype=torch.int32))


class _TpuContext(object):
    @property
    def number_of_shards(self):
        return STATE.num_shards if STATE.num_shards > 1 else None


def run_replicas(fn, num_shards):
    """run fn(replica) on `num_shards` threads sharing variables; returns the list of results."""
    results, errors = [None] * num_shards, []

    def work(r):
        STATE.local.replica = r
        STATE.local.psum_count = 0
        try:
            results[r] = fn(r)
        except BaseException as e:      # noqa
            errors.append(e)
            if STATE.barrier is not None:
                STATE.barrier.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(num_shards)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results


# ----------------------------------------------------------------------------------------------------------------
# module assembly
# ----------------------------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def _noop(*a, **k):
    return None


@contextlib.contextmanager
def _nullctx(*a, **k):
    yield None


def build_modules():
    tf = _mod('tensorflow')
    # dtypes
    for n, d in dict(float32=torch.float32, float64=torch.float64, bfloat16=torch.bfloat16, float16=torch.float16,
                     int32=torch.int32, int64=torch.int64, uint8=torch.uint8, bool=torch.bool,
                     uint32=torch.int64).items():
        setattr(tf, n, d)
    tf.Tensor = torch.Tensor
    tf.AUTO_REUSE = 'AUTO_REUSE'
    tf.executing_eagerly = lambda: True
    tf.logging = _mod('tensorflow.logging', info=_noop, warning=_noop, warn=_noop, error=_noop,
                      set_verbosity=_noop, INFO=20)
    # variables / scopes
    tf.variable_scope = variable_scope
    tf.name_scope = _nullctx
    tf.control_dependencies = _nullctx
    tf.get_variable = get_variable
    tf.get_variable_scope = get_variable_scope
    tf.trainable_variables = trainable_variables
    tf.truncated_normal_initializer = truncated_normal_initializer
    tf.constant_initializer = constant_initializer
    tf.zeros_initializer = zeros_initializer
    tf.variance_scaling_initializer = variance_scaling_initializer
    tf.glorot_uniform_initializer = glorot_uniform_initializer
    tf.GraphKeys = types.SimpleNamespace(GLOBAL_VARIABLES='variables', UPDATE_OPS='update_ops')
    tf.get_collection = lambda key, scope=None: (list(STATE.vars.values()) if key == 'variables' else [])
    # asserts are no-ops (graph-mode control dependencies)
    for n in ('assert_less_equal', 'assert_greater_equal', 'assert_less', 'assert_equal', 'Assert'):
        setattr(tf, n, _noop)
    # array / math
    tf.cast = cast; tf.reshape = reshape; tf.transpose = transpose; tf.tile = tile; tf.concat = concat
    tf.stack = stack; tf.unstack = unstack; tf.slice = tf_slice; tf.range = tf_range; tf.ones = ones
    tf.zeros = zeros; tf.zeros_like = zeros_like; tf.ones_like = ones_like; tf.fill = fill; tf.constant = constant
    tf.identity = identity; tf.shape = tf_shape; tf.matmul = matmul; tf.multiply = multiply; tf.add = add
    tf.subtract = subtract; tf.equal = equal; tf.not_equal = not_equal; tf.less = less; tf.less_equal = less_equal
    tf.greater = greater; tf.greater_equal = greater_equal; tf.logical_and = logical_and
    tf.logical_or = logical_or; tf.logical_not = logical_not; tf.minimum = minimum; tf.maximum = maximum
    tf.floor_div = floor_div; tf.mod = mod; tf.pow = tf_pow; tf.log = log; tf.exp = exp; tf.sqrt = sqrt
    tf.rsqrt = rsqrt; tf.square = square; tf.abs = tf_abs; tf.erf = erf; tf.tanh = tanh
    tf.reduce_sum = reduce_sum; tf.reduce_mean = reduce_mean; tf.reduce_max = reduce_max
    tf.reduce_min = reduce_min; tf.reduce_any = reduce_any; tf.reduce_all = reduce_all; tf.add_n = add_n
    tf.argmax = argmax; tf.sort = tf_sort; tf.argsort = argsort; tf.one_hot = one_hot; tf.gather = gather
    tf.scatter_nd = scatter_nd; tf.where_v2 = where_v2; tf.where = where_v2; tf.clip_by_value = clip_by_value
    tf.clip_by_global_norm = clip_by_global_norm; tf.gradients = gradients; tf.group = group
    tf.random_uniform = random_uniform
    tf.math = _mod('tensorflow.math', log=log, log1p=log1p, lgamma=lgamma, rsqrt=rsqrt, sqrt=sqrt, top_k=top_k,
                   l2_normalize=l2_normalize, logical_not=logical_not, sign=sign, erf=erf, abs=tf_abs,
                   square=square, pow=tf_pow, exp=exp)
    tf.random = _mod('tensorflow.random', uniform=random_uniform, categorical=random_categorical,
                     stateless_uniform=lambda shape, seed, **kw: random_uniform(shape, **kw))
    tf.nn = _mod('tensorflow.nn', softmax=softmax, log_softmax=log_softmax, moments=moments, dropout=tf_dropout,
                 avg_pool2d=avg_pool2d, relu=relu, top_k=top_k, bias_add=bias_add, l2_loss=l2_loss, conv2d=nn_conv2d,
                 sufficient_statistics=sufficient_statistics, normalize_moments=normalize_moments,
                 embedding_lookup=embedding_lookup, l2_normalize=l2_normalize)
    tf.layers = _mod('tensorflow.layers', dense=dense, conv2d=conv2d_layer, Conv2D=Conv2DLayer)
    tf.pad = tf_pad
    tf.image = _mod('tensorflow.image', ResizeMethod=types.SimpleNamespace(BILINEAR=0, NEAREST_NEIGHBOR=1,
                                                                           BICUBIC=2, AREA=3),
                    resize_images=image_resize_images, pad_to_bounding_box=image_pad_to_bounding_box)
    tf.io = _mod('tensorflow.io', decode_raw=io_decode_raw)
    tf.cond = tf_cond
    tf.switch_case = switch_case
    # estimator / tpu / train scaffolding
    tf.estimator = _mod('tensorflow.estimator',
                        ModeKeys=types.SimpleNamespace(TRAIN='train', EVAL='eval', PREDICT='infer'))
    tpu_mod = _mod('tensorflow.tpu', cross_replica_sum=cross_replica_sum)
    tf.tpu = tpu_mod
    train = _mod('tensorflow.train', Optimizer=Optimizer, get_or_create_global_step=get_or_create_global_step,
                 polynomial_decay=polynomial_decay)
    tf.train = train
    v1 = _mod('tensorflow.compat.v1', train=train)
    tf.compat = _mod('tensorflow.compat', v1=v1)
    contrib_tpu = _mod('tensorflow.contrib.tpu', CrossShardOptimizer=CrossShardOptimizer,
                       TPUEstimatorSpec=lambda **kw: kw)
    tf.contrib = _mod('tensorflow.contrib', tpu=contrib_tpu, summary=_mod('tensorflow.contrib.summary'))

    tpu_function = _mod('tensorflow.contrib.tpu.python.tpu.tpu_function', get_tpu_context=lambda: _TpuContext())
    xla = _mod('tensorflow.compiler.tf2xla.python.xla', replica_id=replica_id)
    mods = {
        'tensorflow': tf,
        'tensorflow.python': _mod('tensorflow.python'),
        'tensorflow.python.ops': _mod('tensorflow.python.ops'),
        'tensorflow.python.ops.control_flow_ops': _mod('tensorflow.python.ops.control_flow_ops'),
        'tensorflow.contrib': tf.contrib,
        'tensorflow.contrib.tpu': contrib_tpu,
        'tensorflow.contrib.tpu.python': _mod('tensorflow.contrib.tpu.python'),
        'tensorflow.contrib.tpu.python.ops': _mod('tensorflow.contrib.tpu.python.ops'),
        'tensorflow.contrib.tpu.python.ops.tpu_ops': _mod('tensorflow.contrib.tpu.python.ops.tpu_ops'),
        'tensorflow.contrib.tpu.python.tpu': _mod('tensorflow.contrib.tpu.python.tpu'),
        'tensorflow.contrib.tpu.python.tpu.tpu_function': tpu_function,
        'tensorflow.compiler': _mod('tensorflow.compiler'),
        'tensorflow.compiler.tf2xla': _mod('tensorflow.compiler.tf2xla'),
        'tensorflow.compiler.tf2xla.python': _mod('tensorflow.compiler.tf2xla.python'),
        'tensorflow.compiler.tf2xla.python.xla': xla,
    }
    mods['tensorflow.python.ops.control_flow_ops'].switch = cf_switch
    mods['tensorflow.python.ops.control_flow_ops'].merge = cf_merge
    mods['tensorflow.python.ops'].control_flow_ops = mods['tensorflow.python.ops.control_flow_ops']
    mods['tensorflow.contrib.tpu.python.ops'].tpu_ops = mods['tensorflow.contrib.tpu.python.ops.tpu_ops']
    mods['tensorflow.contrib.tpu.python.tpu'].tpu_function = tpu_function
    mods['tensorflow.compiler.tf2xla.python'].xla = xla
    return mods


def install():
    """register the shim as `tensorflow` (refuses to shadow a real TensorFlow)."""
    if 'tensorflow' in sys.modules and not getattr(sys.modules['tensorflow'], '_merlot_shim', False):
        raise RuntimeError("a real tensorflow is already imported; the shim must not shadow it")
    mods = build_modules()
    mods['tensorflow']._merlot_shim = True
    sys.modules.update(mods)
    return mods['tensorflow']
