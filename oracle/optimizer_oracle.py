"""TEST INFRASTRUCTURE ONLY -- numpy/fp32 restatement of the reference's optimizer arithmetic
(utils/optimization.py:55-416).  Pinned by tests/golden/ref_shim_optimizer.npz (the reference's own AdamOptimizer
executed under oracle/tf_shim.py, see that file's header for what such a fixture does and does not pin).
Never imported by merlot_amd/.
"""
import numpy as np
import torch

MISSING_PRECISION = np.float32(1.00390625)                     # utils/optimization.py:267


def _bf16(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def decode_v(stored_v):
    """utils/optimization.py:268-281: positive sign -> |v|, negative sign -> |v| * (1 + 2^-8)."""
    a = np.abs(stored_v).astype(np.float32)
    return np.where(np.sign(stored_v) > 0, a, a * MISSING_PRECISION).astype(np.float32)


def encode_v(v):
    """utils/optimization.py:283-288: bf16 round, sign bit says which of {enc, enc*(1+2^-8)} is nearer."""
    enc = _bf16(v)
    err0 = np.abs(enc - v)
    err1 = np.abs(enc * MISSING_PRECISION - v)
    return np.where(err0 <= err1, enc, -enc).astype(np.float32)


def learning_rate_scale(global_step, num_train_steps, num_warmup_steps):
    """utils/optimization.py:94-115 in fp32 like the graph: polynomial_decay(base_scale, power 1, end 0), then the
    warm-up override.  `base_scale` keeps the reference's `+ 1.0`."""
    base_scale = float(num_train_steps) / (float(num_train_steps) - float(num_warmup_steps) + 1.0) \
        if num_warmup_steps else 1.0
    gs = np.float32(min(global_step, num_train_steps))
    p = gs / np.float32(num_train_steps)
    scale = np.float32(base_scale) * (np.float32(1.0) - p)
    if num_warmup_steps and global_step < num_warmup_steps:
        scale = np.float32(global_step) / np.float32(num_warmup_steps)
    return np.float32(scale)


def adamw_update(param, grad, m, v, global_step, learning_rate, lr_scale, weight_decay_rate, beta_1=0.9,
                 beta_2=0.98, epsilon=1e-6, use_bfloat16_adam=True):
    """AdamOptimizer.apply_gradients for one parameter (utils/optimization.py:339-416).  `m`, `v` are the STORED
    states as fp32 arrays (bf16-representable when use_bfloat16_adam).  Returns (new_param, new_m, new_v)."""
    f = np.float32
    t = f(global_step) + f(1.0)                                                     # :355
    bc1 = f(1.0) - np.power(f(beta_1), t, dtype=np.float32)
    bc2 = f(1.0) - np.power(f(beta_2), t, dtype=np.float32)
    lr = f(f(learning_rate) * f(lr_scale))                                           # :351
    lr = f(lr * (np.sqrt(bc2, dtype=np.float32) / bc1))                              # :358
    g2 = (np.square(grad, dtype=np.float32) + f(1e-30)).astype(np.float32)           # :360
    gmean = f(g2.mean(dtype=np.float32))                                             # :366-369 (no-ops in fp32)
    lr = f(lr + gmean * f(1e-30))
    eps = f(f(epsilon) + gmean * f(1e-30))
    m_ = m.astype(np.float32)
    v_ = decode_v(v) if use_bfloat16_adam else v.astype(np.float32)
    next_m = (f(beta_1) * m_ + f(1.0 - beta_1) * grad).astype(np.float32)            # :389
    next_v = (f(beta_2) * v_ + f(1.0 - beta_2) * g2).astype(np.float32)              # :390
    update = next_m / (np.sqrt(next_v, dtype=np.float32) + eps)                      # :392
    if weight_decay_rate > 0:
        update = update + f(weight_decay_rate) * param                               # :401-402
    new_param = (param - lr * update).astype(np.float32)                             # :404-406
    if use_bfloat16_adam:
        next_m, next_v = _bf16(next_m), encode_v(next_v)                             # :408-410
    return new_param, next_m, next_v
