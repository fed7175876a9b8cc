"""Mirror of utils/optimization.py: `build_optimizer_from_config` -> AdamW with bias correction, warm-up + linear
decay, regex `param_overrides`, and (use_bfloat16_adam) bf16 m / sign-bit-encoded bf16 v -- as ONE fused HBM-bound
kernel (merlot_adamw_step) per run of identically-configured parameters in the flat arena."""
import math
import re

import torch

from . import ops

# internal (fused) names -> the TF names the reference's regexes see
_TF_ALIASES = {'/qkv/': '/query_layer/'}


def learning_rate_scale(step, num_train_steps, num_warmup_steps):
    """utils/optimization.py:94-115 (note the `+ 1.0` in base_scale, kept)."""
    base_scale = float(num_train_steps) / (float(num_train_steps) - float(num_warmup_steps) + 1.0) if num_warmup_steps else 1.0
    s = min(step, num_train_steps)
    scale = base_scale * (1.0 - s / float(num_train_steps))           # polynomial_decay power=1, end 0
    if num_warmup_steps and step < num_warmup_steps:
        scale = float(step) / float(num_warmup_steps)
    return scale


class AdamOptimizer(object):
    """`create_fixed_adam_optimizer_with_warmup` + `AdamOptimizer` (utils/optimization.py:55-255, 290-416) over the flat
    arena.  Defaults are the reference's: clip_norm 1.0 (:57), weight_decay_rate 1e-4, epsilon 1e-6, beta_2 0.98,
    beta_1 0.9 (:173 -- not configurable at the top level there either, only per parameter through `param_overrides`)."""

    def __init__(self, store, learning_rate, num_train_steps, num_warmup_steps, weight_decay_rate=1e-4,
                 param_overrides=None, freeze_scope=None, epsilon=1e-6, beta_2=0.98, use_bfloat16_adam=False,
                 clip_norm=1.0, grad_reduce='sum', world_size=1, beta_1=0.9, do_param_scale=False,
                 decay_beta2_adafactor=False, grad_reduce_dtype='float32', **kwargs):
        self.clip_norm = float(clip_norm or 0.0)
        self.store = store
        self.lr, self.nts, self.nws = learning_rate, num_train_steps, num_warmup_steps
        self.eps, self.b1, self.b2 = epsilon, beta_1, beta_2
        self.step_count = 0
        self.grad_scale = 1.0 / world_size if grad_reduce == 'mean' else 1.0
        sd = torch.bfloat16 if use_bfloat16_adam else torch.float32
        self.m = torch.zeros(store.numel, device=store.device, dtype=sd)
        self.v = torch.zeros(store.numel, device=store.device, dtype=sd)
        param_overrides = [list(x) for x in (param_overrides or [])]
        if freeze_scope is not None:                          # deprecated spelling of a learning_rate-0 override (:124-127)
            param_overrides.append([[f'^{freeze_scope}'], {'learning_rate': 0}])
        for regexes, over in param_overrides:                 # validated whether or not a variable matches (:131-137)
            for k in over:
                if k not in ('learning_rate', 'weight_decay_rate', 'beta_1', 'beta_2', 'epsilon', 'do_factor'):
                    raise ValueError(f"Regex rule {regexes} -> {over} isn't OK because {k} isn't a changable optimization parameter")
        # per-parameter hyper-parameters (:139-143, 324-351), then merge arena neighbours with equal settings
        runs = []
        self.hparams = {}
        for name, (off, n, _) in store.offsets.items():
            tf_name = name
            for a, b in _TF_ALIASES.items():
                tf_name = tf_name.replace(a, b)
            hp = {'learning_rate': learning_rate, 'weight_decay_rate': weight_decay_rate, 'beta_1': beta_1, 'beta_2': beta_2,
                  'epsilon': epsilon}
            for regexes, over in param_overrides:
                if any(re.search(rx, tf_name) is not None for rx in regexes):
                    hp.update({k: v for k, v in over.items() if k != 'do_factor'})
            self.hparams[name] = hp
            key = (float(hp['weight_decay_rate']), float(hp['learning_rate']), float(hp['beta_1']), float(hp['beta_2']),
                   float(hp['epsilon']))
            end = off + (n + 63) // 64 * 64
            if runs and runs[-1][2] == key and runs[-1][1] == off:
                runs[-1][1] = end
            else:
                runs.append([off, end, key])
        self.runs = runs
        # variables with learning_rate 0 are dropped from `tvars` BEFORE tf.gradients (:145-152): they get no update and
        # their gradients do not enter the global norm
        self.frozen = [(s_, e_) for s_, e_, k in runs if k[1] == 0.0]
        # when only the weight decay differs between parameters (the merlot.yaml case) the whole arena updates in ONE
        # launch driven by a per-64-element flag table; otherwise one launch per run.
        rest = {k[1:] for _, _, k in runs}
        self.single_launch = len(rest) == 1 and next(iter(rest))[0] != 0.0
        self.single_key = next(iter(rest)) if len(rest) == 1 else None      # (lr, beta1, beta2, eps) of that one launch
        if self.single_launch:
            flags = torch.zeros(store.numel // 64, dtype=torch.uint8)
            wds = {k[0] for _, _, k in runs if k[0] > 0}
            if len(wds) > 1:
                self.single_launch = False
            else:
                self._wd = next(iter(wds)) if wds else 0.0
                for s_, e_, k in runs:
                    if k[0] > 0:
                        flags[s_ // 64:e_ // 64] = 1
                self._wd_flags = flags.to(store.device)

    def clip_local_gradients(self):
        """tf.clip_by_global_norm(grads, clip_norm) (utils/optimization.py:233-237): grads * clip / max(||grads||, clip),
        applied to THIS replica's gradients BEFORE the cross-replica sum (the reference clips, then CrossShardOptimizer
        reduces, :241-245) -- the trainer therefore defers the all-reduce until after this call when clip_norm > 0.
        Frozen variables (learning_rate 0) are not in the reference's `grads` list: their slots are zeroed first so they
        do not contribute.  Returns the global norm (device scalar)."""
        g = self.store.grad                                   # arena padding is zero: it does not change the norm
        for s_, e_ in self.frozen:
            g[s_:e_].zero_()
        norm = torch.linalg.vector_norm(g, dtype=torch.float64).float()   # 2e8 addends: accumulate in fp64
        if self.clip_norm > 0.0:
            g.mul_(self.clip_norm / torch.clamp(norm, min=self.clip_norm))
        return norm

    def current_lr(self):
        return self.lr * learning_rate_scale(self.step_count, self.nts, self.nws)

    def _lr_mult(self, b1, b2):
        t = self.step_count + 1.0                                    # utils/optimization.py:354-358
        bc1 = 1.0 - math.pow(b1, t)
        bc2 = 1.0 - math.pow(b2, t)
        return learning_rate_scale(self.step_count, self.nts, self.nws) * math.sqrt(bc2) / bc1

    def step(self):
        st = self.store
        if self.single_launch:
            lr, b1, b2, eps = self.single_key                        # the ONE (lr, beta1, beta2, eps) every run shares: a
            ops.adamw_step(st.master, st.grad, self.m, self.v, lr * self._lr_mult(b1, b2), b1, b2,   # uniform param_override included
                           eps, self._wd, self.grad_scale, wd_flags=self._wd_flags)
            self.step_count += 1
            st.master_version += 1
            return
        for s, e, (wd, lr_p, b1, b2, eps) in self.runs:
            if lr_p == 0.0:                                          # frozen parameters (:145-152)
                continue
            ops.adamw_step(st.master[s:e], st.grad[s:e], self.m[s:e], self.v[s:e], lr_p * self._lr_mult(b1, b2), b1, b2,
                           eps, wd, self.grad_scale)
        self.step_count += 1
        st.master_version += 1


def build_optimizer_from_config(store, optimizer_config, device_config=None, **extra):
    """utils/optimization.py:11-30."""
    kwargs = dict(optimizer_config)
    if device_config is not None:
        kwargs.update({k: v for k, v in device_config.items() if k in ('use_tpu',)})
    typ = kwargs.pop('type')
    if typ != 'adam_optimizer':
        raise ValueError("The optimizer type {} isn't supported".format(typ))
    if kwargs.pop('adafactor', False):
        raise ValueError("Adafactor not supported rn")
    kwargs.pop('use_tpu', None)
    kwargs.pop('verbose', None)
    kwargs.update(extra)
    return AdamOptimizer(store, **kwargs)
