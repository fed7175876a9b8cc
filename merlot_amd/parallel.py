"""Data parallelism for the MERLOT path: one process per GPU, torch.distributed over RCCL/xGMI ("nccl" backend).

Replaces the reference's three cross-replica call sites:
  * tpu_cross_replica_stack x2 (utils/model_utils.py:673-707, model/modeling.py:504-510): a scatter-into-zeros +
    cross_replica_sum, i.e. an all-gather whose autodiff is an all-reduce-sum + slice.  Here: ONE fused RCCL
    all-gather of [n_local, 2, C] forward and a reduce-scatter(sum) backward (`DistContext.all_gather_cat`).
  * CrossShardOptimizer (utils/optimization.py:241-245): per-variable cross_replica_sum of the gradients.  Here the
    flat fp32 gradient arena is all-reduced in per-layer buckets (28 MB each) launched from inside the backward as
    soon as a layer's weight gradients are complete, overlapping with the remaining backward; the tail (embeddings,
    heads, stems) goes in `finish()`.  Reduction is a SUM (reference-faithful); `mean` is
    available through the optimizer's grad_scale at no extra pass.
"""
import os

import torch
import torch.distributed as dist

# MERLOT_FORCE_DIST=1 issues every collective even at world_size 1 (exercises the RCCL code path on a 1-GPU box)
FORCE = os.environ.get('MERLOT_FORCE_DIST', '0') == '1'


class _AllGatherCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, gout):
        group = ctx.group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        gout = gout.contiguous()
        n = gout.shape[0] // world
        if dist.get_backend(group) == 'nccl':
            gin = torch.empty((n,) + tuple(gout.shape[1:]), device=gout.device, dtype=gout.dtype)
            dist.reduce_scatter_tensor(gin, gout, op=dist.ReduceOp.SUM, group=group)
        else:                                    # gloo has no reduce_scatter: all-reduce then slice (same result)
            dist.all_reduce(gout, op=dist.ReduceOp.SUM, group=group)
            gin = gout[rank * n:(rank + 1) * n].clone()
        return gin, None


class DistContext(object):
    """DP context handed to MerlotModel(dist=...)."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)

    def all_gather_cat(self, x):
        if self.world_size == 1 and not FORCE:
            return x
        return _AllGatherCat.apply(x, self.group)


class GradReducer(object):
    """Bucketed, overlapped all-reduce of the ParamStore gradient arena."""

    def __init__(self, store, ctx, expected_passes=None, defer=False, payload='fp32'):
        """defer=True: nothing is launched from inside the backward; `finish()` reduces the whole arena (needed when the
        local gradients are clipped by their global norm before the cross-replica sum, utils/optimization.py:233-245).
        payload='bf16': each bucket travels as bf16 (half the xGMI bytes and half the time RCCL's kernels hold CUs beside
        the backward GEMMs); the sum is formed on bf16-rounded addends and written back to the fp32 arena.  The reference
        reduces fp32 (utils/optimization.py:241-245), so 'fp32' is the default and 'bf16' an opt-in
        (`optimizer.grad_reduce_dtype: bfloat16`)."""
        if payload not in ('fp32', 'bf16'):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        self.store = store
        self.ctx = ctx
        self.defer = defer
        self.payload = payload
        self._staged = []        # [(start, end, bf16 buffer)] waiting to be copied back after their all-reduce
        self.expected = dict(expected_passes or {})
        self._seen = {}
        self._done = []          # [(start, end)]
        self._work = []
        store.grad_ready_hook = self._on_ready
        self._ranges = {}

    def set_expected(self, expected_passes):
        self.expected = dict(expected_passes)

    def _group_range(self, group):
        r = self._ranges.get(group)
        if r is None:
            r = self.store.group_range(group + '/')
            self._ranges[group] = r
        return r

    def _on_ready(self, group):
        if self.defer or (self.ctx.world_size == 1 and not FORCE):
            return
        c = self._seen.get(group, 0) + 1
        self._seen[group] = c
        if c < self.expected.get(group.split('/layer')[0] if '/layer' in group else group, self.expected.get('*', 1)):
            return
        s, e = self._group_range(group)
        self._launch(s, e)

    def _launch(self, s, e):
        if self.payload == 'bf16':
            buf = self.store.grad[s:e].to(torch.bfloat16)
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.ctx.group, async_op=True)
            self._staged.append((s, e, buf))
        else:
            w = dist.all_reduce(self.store.grad[s:e], op=dist.ReduceOp.SUM, group=self.ctx.group, async_op=True)
        self._work.append(w)
        self._done.append((s, e))

    def finish(self):
        """all-reduce whatever the backward hooks have not covered, then wait for everything."""
        if self.ctx.world_size > 1 or FORCE:
            done = sorted(self._done)
            pos = 0
            for s, e in done + [(self.store.numel, self.store.numel)]:
                if s > pos:
                    self._launch(pos, s)
                pos = max(pos, e)
            for w in self._work:
                w.wait()
            for s, e, buf in self._staged:
                self.store.grad[s:e].copy_(buf)
        self._work, self._done, self._seen, self._staged = [], [], {}, []
