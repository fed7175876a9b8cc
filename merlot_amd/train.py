"""Step loop of the MI355X MERLOT path (the role of model/train.py + TPUEstimator.train in the reference):
forward (ViT, text-only, joint, three heads) -> backward (grads accumulate into the flat arena, per-layer all-reduce
buckets launched from inside the backward) -> fused AdamW."""
import numpy as np
import torch

from .modeling import model_fn_builder, draw_mask_noise
from .optimization import build_optimizer_from_config
from .parallel import GradReducer
from .params import ParamStore
from . import checkpoint as ckpt_io


def synthetic_batch(config, examples, device, seed=1234, num_chunks=None):
    """Synthetic inputs of SURVEY.md 8(d): frames U[0,1) bf16, captions START + 8..31 tokens + padding, one video per
    example, shuffled_idx_img from the model/dataloader.py:224-257 generator, explicit masking noise."""
    from . import ops
    m = config.model
    nc = num_chunks or config.data['num_chunks']
    Lc = config.data.get('chunk_text_len', 32)
    n = m['num_chunks_in_group']
    H, W = m['image_size']
    g = torch.Generator().manual_seed(seed)
    N = examples * nc
    images = torch.rand((N, H, W, 3), generator=g, dtype=torch.float32).to(torch.bfloat16).to(device)
    lens = torch.randint(8, Lc, (examples, nc), generator=g)
    ids = torch.randint(100, 50354, (examples, nc, Lc), generator=g)
    ids[:, :, 0] = 2
    ids = torch.where(torch.arange(Lc)[None, None] < lens[..., None], ids, torch.zeros_like(ids)).int().to(device)
    B = N // n
    p = m.get('image_shuffle_prob', 0.5)
    probs = torch.tensor([1.0 - p, 1e-6] + [p / (n - 1)] * (n - 1), dtype=torch.float64)
    num_shuffle = torch.multinomial(probs, B, replacement=True, generator=g).int().to(device)
    u_sel = torch.rand((B, n), generator=g).to(device)
    u_perm = torch.rand((B, n), generator=g).to(device)
    sidx = ops.shuffled_idx(num_shuffle, u_sel, u_perm, B, n, 16)
    noise = draw_mask_noise(B, Lc * n, m, m['vocab_size'], g)
    noise = {k: v.to(device) for k, v in noise.items()}
    return {'images': images, 'input_ids': ids, 'shuffled_idx_img': sidx,
            'video_src_ids': torch.zeros((examples, nc), dtype=torch.int32, device=device), 'noise': noise}


class Trainer(object):
    def __init__(self, config, device, dist_ctx=None, seed=0, grad_reduce='sum'):
        self.config = config
        self.device = torch.device(device)
        self.dist = dist_ctx
        self.store = ParamStore(config.model, self.device, seed=seed)
        world = dist_ctx.world_size if dist_ctx is not None else 1
        self.opt = build_optimizer_from_config(self.store, config.optimizer, world_size=world, grad_reduce=grad_reduce)
        # a training step always runs the backward: let the attention LOG metrics (model/modeling.py:186-203, :709) come out of the
        # attention backward instead of a second Q K^T walk in every joint-encoder forward launch (they are read at the END of the
        # step, as in the reference's train op); `attention_log_in_backward: False` in the YAML restores forward-time values
        config.model.setdefault('attention_log_in_backward', True)
        self.model_fn = model_fn_builder(config)
        self.reducer = None
        from .parallel import FORCE
        if dist_ctx is not None and (world > 1 or FORCE):
            # encoder weights receive gradients from the joint AND the text-only pass before they may be reduced
            shared = 2 if config.model.get('share_params', True) else 1
            payload = 'bf16' if str(config.optimizer.get('grad_reduce_dtype', 'float32')) in ('bfloat16', 'bf16') else 'fp32'
            self.reducer = GradReducer(self.store, dist_ctx,
                                       expected_passes={'encoder': shared, 'encoder/LayerNorm_ln_final': shared, '*': 1},
                                       defer=self.opt.clip_norm > 0.0, payload=payload)
        self.step_idx = 0
        self._pending_check = None       # (pinned host bool, event, step): last step's token-id range flag, fetched one step late
        # model/modeling.py:724-738: variables the init checkpoint also holds (weights and, since this is the training
        # graph, the Adam slots) start from it; global_step does not.
        self.initialized_variable_names = {}
        init_checkpoint = config.model.get('init_checkpoint', None)
        if init_checkpoint:
            self.initialized_variable_names = ckpt_io.init_from_checkpoint(self.store, init_checkpoint, self.opt)

    def save(self, model_dir=None):
        """`model.ckpt-<global_step>` in the TF bundle format the reference's Estimator writes to device.output_dir."""
        return ckpt_io.save_checkpoint(model_dir or self.config.device['output_dir'], self.store, self.opt)

    def restore(self, model_dir_or_prefix=None):
        """Estimator auto-resume from the latest checkpoint of output_dir (weights, Adam slots, global_step)."""
        self.step_idx = ckpt_io.restore_checkpoint(model_dir_or_prefix or self.config.device['output_dir'], self.store,
                                                   self.opt)
        return self.step_idx

    def step_seed(self):
        """seed of this step's dropout masks and MLM masking noise on THIS replica.  TPU replicas draw independent
        randomness (every replica runs its own `tf.random` ops), so the rank is folded in: distinct for every
        (step, rank) pair.  Parameter initialisation (`ParamStore(seed=...)`) stays rank-independent."""
        world = self.dist.world_size if self.dist is not None else 1
        rank = self.dist.rank if self.dist is not None else 0
        return (self.step_idx + 1) * world + rank

    def forward_only(self, features):
        """one forward pass (ViT, text-only, masking, joint, the three losses) with the training graph's ops but no
        tape: the `fwd-only` figure SURVEY.md 8(d) asks for next to the training step."""
        with torch.no_grad():
            return self.model_fn(features, None, 'train', {'store': self.store, 'dist': self.dist, 'seed': self.step_seed()})

    def check_inputs(self, wait=True):
        """raise if the step embedded a token id outside [0, vocab) (utils/model_utils.py:256-258's in-graph assertion).
        The flag is copied to pinned host memory asynchronously behind the forward pass; `Trainer.step` reads it after the
        backward has been enqueued and before the optimizer update."""
        if self._pending_check is None:
            return
        host, ev, step = self._pending_check
        if not wait and not ev.query():
            return
        ev.synchronize()
        self._pending_check = None
        if bool(host):
            raise ValueError(f"token id out of range (training step {step})")

    def _defer_check(self, flag):
        from .parallel import FORCE
        multi = self.dist is not None and (self.dist.world_size > 1 or FORCE)
        if flag is None:
            if not multi:
                return
            # a rank whose graph produced no flag still takes part in the collective below (a zero): returning early here while the
            # other ranks reduce would deadlock them
            flag = torch.zeros((), dtype=torch.bool, device=self.device)
        if multi:
            # the reference's in-graph assertion fails the whole job: every rank has to see ANY rank's bad batch, or the
            # good ranks would apply an update that already contains the bad rank's gradients and then hang at the next
            # collective.  One 4-byte MAX all-reduce, queued right behind the forward like the copy below.
            import torch.distributed as dist
            flag = flag.to(torch.int32).reshape(1)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.dist.group)
            flag = flag.reshape(()) != 0
        if flag.is_cuda:
            host = torch.empty((), dtype=torch.bool, pin_memory=True)
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            class _Done(object):
                def query(self):
                    return True

                def synchronize(self):
                    pass
            host, ev = flag, _Done()
        self._pending_check = (host, ev, self.step_idx)

    def step(self, features):
        self.store.zero_grad()
        out = self.model_fn(features, None, 'train', {'store': self.store, 'dist': self.dist, 'seed': self.step_seed()})
        self._defer_check(out.get('token_id_out_of_range'))  # async copy of the flag, recorded right behind the forward
        out['loss'].backward()
        if self.opt.clip_norm > 0.0:                         # clip the local gradients, then sum across replicas
            out['grad_norm'] = self.opt.clip_local_gradients()
        if self.reducer is not None:
            self.reducer.finish()
        # THIS step's flag, before its update is applied (the reference fails the step in-graph, utils/model_utils.py:256-258):
        # the host waits for the forward's embedding lookups only -- the whole backward is already queued behind them, so
        # the GPU never idles -- and a bad batch neither reaches the weights nor a checkpoint (on ANY rank: the flag is
        # MAX-reduced across the replicas in _defer_check, so all of them raise together).
        self.check_inputs(wait=True)
        self.opt.step()
        self.step_idx += 1
        # the caller gets values, not a graph: the step's autograd nodes (and what their ctx objects still hold: LayerNorm inputs, the
        # heads' activations -- several GB at the bench's batch) are released here instead of when the caller drops `out` a step later
        def values(o):
            if isinstance(o, torch.Tensor):
                return o.detach()
            if isinstance(o, dict):
                return {k: values(v) for k, v in o.items()}
            if isinstance(o, tuple) and hasattr(o, '_fields'):           # a namedtuple takes its fields positionally, not one iterable
                return type(o)(*(values(v) for v in o))
            if isinstance(o, (list, tuple)):
                return type(o)(values(v) for v in o)
            return o
        return values(out)


def train(config, device, dist_ctx=None, max_steps=None, num_workers=None, log_every=100, seed=0):
    """`estimator.train(input_fn_builder(config, is_training=True), max_steps=optimizer.num_train_steps)` of
    model/train.py:17-26 with the Estimator's housekeeping: resume from the latest checkpoint of `device.output_dir` if
    there is one, a checkpoint every `device.iterations_per_loop` steps (utils/neat_config.py:140) and at the end, written
    by rank 0 in the reference's own bundle format.  `train_batch_size` is the GLOBAL batch (examples), split evenly over
    the ranks like the TPU replicas.  -> the Trainer."""
    import os

    from .input_pipeline import InputPipeline
    rank = dist_ctx.rank if dist_ctx is not None else 0
    world = dist_ctx.world_size if dist_ctx is not None else 1
    global_bs = int(config.device['train_batch_size'])
    if global_bs % world != 0:
        raise ValueError(f"train_batch_size {global_bs} is not divisible by {world} replicas")
    trainer = Trainer(config, device, dist_ctx, seed=seed)
    out_dir = config.device['output_dir']
    if ckpt_io.latest_checkpoint(out_dir) is not None:
        trainer.restore(out_dir)
    max_steps = int(max_steps if max_steps is not None else config.optimizer['num_train_steps'])
    every = int(config.device.get('iterations_per_loop', 1000))
    if num_workers is None:
        num_workers = int(config.data.get('num_workers', 0))
    pipe = InputPipeline(config, True, global_bs // world, device, rank=rank, world_size=world,
                         seed=seed + trainer.step_idx, num_workers=num_workers)
    try:
        for features in pipe:
            if trainer.step_idx >= max_steps:
                break
            out = trainer.step(features)
            if rank == 0 and log_every and trainer.step_idx % log_every == 0:
                print(f"step {trainer.step_idx}: loss {float(out['loss'].detach()):.4f}", flush=True)
            if rank == 0 and (trainer.step_idx % every == 0 or trainer.step_idx == max_steps):
                os.makedirs(out_dir, exist_ok=True)
                trainer.save(out_dir)
    finally:
        pipe.close()
    return trainer


def main(argv=None):
    """`python -m merlot_amd.train configs/merlot.yaml` (one process per GPU under torch.distributed.run)."""
    import argparse
    import os

    import torch.distributed as dist

    from .config import NeatConfig
    from .parallel import DistContext
    ap = argparse.ArgumentParser(description='MERLOT pretraining on MI355X')
    ap.add_argument('config_file')
    ap.add_argument('--max-steps', type=int, default=None)
    ap.add_argument('--num-workers', type=int, default=None, help='loader processes per rank (default data.num_workers or 0)')
    args = ap.parse_args(argv)
    config = NeatConfig.from_yaml(args.config_file)
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    ctx = None
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=device)
        ctx = DistContext()
    train(config, device, ctx, max_steps=args.max_steps, num_workers=args.num_workers)
    if ctx is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
