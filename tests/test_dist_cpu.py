"""CPU, world_size 2 over gloo: the DP path of merlot_amd.parallel -- the differentiable in-batch all-gather
(model/modeling.py:504-510 + utils/model_utils.py:673-707) and the bucketed gradient all-reduce
(utils/optimization.py:241-245, SUM semantics) -- driven by the real MerlotModel host code with the HIP ops swapped
for their torch emulation.  Reference behaviour reproduced on ONE process by the oracle fed with the gathered
embeddings: sum over replicas of (local MLM + local temporal + contrastive-with-global-negatives)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _fixture_noise(rank):
    fx = np.load(os.path.join(HERE, 'golden', 'ref_shim_dp2.npz'))
    return {k: fx[f'r{rank}/noise/{k}'] for k in ('gumbel', 'span_lower', 'span_upper', 'random_ids', 'option')}


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import emu_ops
    import merlot_amd.ops as real

    class MP(object):
        def setattr(self, o, n, v):
            setattr(o, n, v)
    emu_ops.install(MP())
    from common import tiny_config, synth_batch
    from merlot_amd import MerlotModel, ParamStore
    from merlot_amd.parallel import DistContext, GradReducer
    from oracle import merlot_oracle as mo

    cfg = tiny_config()
    w = mo.init_weights(cfg, 0)
    b = synth_batch(cfg, seed=10 + rank)                       # each replica its own data
    b['noise'] = _fixture_noise(rank)                          # the draws the reference program made (dp2 fixture)
    st = ParamStore(cfg, 'cpu', seed=0)
    st.load_tf_weights(w)
    ctx = DistContext()
    red = GradReducer(st, ctx, expected_passes={'encoder': 2, 'encoder/LayerNorm_ln_final': 2, '*': 1})
    st.zero_grad()
    pm = MerlotModel(cfg, True, False, b['image'], b['input_ids'], mask_input=True,
                     shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']), params=st,
                     noise={k: torch.from_numpy(v) for k, v in b['noise'].items()}, dist=ctx)
    l1 = pm.mask_loss()[0]
    l2, i2 = pm.contrastive_loss()
    l3 = pm.temporal_loss(torch.from_numpy(b['shuffled_idx_img']), torch.from_numpy(b['video_src_ids']))[0]
    (l1 + l2 + l3).backward()
    n_async = len(red._work)
    red.finish()
    torch.save({'grad': st.export_tf_grads(), 'loss': float(l1 + l2 + l3), 'contr': {k: float(v) for k, v in i2.items()},
                'n_async': n_async}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp2_matches_single_process_oracle(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f'rank{r}.pt')) for r in range(world)]
    check_dp2(res)


def check_dp2(res, world=2):
    """What two replicas must have produced (shared with tests/test_dist_gpu.py, where the two ranks run the HIP kernels)."""
    # every replica ends with the same (summed) gradient arena
    for k in res[0]['grad']:
        assert torch.allclose(res[0]['grad'][k], res[1]['grad'][k], rtol=0, atol=0), k
    assert res[0]['n_async'] >= 4            # per-layer buckets were launched from inside the backward

    # single-process reference: oracle on both replicas' data with the gathered contrastive sets
    from common import tiny_config, synth_batch, rel_l2
    from oracle import merlot_oracle as mo
    cfg = tiny_config()
    w = mo.init_weights(cfg, 0)
    for t in w.values():
        t.requires_grad_(True)
    models, batches = [], []
    for r in range(world):
        b = synth_batch(cfg, seed=10 + r)
        b['noise'] = _fixture_noise(r)
        m = mo.MerlotOracle(cfg, w, b['image'], b['input_ids'], mask_input=True, shuffled_idx_img=b['shuffled_idx_img'],
                            noise=b['noise'])
        models.append(m)
        batches.append(b)
    embs = [m.contrastive_embeddings() for m in models]
    all_lang = torch.cat([e[0] for e in embs], 0)
    all_viz = torch.cat([e[1] for e in embs], 0)
    total = 0.0
    for r, (m, b) in enumerate(zip(models, batches)):
        # the reference's psum gradient keeps cross-replica terms: do NOT detach the other replicas' embeddings
        lc, ic = m.contrastive_loss(all_lang=all_lang, all_viz=all_viz, my_group_idx=r)
        lt = m.mask_loss()[0] + lc + m.temporal_loss(b['shuffled_idx_img'], b['video_src_ids'])[0]
        assert abs(float(lt) - res[r]['loss']) < 3e-2
        assert abs(float(ic['lang_to_viz']) - res[r]['contr']['lang_to_viz']) < 2e-2
        total = total + lt
    total.backward()                                         # objective = SUM over replicas
    rels = []
    for k, v in w.items():
        if v.grad is None or k.endswith('key_layer/bias'):
            continue
        rels.append(rel_l2(res[0]['grad'][k], v.grad))
        assert rels[-1] < 0.15, (k, rels[-1])
    assert np.median(rels) < 3e-2

    # and directly against the reference's own two-replica program (tpu_cross_replica_stack + CrossShardOptimizer
    # executed under oracle/tf_shim.py, tests/golden/ref_shim_dp2.npz): per-replica losses, summed gradients
    from common import head
    fx = np.load(os.path.join(HERE, 'golden', 'ref_shim_dp2.npz'))
    for r in range(world):
        assert abs(res[r]['loss'] - float(fx[f'r{r}/loss'])) < 3e-2
        assert abs(res[r]['contr']['lang_to_viz'] - float(fx[f'r{r}/contr/lang_to_viz'])) < 2e-2
    for k in fx.files:
        if k.startswith('grad/'):
            n = k[5:]
            r_ = rel_l2(torch.from_numpy(head(res[0]['grad'][n].float().numpy())), torch.from_numpy(fx[k]))
            assert r_ < 0.15, (n, r_)


def _train_worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import emu_ops

    class MP(object):
        def setattr(self, o, n, v):
            setattr(o, n, v)
    emu_ops.install(MP())
    from common import tiny_config
    from merlot_amd import train as T
    from merlot_amd.config import NeatConfig
    from merlot_amd.parallel import DistContext
    config = NeatConfig.from_dict({
        'data': {'train_file': os.path.join(tmp, 'train*.tfrecord'), 'num_chunks': 4, 'chunk_text_len': 32, 'shuffle_buffer_size': 2,
                 'num_threads': 2},
        'model': dict(tiny_config(), vocab_size=2048),
        'optimizer': {'type': 'adam_optimizer', 'learning_rate': 1e-4, 'num_train_steps': 100, 'num_warmup_steps': 10,
                      'weight_decay_rate': 0.1, 'beta_2': 0.98, 'use_bfloat16_adam': True},
        'device': {'output_dir': os.path.join(tmp, 'out'), 'train_batch_size': 4, 'iterations_per_loop': 2}})
    t = T.train(config, 'cpu', DistContext(), max_steps=2, log_every=0)
    seeds = []
    for k in range(3):                                        # the seeds the next three steps would use on this rank
        seeds.append(t.step_seed())
        t.step_idx += 1
    t.step_idx -= 3
    from merlot_amd.modeling import draw_mask_noise
    gum = draw_mask_noise(2, 128, config.model, 2048, torch.Generator().manual_seed(seeds[0] * 7919 + 17))['gumbel']
    torch.save({'master': t.store.master.clone(), 'step': t.step_idx, 'seeds': seeds, 'gumbel': gum},
               os.path.join(tmp, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_train_loop_two_replicas(tmp_path):
    """merlot_amd.train.train on 2 gloo ranks: the files are sharded by rank, the global batch of 4 examples is split 2 + 2,
    both replicas hold the same weights after two steps (summed gradients, same update), rank 0 writes the checkpoint."""
    from test_input_pipeline import _write_records
    from merlot_amd import checkpoint as ck
    for i in range(4):
        _write_records(str(tmp_path / f'train{i:03d}.tfrecord'), 3, 4, seed=60 + i, vocab=2000)
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(str(tmp_path / 'rank0.pt')), torch.load(str(tmp_path / 'rank1.pt'))
    assert r0['step'] == r1['step'] == 2
    assert torch.equal(r0['master'], r1['master'])
    # replicas must not share dropout masks / masking noise -- the per-step seed carries the rank
    assert len(set(r0['seeds'] + r1['seeds'])) == 6 and not torch.equal(r0['gumbel'], r1['gumbel'])
    prefix = ck.latest_checkpoint(str(tmp_path / 'out'))
    assert prefix.endswith('model.ckpt-2') and int(ck.load_variable(prefix, 'global_step')) == 2


@pytest.mark.timeout(900)
def test_bench_py_code_path_two_ranks_emulated(tmp_path):
    """bench.py's OWN code path at world_size 2 -- the torch.distributed.run launch contract (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_*), Trainer with the overlapped GradReducer and the contrastive all-gather, barrier +
    max-over-ranks timing, one JSON line from rank 0 with the whole-job aggregate -- on gloo with the HIP ops emulated
    (`--cpu-emulate`: 2-layer 64x64 model).  No throughput claim: a plumbing test."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    port = _free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--examples', '1', '--cpu-emulate']
    env = dict(os.environ, OMP_NUM_THREADS='4')
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                   # rank 0 only
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['steps'] == 2 and res['warmup'] == 1 and res['scaling'] == 'weak'
    assert res['config']['parallelism'] == 'dp2' and res['config']['segments_per_gpu_per_step'] == 4
    assert abs(res['value'] - 2 * 4 * 2 / (res['ms_per_step'] * 2 / 1e3)) < 1e-6 * res['value']     # whole-job aggregate
    assert res['config']['final_loss'] == res['config']['final_loss'] and res['config']['final_loss'] < 30.0


@pytest.mark.timeout(900)
def test_bench_py_self_launches_when_started_without_torchrun(tmp_path):
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment (the command shape of the
    driver's 1-GPU record) must not die before the first collective: it re-executes itself under torch.distributed.run and
    rank 0's JSON line comes through."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['OMP_NUM_THREADS'] = '4'
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--examples', '1',
           '--cpu-emulate']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['parallelism'] == 'dp2'
