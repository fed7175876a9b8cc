"""GPU, world_size 2 on ONE device: two processes, each running the HIP kernels on cuda:0, joined by a gloo group (RCCL refuses two
ranks on one device; gloo stages device tensors through the host).  The DP host path of merlot_amd.parallel -- the differentiable
in-batch all-gather and the bucketed gradient all-reduce launched from inside the backward -- with the PRODUCT ops underneath: what
tests/test_dist_cpu.py checks on the torch emulation (the single-process oracle fed with the gathered embeddings; the reference's own
two-replica program, tests/golden/ref_shim_dp2.npz), and two processes' kernels -- persistent GEMMs and attention kernels claiming their
tiles and items from per-process counters -- sharing the CUs of one GPU while they do it."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from test_dist_cpu import _fixture_noise
    from common import tiny_config, synth_batch
    from merlot_amd import MerlotModel, ParamStore
    from merlot_amd.parallel import DistContext, GradReducer
    from oracle import merlot_oracle as mo

    cfg = tiny_config()
    w = mo.init_weights(cfg, 0)
    b = synth_batch(cfg, seed=10 + rank)                       # each replica its own data
    b['noise'] = _fixture_noise(rank)                          # the draws the reference program made (dp2 fixture)
    st = ParamStore(cfg, 'cuda', seed=0)
    st.load_tf_weights(w)
    ctx = DistContext()
    red = GradReducer(st, ctx, expected_passes={'encoder': 2, 'encoder/LayerNorm_ln_final': 2, '*': 1})
    st.zero_grad()
    sidx = torch.from_numpy(b['shuffled_idx_img']).cuda()
    pm = MerlotModel(cfg, True, False, b['image'].cuda(), b['input_ids'].cuda(), mask_input=True, shuffled_idx_img=sidx, params=st,
                     noise={k: torch.from_numpy(v) for k, v in b['noise'].items()}, dist=ctx)
    l1 = pm.mask_loss()[0]
    l2, i2 = pm.contrastive_loss()
    l3 = pm.temporal_loss(sidx, torch.from_numpy(b['video_src_ids']).cuda())[0]
    (l1 + l2 + l3).backward()
    n_async = len(red._work)
    red.finish()
    torch.cuda.synchronize()
    torch.save({'grad': {k: v.float().cpu() for k, v in st.export_tf_grads().items()}, 'loss': float(l1 + l2 + l3),
                'contr': {k: float(v) for k, v in i2.items()}, 'n_async': n_async}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_match_the_oracle_and_the_reference_two_replica_program(tmp_path):
    from test_dist_cpu import _free_port, check_dp2
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f'rank{r}.pt')) for r in range(world)]
    check_dp2(res)


def _train_worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
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
    t = T.train(config, 'cuda', DistContext(), max_steps=2, log_every=0)
    torch.cuda.synchronize()
    torch.save({'master': t.store.master.float().cpu().clone(), 'step': t.step_idx}, os.path.join(tmp, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_train_loop_two_ranks_on_one_gpu_end_with_the_same_weights(tmp_path):
    """merlot_amd.train.train on two ranks (their own shards of the tfrecords, the overlapped gradient reduction, the cross-rank token-id
    flag, AdamW on the summed gradients) with the HIP kernels: after two steps both replicas hold the same master weights, bit for bit,
    they differ from the initial ones, and the checkpoint of step 2 is there."""
    from test_dist_cpu import _free_port
    from test_input_pipeline import _write_records
    from merlot_amd import checkpoint as ck
    for i in range(4):
        _write_records(str(tmp_path / f'train{i:03d}.tfrecord'), 3, 4, seed=60 + i, vocab=2000)
    mp.spawn(_train_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(str(tmp_path / 'rank0.pt')), torch.load(str(tmp_path / 'rank1.pt'))
    assert r0['step'] == r1['step'] == 2
    assert torch.equal(r0['master'], r1['master']) and bool(torch.isfinite(r0['master']).all())
    prefix = ck.latest_checkpoint(str(tmp_path / 'out'))
    assert prefix.endswith('model.ckpt-2') and int(ck.load_variable(prefix, 'global_step')) == 2


@pytest.mark.timeout(900)
def test_bench_py_two_ranks_on_one_gpu(tmp_path):
    """bench.py's OWN N > 1 code path (the torch.distributed.run launch contract, Trainer with the overlapped GradReducer and the contrastive
    all-gather, barrier + max-over-ranks timing, ONE JSON line from rank 0 with the whole-job aggregate) on the product kernels and the full 12 + 12 + 12
    layer model, two ranks of 4 examples sharing the one GPU of the box over gloo (`--one-device-gloo`, tests only).  A plumbing run: no throughput claim."""
    import json
    import subprocess
    from test_dist_cpu import _free_port
    root = os.path.dirname(HERE)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--examples', '4', '--one-device-gloo', '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=850, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                   # rank 0 only
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['steps'] == 2 and res['warmup'] == 1 and res['scaling'] == 'weak' and res['dtype'] == 'bf16'
    assert res['config']['parallelism'].startswith('dp2 (all ranks on ONE device') and res['config']['segments_per_gpu_per_step'] == 64
    assert abs(res['value'] - 2 * 64 * 2 / (res['ms_per_step'] * 2 / 1e3)) < 1e-6 * res['value']     # whole-job aggregate over both ranks
    assert res['config']['final_loss'] == res['config']['final_loss'] and 5.0 < res['config']['final_loss'] < 30.0
    assert res['roofline']['frac'] > 0.02
