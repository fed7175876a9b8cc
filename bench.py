#!/usr/bin/env python
"""bench.py -- MERLOT pretraining step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one full training step of BASELINE config #2/#3 on one batch of synthetic frame-caption segments
already resident in HBM: patch-embed ViT-B/16 (12 L) over every frame, 12-L text-only pass, attention-guided
masking, 12-L joint encoder, MLM + contrastive + temporal heads, backward, DP gradient all-reduce (RCCL, overlapped),
fused AdamW.  bf16 MFMA compute / fp32 master weights, dropout 0.1 on (merlot.yaml), nothing skipped.
Prints ONE JSON line on rank 0; value = whole-job segments/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_GFLOP_PER_SEGMENT = 170.4          # 56.79 fwd x 3, num_chunks=16, 224^2, n=4
# --resnet-stem: lite_resnet50 [3,4,9] + conv_postresnet_proj = 10.041 GFLOP/frame forward instead of the 0.231 of the
# patch conv (2*MACs over the convolutions of utils/vision_transformer.py:114-170, 213-223 at 224^2)
TRAIN_GFLOP_PER_SEGMENT_RESNET = 170.4 + 3.0 * (10.041 - 0.231)
PEAK_BF16_TFLOPS = 2500.0                # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0                 # MI355X dense fp8 MFMA (MX-scaled K=64/128 forms; same guide)


def fwd_gflop_per_segment(image, n_group, text_len=32, hidden=768, inter=3072, layers=12, patch=16, ncls=2, pool=2):
    """forward GFLOP of one frame-caption segment (2 x MACs of the GEMMs and the attention contractions), the closed form
    behind SURVEY.md 8(d)'s 56.79 (224^2, groups of 4): ViT on 2 + (image/16)^2 tokens, the joint encoder on a group of
    n x (1 + pooled grid + text_len) tokens, the text-only encoder on text_len tokens, + 1.09 for the heads."""
    lin = 2.0 * (4 * hidden * hidden + 2 * hidden * inter)             # per token and layer
    att = lambda s: 4.0 * s * s * hidden                               # per sequence and layer: QK^T + PV
    ih, iw = (image, image) if isinstance(image, int) else image      # merlot.yaml:36 ships a non-square frame, [192, 352]
    grid = (ih // patch) * (iw // patch)
    sv = ncls + grid
    sj = n_group * (1 + grid // (pool * pool) + text_len)
    vit = layers * (lin * sv + att(sv)) + 2.0 * grid * hidden * patch * patch * 3
    joint = layers * (lin * sj + att(sj)) / n_group
    text = layers * (lin * text_len + att(text_len))
    return (vit + joint + text) / 1e9 + 1.09


def usable_cores():
    """host cores this process may really use: affinity mask, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline_worker(threads):
    """(child process) the reference's CPU path, timed as the oracle restatement (fp32, unfused, torch-CPU) on a bounded
    sample of the bench workload: ONE example of 4 segments at 224^2, full 12+12+12-layer model, forward+backward."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import synth_batch
    from merlot_amd import NeatConfig
    from oracle import merlot_oracle as mo
    torch.set_num_threads(threads)
    config = NeatConfig.from_yaml(os.path.join(ROOT, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
    cfg = dict(config.model)
    cfg['hidden_dropout_prob'] = 0.0
    w = mo.init_weights(cfg, 0, perturb=False)
    for t in w.values():
        t.requires_grad_(True)
    b = synth_batch(cfg, E=1, num_chunks=4, seed=3, two_videos=False)
    times = []
    t_start = time.time()
    for it in range(4):
        for t in w.values():
            t.grad = None
        t0 = time.time()
        m = mo.MerlotOracle(cfg, w, b['image'], b['input_ids'], mask_input=True, shuffled_idx_img=b['shuffled_idx_img'],
                            noise=b['noise'])
        loss, _ = m.total_loss(b['shuffled_idx_img'], b['video_src_ids'])
        loss.backward()
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        if time.time() - t_start > 20.0 and times:
            break
    med = float(np.median(times))
    # BASELINE config #1 exactly as stated (64^2 frames, 2+2+2 layers, batch 2 x 4 segments): the reference's own
    # CPU-runnable case, timed the same way
    from common import tiny_config
    cfg1 = tiny_config(use_bfloat16=False)
    w1 = mo.init_weights(cfg1, 0, perturb=False)
    for t in w1.values():
        t.requires_grad_(True)
    b1 = synth_batch(cfg1, E=2, num_chunks=4, seed=3)
    t1 = []
    for it in range(4):
        for t in w1.values():
            t.grad = None
        t0 = time.time()
        m1 = mo.MerlotOracle(cfg1, w1, b1['image'], b1['input_ids'], mask_input=True, shuffled_idx_img=b1['shuffled_idx_img'],
                             noise=b1['noise'])
        l1, _ = m1.total_loss(b1['shuffled_idx_img'], b1['video_src_ids'])
        l1.backward()
        if it > 0:
            t1.append(time.time() - t0)
    med1 = float(np.median(t1))
    print(json.dumps({'value': 4.0 / med, 'unit': 'segments/s', 'cores': threads, 'kind': 'port',
                      'config1': {'value': 8.0 / med1, 'unit': 'segments/s', 'sample': f'BASELINE config #1 (64^2, 2+2+2 layers, '
                                  f'batch 2 x 4 segments), oracle fwd+bwd, median of {len(t1)} steps ({med1:.2f} s/step)'},
                      'sample': f'oracle (torch-CPU fp32, unfused, op-for-op restatement of the TF graph) fwd+bwd, 1 example x 4 '
                                f'segments @224^2, 12+12+12 layers, median of {len(times)} steps after 1 warm-up '
                                f'({med:.2f} s/step), {threads} threads'}), flush=True)


def cpu_baseline():
    """run the bounded CPU sample in a child with a hard timeout so the default bench run always ends in minutes."""
    import subprocess
    why = 'oracle sample did not finish within the time bound on this host'
    for threads in (min(usable_cores(), 32), 8):
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', str(threads)],
                                 capture_output=True, text=True, timeout=150)
            for line in reversed(out.stdout.strip().splitlines()):
                if line.startswith('{'):
                    return json.loads(line)
            if out.returncode != 0:                  # a crash must not masquerade as a timeout
                why = 'oracle worker failed: ' + (out.stderr.strip().splitlines() or ['?'])[-1][:200]
        except subprocess.TimeoutExpired:
            continue
    return {'value': None, 'unit': 'segments/s', 'cores': usable_cores(), 'kind': 'port', 'sample': why}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N ...` started without torchrun (no RANK / WORLD_SIZE): run the same command line as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
    and hand its stdout / stderr / exit status through unchanged."""
    import subprocess
    port = os.environ.get('MASTER_PORT') or str(free_port())
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '1')
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


GEMM_SOURCES = ('gemm.hip', 'gemm_p8.inc', 'gemm_ring.h', 'common.h')
ATTENTION_SOURCES = ('attention.hip', 'attention_res.inc', 'attention_fb.inc', 'attention_pp.inc', 'common.h')
TRAFFIC_FILE = 'r06_traffic.txt'


def source_hash(files):
    """sha256 (first 16 hex digits) of a kernel family's sources: ties the rows of a PMC traffic file to the code they measured."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, 'merlot_amd', 'csrc', f), 'rb').read())
    return h.hexdigest()[:16]


def gemm_source_hash():
    return source_hash(GEMM_SOURCES)


def measured_traffic():
    """HBM-side bytes per launch of the representative dominant launch (M=101376 N=3072 K=768, bias epilogue), from the
    TCC counters collected in separate rocprofv3 --pmc passes over THIS kernel (profiles/<TRAFFIC_FILE>, written by
    scripts/gpu_traffic.sh): 2 * FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, in bytes.  The
    file records the hash of the GEMM sources it was measured on; if the sources changed since, the figure is STALE and
    is not reported (traffic: null, with the reason)."""
    path = os.path.join(ROOT, 'profiles', TRAFFIC_FILE)
    if not os.path.exists(path):
        return {'bytes_per_launch': None, 'why': f'profiles/{TRAFFIC_FILE} absent'}
    fetch = write = src = asrc = None
    attn = {}
    for line in open(path):
        if line.startswith('gemm_source_hash'):
            src = line.split()[-1]
        if line.startswith('attention_source_hash'):
            asrc = line.split()[-1]
        if line.startswith('HBM_MB') and 'attn_' in line:
            attn[line.split(' | ')[1].strip()] = float(line.split()[-1]) * 1024.0 * 1024.0
        if 'gemm_nt_p8_kernel<0, false, false, true' in line and line.split(' | ')[0] in ('FETCH_SIZE', 'WRITE_SIZE'):   # (+ ', false>' since round 6: the LayerNorm-fold flag)
            kb = float(line.split()[-1])                      # '<counter> | <kernel> | launches n | KB_per_launch v'
            if line.startswith('FETCH_SIZE'):
                fetch = kb
            else:
                write = kb
    if fetch is None or write is None:
        return {'bytes_per_launch': None, 'why': f'profiles/{TRAFFIC_FILE} holds no gemm_nt_p8_kernel<0, false, false, true, ..> rows'}
    if src != gemm_source_hash():
        return {'bytes_per_launch': None, 'why': f'profiles/{TRAFFIC_FILE} was measured on GEMM sources {src}, the tree has '
                                                 f'{gemm_source_hash()}: stale, re-run scripts/gpu_traffic.sh'}
    # the attention rows of the same file carry their own hash: reported only while the attention sources are the measured ones
    if asrc == source_hash(ATTENTION_SOURCES):
        attention = {'hbm_bytes_per_launch': attn, 'attention_source_hash': asrc}
    else:
        attention = {'hbm_bytes_per_launch': None, 'why': f'attention rows measured on sources {asrc}, the tree has {source_hash(ATTENTION_SOURCES)}: stale'}
    return {'bytes_per_launch': (2.0 * fetch + write) * 1024.0, 'algorithmic_bytes_per_launch': 3119.0e6, 'attention': attention,
            'launch': 'forward Linear M=405504 (= 128 examples x 16 frames x 198 tokens) N=3072 K=768, bias epilogue, bf16 out (gemm_nt_p8_kernel<0,false,false,true>)',
            'source': f'profiles/{TRAFFIC_FILE}', 'gemm_source_hash': src}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--examples', type=int, default=None, help='examples (x16 segments) per GPU per step (default 128 = 2048 segments: the throughput plateau of the batch sweep, '
                         '208 GB of the 288; round 1 and most of round 2 were quoted at 32; config 5: 48 = 768 segments of 384^2, ~210 GB)')
    ap.add_argument('--config', type=int, default=2, choices=(2, 5),
                    help='BASELINE.json configs[]: 2 = the headline 4-segment 224^2 bf16 workload (configs[1]; DP over --gpus); '
                         '5 = NOT the headline: the 16-segment 384^2 long-video variant with fp8 forward GEMMs (configs[4])')
    ap.add_argument('--bf16', action='store_true', help='with --config 5: keep every GEMM in bf16 (the comparison line)')
    ap.add_argument('--fp8-fc2', action='store_true', help='with --config 5: also run fc2 on e4m3 operands (its input needs a separate two-pass quantisation)')
    ap.add_argument('--fp8', action='store_true',
                    help='NOT the headline (whose dtype is bf16): the headline GEOMETRY (config 2) with the fp8 forward GEMMs of config 5 (QKV / fc1 on e4m3 operands); '
                         'combine with --fp8-bwd.  Reported with its own metric and dtype strings')
    ap.add_argument('--fp8-bwd', type=str, default=None,
                    help='8-bit float operands in the BACKWARD (model.fp8_backward; never the headline): a comma-separated list of w1, w2, wqkv, wproj '
                         '(that weight gradient through merlot_gemm_f8_tn), e4m3 (gradient operands in e4m3 instead of e5m2), fuse (the copies come out of the '
                         'launches that produce the tensors), noa (fc1 stores only the e4m3 copy of its output; fc2 reads it), dgrad1 (fc1\'s input gradient on '
                         'the copy of du).  dgradqkv (with wqkv: the QKV input gradient on the copy of dqkv).  The configuration that pays (config 5\'s default): w1,w2,wqkv,fuse,noa,dgrad1,dgradqkv')
    ap.add_argument('--resnet-stem', action='store_true',
                    help='NOT the headline config: swap the patch stem for the ResNet-hybrid stem of merlot.yaml:30 (resnet_layers [3, 4, 9])')
    ap.add_argument('--native-yaml', action='store_true',
                    help='NOT the headline config: the model as model/configs/merlot.yaml ships it -- image_size [192, 352] (12 x 22 patches, joint S = 396) '
                         'and resnet_layers [3, 4, 9]; everything else as the headline (12 + 12 + 12 layers, 16 segments per example in groups of 4)')
    ap.add_argument('--explicit-conv', action='store_true', help='with --resnet-stem: the 3x3 convolutions on explicit im2col matrices (the path of rounds 1-3) instead of the implicit GEMM')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-tn-colsum', action='store_true', help='A/B only: the Q-third bias gradient as a stand-alone column-sum pass over dQKV (rounds 1-5) instead of merlot_gemm_bf16_tn_cs')
    ap.add_argument('--ln-fold', action='store_true', help='A/B only: the LayerNorm behind every residual GEMM from that GEMM\'s launch (merlot_gemm_bf16_nt_ln, round 6: measured level to 0.4 %% slower, off by default)')
    ap.add_argument('--no-ln-fold', action='store_true', help=argparse.SUPPRESS)   # (the default since the A/B; kept so that round 6's scripts still run)
    ap.add_argument('--gn-fused', action='store_true', help='A/B only (hybrid stem): GroupNorm in one launch in BOTH directions (default: one-launch forward, two-launch backward -- the one-launch backward is slower on every shape)')
    ap.add_argument('--no-gn-fused', action='store_true', help='A/B only (hybrid stem): GroupNorm as two launches per direction (rounds 3-5)')
    ap.add_argument('--exp-lib', action='store_true', help=argparse.SUPPRESS)       # A/B only: run on libmerlot_hip_exp.so (reads the MERLOT_* experiment switches)
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--cpu-baseline-worker', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-emulate', action='store_true', help=argparse.SUPPRESS)   # tests only: gloo + torch-CPU emulated ops, tiny model
    ap.add_argument('--one-device-gloo', action='store_true', help=argparse.SUPPRESS)   # tests only: every rank on cuda:0, gloo group (RCCL refuses two ranks on one device)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.cpu_baseline_worker)
        return
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec THIS command line under torch.distributed.run, one rank
        # per GPU on this node (the contract's own launch line); rank 0's JSON line and the exit status pass through.
        return self_launch(args.gpus)

    if args.exp_lib:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'scripts'))
        import _exp_lib  # noqa: F401
    import torch
    import torch.distributed as dist
    from merlot_amd import NeatConfig, ops
    from merlot_amd.parallel import DistContext
    from merlot_amd.train import Trainer, synthetic_batch
    if args.gn_fused or args.no_gn_fused:
        ops.GN_FUSED = bool(args.gn_fused)
    if args.ln_fold or args.no_tn_colsum:
        from merlot_amd import layers
        layers.FUSE_LN = bool(args.ln_fold)
        layers.TN_COLSUM = not args.no_tn_colsum

    if args.examples is None:
        args.examples = 48 if args.config == 5 else 56 if args.native_yaml else ((64 if args.explicit_conv else 80) if args.resnet_stem else 128)   # the hybrid stem keeps several times the activations per frame (implicit 3x3 convolutions, 80 / 96 / 112 examples: 3 030 / 3 048 / 3 086 segments/s at 193 / 231 / 269 GB)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.cpu_emulate:
        # tests/test_dist_cpu.py: THIS file's code path (launch contract, barrier + max-over-ranks timing, the JSON line) on
        # 2 gloo ranks with the HIP ops swapped for their torch-CPU emulation and the 2-layer 64x64 model of config #1
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import emu_ops

        class _MP(object):
            def setattr(self, o, n, v):
                setattr(o, n, v)
        emu_ops.install(_MP())
        torch.cuda.synchronize = lambda *a, **k: None
        ops.KernelTimer = None
        device = torch.device('cpu')
    else:
        # (--one-device-gloo, tests/test_dist_gpu.py: this file's N > 1 code path on the PRODUCT kernels where only one GPU exists -- a plumbing run)
        dev_idx = 0 if args.one_device_gloo else local_rank
        torch.cuda.set_device(dev_idx)
        device = torch.device('cuda', dev_idx)
    ctx = None
    force_dist = os.environ.get('MERLOT_FORCE_DIST', '0') == '1' and 'RANK' in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL's kernels each hold a CU while a collective runs and a 160 KiB-LDS GEMM workgroup cannot share that CU: cap
        # the channels so the overlapped gradient all-reduce takes at most 16 of the 256 CUs (the GEMMs claim their tiles
        # dynamically and absorb that, profiles/r02_c_dp_contention.txt).  Override by exporting NCCL_MAX_NCHANNELS.
        os.environ.setdefault('NCCL_MAX_NCHANNELS', '16')
        if args.cpu_emulate or args.one_device_gloo:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)
        ctx = DistContext()

    config = NeatConfig.from_yaml(os.path.join(ROOT, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
    if args.cpu_emulate:
        config.model.update(image_size=[64, 64], num_hidden_layers=2, num_vision_transformer_hidden_layers=2,
                            num_lang_transformer_hidden_layers=2, hidden_dropout_prob=0.0)    # the emulation has no dropout
        config.data['num_chunks'] = 4
    train_gflop = TRAIN_GFLOP_PER_SEGMENT
    fp8 = False
    if args.config == 5:
        # BASELINE configs[4]: 16 segments per group at 384^2 (Sv = 578, joint S = 2832); everything else as merlot.yaml
        fp8 = not args.bf16
        config.model.update(image_size=[384, 384], num_chunks_in_group=16, fp8_forward=(True if args.fp8_fc2 else 'ln') if fp8 else False)
        train_gflop = 3.0 * fwd_gflop_per_segment(384, 16)
    if args.fp8 and args.config == 2:
        fp8 = True
        config.model['fp8_forward'] = 'ln'
    if args.config == 5 and fp8 and args.fp8_bwd is None:
        # round 6: config #5's line runs the configuration that pays (profiles/r06_s_bench5_*.json: 549 ms all-bf16, 532 ms fp8 forward only, 479 ms with the
        # MLP half's 8-bit backward, 469 ms with the QKV weight / input gradients on dqkv's copy as well, on one box); --fp8-bwd none gives the forward-only line of rounds 2-5
        args.fp8_bwd = 'w1,w2,wqkv,fuse,noa,dgrad1,dgradqkv'
    if args.fp8_bwd in ('none', 'off', ''):
        args.fp8_bwd = None
    if args.fp8_bwd:
        config.model['fp8_backward'] = args.fp8_bwd
    if args.native_yaml:
        # model/configs/merlot.yaml:30,36.  SURVEY 8(d): 72.92 GFLOP forward per segment on the patch stem at 192 x 352 (the closed form below
        # gives 72.9); the stem's 10.041 GFLOP per 224^2 frame scale with the pixel count (every convolution is 'SAME' / stride-aligned)
        args.resnet_stem = True
        config.model['image_size'] = [192, 352]
        stem = 10.041 * (192 * 352) / (224 * 224)
        train_gflop_native = 3.0 * (fwd_gflop_per_segment((192, 352), 4) - 2.0 * 264 * 768 * 768 / 1e9 + stem)
    if args.resnet_stem:
        config.model['resnet_layers'] = [3, 4, 9]
        config.model['resnet_implicit_conv'] = not args.explicit_conv
        train_gflop = train_gflop_native if args.native_yaml else TRAIN_GFLOP_PER_SEGMENT_RESNET
    trainer = Trainer(config, device, ctx, seed=0)
    batch = synthetic_batch(config, args.examples, device, seed=1234 + rank)
    seg_per_gpu = args.examples * config.data['num_chunks']

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = trainer.step(batch)
    # per-launch HIP events (on the launch stream) for the roofline figure: recorded during the LAST step of the timed
    # region only -- ~700 GEMM launches are sample enough, and bracketing every launch of every step with two events
    # costs ~3 % of the step (151 -> 156 ms), which would be charged to `value`.
    timer = None if (args.no_kernel_timing or args.cpu_emulate) else ops.KernelTimer()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timer is not None and i == args.steps - 1:
            ops.TIMER = timer
        out = trainer.step(batch)
    sync()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    timed_steps = 1
    loss = float(out['loss'].detach())
    # forward-only figure, measured AFTER the timed region with the same bracket; not part of `value`
    fwd_steps = max(2, min(args.steps, 4))
    trainer.forward_only(batch)
    sync()
    t1 = time.perf_counter()
    for _ in range(fwd_steps):
        trainer.forward_only(batch)
    sync()
    fwd_elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([elapsed, fwd_elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, fwd_elapsed = float(t[0].item()), float(t[1].item())

    hbm = None
    if device.type == 'cuda':
        free_b, total_b = torch.cuda.mem_get_info(device)
        hbm = {'peak_allocated_gb': torch.cuda.max_memory_allocated(device) / 1e9, 'peak_reserved_gb': torch.cuda.max_memory_reserved(device) / 1e9,
               'device_total_gb': total_b / 1e9, 'device_free_after_gb': free_b / 1e9}
    if rank == 0:
        value = world * seg_per_gpu * args.steps / elapsed
        res = {
            'metric': 'frame-caption segments/sec/node (4-seg, 192x352 as merlot.yaml ships it, bf16)' if args.native_yaml else
                      'frame-caption segments/sec/node (4-seg, 224^2, bf16)' if (args.config == 2 and not fp8) else
                      ('frame-caption segments/sec/node (4-seg, 224^2, NOT the headline dtype: fp8 QKV/fc1 forward GEMMs' + (' + 8-bit backward (' + args.fp8_bwd + ')' if args.fp8_bwd else '') + ')') if args.config == 2 else
                      'frame-caption segments/sec/node (16-seg, 384^2, %s)' % ((('fp8 QKV/fc1/fc2 forward GEMMs' if args.fp8_fc2 else 'fp8 QKV/fc1 forward GEMMs') + (' + 8-bit backward (' + args.fp8_bwd + ')' if args.fp8_bwd else '')) if fp8 else 'bf16'),
            'value': value, 'unit': 'segments/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': (('fp8 (e4m3 operands, fp32 accumulate: QKV / fc1%s forward GEMMs; ' % (' / fc2' if args.fp8_fc2 else '')) +
                      (('8-bit backward [' + args.fp8_bwd + ']: weight gradients on e5m2 x e4m3 operands' + (', fc2 forward on the e4m3 copy fc1 wrote' if 'noa' in args.fp8_bwd else '') +
                        (", fc1's input gradient on the e5m2 copy the GELU' epilogue wrote" if 'dgrad1' in args.fp8_bwd else '') + (", the QKV input gradient on dqkv's e5m2 copy" if 'dgradqkv' in args.fp8_bwd else '') + '; bf16 elsewhere)') if args.fp8_bwd else 'bf16 elsewhere)')) if fp8 else 'bf16',
            'data': 'synthetic',
            'config': {'workload': (('merlot.yaml 4-segment ResNet-hybrid [3,4,9] + ViT-B/16' if args.resnet_stem else
                                     'merlot.yaml 4-segment full ViT-B/16 (patch stem)') + ' + 12-layer joint + 12-layer text-only, '
                                    + ('192 x 352 frames (model/configs/merlot.yaml:30,36)' if args.native_yaml else '224^2 frames') +
                                    ', 32-token captions, fwd+bwd+DP all-reduce+AdamW, dropout 0.1') if args.config == 2 else
                                   ('BASELINE configs[4]: 16-segment long-video variant, full ViT-B/16 at 384^2 (578 tokens/frame) + '
                                    '12-layer joint over 2832-token groups + 12-layer text-only, fwd+bwd+AdamW, dropout 0.1; '
                                    + (('fp8 forward GEMMs' + (' + 8-bit backward' if args.fp8_bwd else '')) if fp8 else 'all-bf16 comparison run')),
                       'baseline_config': args.config,
                       'segments_per_gpu_per_step': seg_per_gpu, 'examples_per_gpu': args.examples,
                       'num_chunks': config.data['num_chunks'], 'parallelism': f'dp{world}' + (' (all ranks on ONE device over gloo: a plumbing run, not a scaling number)' if args.one_device_gloo else ''), 'grad_reduce': 'sum',
                       'stem': 'resnet-hybrid [3,4,9] (merlot.yaml:30)' if args.resnet_stem else 'patch 16x16 (north_star)',
                       'image_size': list(config.model['image_size']), 'train_gflop_per_segment': train_gflop,
                       'final_loss': loss, **({'fp8_backward': args.fp8_bwd} if args.fp8_bwd else {})},
            'model_flops_utilization': value * train_gflop / 1e3 / (world * PEAK_BF16_TFLOPS),
            'forward_only': {'value': world * seg_per_gpu * fwd_steps / fwd_elapsed, 'unit': 'segments/s',
                             'ms_per_pass': 1e3 * fwd_elapsed / fwd_steps, 'passes': fwd_steps,
                             'model_flops_utilization': world * seg_per_gpu * fwd_steps / fwd_elapsed *
                             (train_gflop / 3.0) / 1e3 / (world * PEAK_BF16_TFLOPS)},
        }
        if hbm is not None:
            res['hbm'] = hbm                                 # rank 0's allocator peaks (torch caching allocator; RCCL's buffers are in `device_free_after`)
        if timer is not None:
            summ = timer.summary()
            f, t, n = summ.get('gemm_nt', (0.0, 1.0, 0))
            traffic = measured_traffic()                     # PMC bytes of the representative launch, or None
            res['roofline'] = {'bound': 'mfma',
                               'kernel': 'merlot_gemm_bf16_nt = gemm_nt_p8_kernel<EPI,OUT,FP8,PH2> (persistent ping-pong 256x256, two phases per K-tile) + gemm_nt_ring_kernel<Cfg<2,4,2,2,32,3>,EPI,OUT> '
                                         '(bf16 MFMA 32x32x16, all epilogues; the dominant kernel family of the step)',
                               'achieved': f / t / 1e12, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': f / t / 1e12 / PEAK_BF16_TFLOPS, 'traffic': (traffic or {}).get('bytes_per_launch'),
                               'traffic_detail': traffic, 'launches': n,
                               'avg_launch_us': 1e6 * t / max(n, 1), 'gflop_per_launch': f / max(n, 1) / 1e9,
                               'share_of_step_time': t / timed_steps / (elapsed / args.steps),
                               'timed_steps': timed_steps}
            if 'gemm_fp8_nt' in summ:
                f8, t8, n8 = summ['gemm_fp8_nt']
                res['roofline_fp8'] = {'bound': 'mfma', 'kernel': 'merlot_gemm_fp8_nt = gemm_nt_p8_kernel<EPI,OUT,FP8> (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales)',
                                       'achieved': f8 / t8 / 1e12, 'peak': PEAK_FP8_TFLOPS, 'unit': 'TFLOP/s',
                                       'frac': f8 / t8 / 1e12 / PEAK_FP8_TFLOPS, 'launches': n8,
                                       'share_of_step_time': t8 / timed_steps / (elapsed / args.steps)}
            if 'gemm_f8_tn' in summ:
                f3, t3, n3 = summ['gemm_f8_tn']
                res['roofline_f8_wgrad'] = {'bound': 'mfma', 'kernel': 'merlot_gemm_f8_tn = gemm_tn_q8_kernel (ds_read_b64_tr_b8 fragments, v_mfma_scale_f32_32x32x64_f8f6f4) + tn_reduce_kernel',
                                            'achieved': f3 / t3 / 1e12, 'peak': PEAK_FP8_TFLOPS, 'unit': 'TFLOP/s', 'frac': f3 / t3 / 1e12 / PEAK_FP8_TFLOPS,
                                            'launches': n3, 'share_of_step_time': t3 / timed_steps / (elapsed / args.steps)}
                if 'gemm_fp8_nt' in summ:
                    res['roofline_f8_all'] = {'bound': 'mfma', 'kernel': 'every 8-bit GEMM launch of the step (merlot_gemm_fp8_nt / _q8, merlot_gemm_f8_nt, merlot_gemm_f8_tn)',
                                              'achieved': (f8 + f3) / (t8 + t3) / 1e12, 'peak': PEAK_FP8_TFLOPS, 'unit': 'TFLOP/s',
                                              'frac': (f8 + f3) / (t8 + t3) / 1e12 / PEAK_FP8_TFLOPS, 'launches': n8 + n3,
                                              'share_of_step_time': (t8 + t3) / timed_steps / (elapsed / args.steps)}
            if 'gemm_tn' in summ:
                f2, t2, n2 = summ['gemm_tn']
                res['roofline_wgrad'] = {'kernel': 'merlot_gemm_bf16_tn = gemm_tn_p1_kernel (one phase per K-tile; + gemm_tn_ring_kernel for small shapes) + tn_reduce_kernel', 'achieved': f2 / t2 / 1e12, 'unit': 'TFLOP/s',
                                         'frac': f2 / t2 / 1e12 / PEAK_BF16_TFLOPS, 'launches': n2,
                                         'share_of_step_time': t2 / timed_steps / (elapsed / args.steps)}
        if world == 1 and not args.no_cpu_baseline and not args.resnet_stem and not args.cpu_emulate and args.config == 2:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
