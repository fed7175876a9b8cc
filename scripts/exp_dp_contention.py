"""DP de-risk on ONE GPU: what does the overlapped gradient all-reduce cost the step?

There is no multi-GPU box for this build, so the communication kernel is emulated by its footprint: RCCL holds one CU
per channel while a collective runs, and a 160 KiB-LDS GEMM workgroup cannot share that CU.  `merlot_probe_cu_hog`
launches H one-wave workgroups that each pin a whole CU's LDS and spin for a given time.  Rows:

  base          MERLOT_FORCE_DIST off, no collectives (the N=1 bench configuration)
  rccl-w1       every collective issued on a world-size-1 RCCL group (the real code path, the real launches)
  bucket H p    + for every gradient bucket, H hogged CUs for the time an 8-GPU ring all-reduce of that bucket takes at
                BUSBW (payload p = fp32 | bf16), started on a side stream where the bucket's all-reduce is launched --
                the all-reduce-shaped contention NCCL_MAX_NCHANNELS=H produces
  always H      H CUs hogged for the WHOLE step (worst case: a collective that never ends); the ideal cost is H/(256-H)

Run on the GPU box:  MERLOT_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 \
                     python scripts/exp_dp_contention.py > gpurun_out/r02_c_dp_contention.txt
"""
import os
import sys
import time

os.environ.setdefault('MERLOT_FORCE_DIST', '1')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('LOCAL_RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29517')
os.environ.setdefault('NCCL_MAX_NCHANNELS', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from merlot_amd import NeatConfig, parallel  # noqa: E402
from merlot_amd.parallel import DistContext, GradReducer  # noqa: E402
from merlot_amd.train import Trainer, synthetic_batch  # noqa: E402
from probe_lib import cu_hog  # noqa: E402

BUSBW = float(os.environ.get('BUSBW_GBS', '200')) * 1e9       # conservative 8-GPU ring bus bandwidth on 16 channels
STEPS = int(os.environ.get('STEPS', '6'))
EXAMPLES = int(os.environ.get('EXAMPLES', '32'))


def main():
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    dist.init_process_group('nccl', device_id=device)
    ctx = DistContext()
    config = NeatConfig.from_yaml(os.path.join(ROOT, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
    trainer = Trainer(config, device, ctx, seed=0)
    batch = synthetic_batch(config, EXAMPLES, device, seed=1234)
    side = torch.cuda.Stream()
    sink = torch.zeros(4, device=device, dtype=torch.int32)

    # calibrate the hog's clock: cycles per second of its s_memtime loop
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    cu_hog(1, 160 * 1024, 20_000_000, sink)
    e1.record()
    torch.cuda.synchronize()
    hz = 20_000_000 / (e0.elapsed_time(e1) * 1e-3)
    print(f"# hog clock {hz / 1e6:.1f} MHz; BUSBW {BUSBW / 1e9:.0f} GB/s; NCCL_MAX_NCHANNELS={os.environ['NCCL_MAX_NCHANNELS']}; "
          f"{EXAMPLES} examples x 16 segments per step, {STEPS} timed steps per row")

    orig_launch = GradReducer._launch
    mode = {'hog': 0, 'bytes_per_el': 4, 'total_s': 0.0}

    def launch(self, s, e):
        if mode['hog']:
            secs = (e - s) * mode['bytes_per_el'] * 2.0 * 7.0 / 8.0 / BUSBW
            mode['total_s'] += secs
            side.wait_stream(torch.cuda.current_stream())          # the bucket's gradients exist from here on
            with torch.cuda.stream(side):
                cu_hog(mode['hog'], 160 * 1024, int(secs * hz), sink)
        return orig_launch(self, s, e)
    GradReducer._launch = launch

    def run(label, force, hog=0, bytes_per_el=4, always=0, base=None):
        parallel.FORCE = force
        mode.update(hog=hog, bytes_per_el=bytes_per_el, total_s=0.0)
        for _ in range(2):
            trainer.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            if always:
                with torch.cuda.stream(side):
                    cu_hog(always, 160 * 1024, int(always_s * hz), sink)
            trainer.step(batch)
            if always:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / STEPS * 1e3
        extra = f"  hogged {mode['total_s'] / (STEPS + 2) * 1e3:5.1f} ms/step" if hog else ''
        rel = f"  {100.0 * (ms / base - 1.0):+5.1f} %" if base else ''
        print(f"{label:<22s} {ms:8.2f} ms/step{rel}{extra}", flush=True)
        return ms

    base = run('base', False)
    always_s = base * 1e-3 * 0.98
    run('rccl-w1', True, base=base)
    for h in (8, 16, 32):
        run(f'bucket {h:2d} fp32', True, hog=h, bytes_per_el=4, base=base)
    for h in (8, 16, 32):
        run(f'bucket {h:2d} bf16', True, hog=h, bytes_per_el=2, base=base)
    for h in (8, 16, 32):
        run(f'always {h:2d}', True, always=h, base=base)
    run('base (again)', False, base=base)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
