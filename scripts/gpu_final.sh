#!/bin/bash
# final pass of the round: full GPU suite, smoke, bench (with cpu_baseline), rocprof kernel stats, RCCL path on one rank,
# input-pipeline measurements
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py --steps 8 --warmup 3 > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log | cut -c1-400
MERLOT_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_dist1.log 2>&1; tail -1 gpurun_out/bench_dist1.log | cut -c1-200
timeout 300 python scripts/exp_image_frames.py > gpurun_out/image_frames.txt 2>&1; cat gpurun_out/image_frames.txt | grep -v amdgpu.ids
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_bench.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $R/gpurun_out/kernel_stats.csv 2>/dev/null
echo prof done
