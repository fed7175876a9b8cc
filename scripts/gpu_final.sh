#!/bin/bash
# final pass of the round: full GPU suite, smoke, bench (with cpu_baseline), rocprof kernel stats of the same command
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 150 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -1 gpurun_out/pytest_gpu.log
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 120 python bench.py --steps 8 --warmup 3 > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log | cut -c1-330
cd /tmp; rm -rf /tmp/prof
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_bench.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $R/gpurun_out/kernel_stats.csv 2>/dev/null
echo prof done
