#!/bin/bash
# One GPU-box pass: kernel + model parity tests, smoke, a short bench.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
nproc >> gpurun_out/gpu_info.txt; lscpu | grep "Model name" >> gpurun_out/gpu_info.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --examples ${EX:-4} --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_small.log 2>&1; tail -3 gpurun_out/bench_small.log
