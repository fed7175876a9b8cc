#!/bin/bash
# round 2: all -m gpu tests (new persistent-kernel / depth-12 / shuffle fixture tests included), smoke, bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 2>&1 | tail -80 > gpurun_out/${TAG:-r2}_pytest_gpu.log; tail -5 gpurun_out/${TAG:-r2}_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py --steps ${STEPS:-6} --warmup 3 ${EXTRA:---no-cpu-baseline} > gpurun_out/${TAG:-r2}_bench.log 2>&1; tail -1 gpurun_out/${TAG:-r2}_bench.log | cut -c1-1500
