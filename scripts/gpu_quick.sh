#!/bin/bash
# usage: gpu_quick.sh "<pytest -k expr>" [seg]
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$1" 2>&1 | tail -40
timeout 300 python scripts/bench_kernels.py ${2:-256} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kernels.log
