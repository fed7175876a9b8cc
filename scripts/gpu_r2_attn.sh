#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -15
timeout 300 python scripts/bench_kernels.py 512 2>&1 | grep -i "attention" 
