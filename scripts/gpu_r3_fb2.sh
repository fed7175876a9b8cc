#!/bin/bash
# product build: attention tests + bench after the fused short-sequence backward
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -5
timeout 600 python bench.py > gpurun_out/r03_i_bench.json 2> gpurun_out/r03_i_bench.err; tail -c 1500 gpurun_out/r03_i_bench.json
