#!/bin/bash
# the whole GPU suite on the current build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) 2>&1 | tee gpurun_out/r04_h_gpu_suite.txt
