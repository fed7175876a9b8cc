#!/bin/bash
# round 6, call 59: positions per loop iteration of the two-launch GroupNorm backward (scripts/exp_gn_unroll.py) at 896 and 224 frames; stem kernel tests on the rebuilt library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 896 224; do timeout 300 python -u scripts/exp_gn_unroll.py $n 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee gpurun_out/r06_z18_gn_unroll.txt
timeout 600 python -m pytest tests/test_stem_kernels_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -2 | cut -c1-200 | tee -a gpurun_out/r06_z18_gn_unroll.txt
