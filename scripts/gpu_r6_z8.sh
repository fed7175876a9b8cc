#!/bin/bash
# round 6, call 45: blocks per launch of the two-launch GroupNorm backward (scripts/exp_gn_blocks.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python -u scripts/exp_gn_blocks.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_z8_gn_blocks.txt
