#!/bin/bash
# round 6, call 46: blocks per launch of the two-launch GroupNorm backward at smaller batches (which target serves every N)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 448 224 64; do timeout 200 python -u scripts/exp_gn_blocks.py $n 256,512,768,1024,2048 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee gpurun_out/r06_z9_gn_blocks_n.txt
