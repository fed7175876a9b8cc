#!/bin/bash
# round 6, call 48: blocks per launch of the LayerNorm backward (scripts/exp_ln_bwd_blocks.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -u scripts/exp_ln_bwd_blocks.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_z11_ln_bwd_blocks.txt
