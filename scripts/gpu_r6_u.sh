#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python scripts/exp_attn_bwd_q8.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_u_attn_bwd_q8.txt
