#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python scripts/exp_gn_chunks.py 2>&1 | grep -v amdgpu.ids ) 2>&1 | tee gpurun_out/r04_g_gn_chunks.txt | cut -c1-400
