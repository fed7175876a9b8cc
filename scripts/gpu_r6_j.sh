#!/bin/bash
# round 6, call 11: the 8-bit weight-gradient kernel -- tr_b8 lane map, correctness, A/B against the bf16 kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python scripts/exp_f8_tn.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_j_f8_tn.txt
