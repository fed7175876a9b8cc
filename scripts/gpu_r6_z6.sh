#!/bin/bash
# round 6, call 42: stress of the one-launch GroupNorm forward (the product default): 300 individually timed calls per as-shipped shape; the one-launch backward of the 66-slice shapes the same way
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -u scripts/exp_gn_stress.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_z6_gn_stress.txt
for i in 4 1; do timeout 150 python -u scripts/exp_gn_stress.py bwd $i 2>&1 | grep -v "Warning\|amdgpu.ids"; [ ${PIPESTATUS[0]} = 124 ] && echo "   shape $i backward: stopped by the 150 s timeout"; done | tee -a gpurun_out/r06_z6_gn_stress.txt
