#!/bin/bash
# round 4, call E: column-grouped tile order (experiments build) -- whole launch vs loop-only time, and the per-tile timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( CGS=0,6,3,4,0 timeout 400 python scripts/exp_p8_cg.py 2>&1 | grep -v amdgpu.ids
for cg in 0 6 3; do echo "== MERLOT_P8_CG=$cg"; MERLOT_P8_CG=$cg timeout 300 python scripts/exp_p8_trace.py 2>&1 | grep "cycles per tile"; done
) 2>&1 | tee gpurun_out/r04_e_cg_timeline.txt | cut -c1-400
