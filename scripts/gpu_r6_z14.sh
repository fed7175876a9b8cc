#!/bin/bash
# round 6, call 54: the one-launch GroupNorm forward with claimed items (product) against item = workgroup id (MERLOT_GN_STATIC=1, experiments build): per-shape medians of 300 calls
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ echo "== claimed items (product library)"; timeout 300 python -u scripts/exp_gn_stress.py 2>&1 | grep -v "Warning\|amdgpu.ids"
  echo "== item = workgroup id (MERLOT_GN_STATIC=1, experiments library)"; MERLOT_GN_STATIC=1 timeout 300 python -u scripts/exp_gn_stress.py 2>&1 | grep -v "Warning\|amdgpu.ids"
  echo "== claimed items again"; timeout 300 python -u scripts/exp_gn_stress.py 2>&1 | grep -v "Warning\|amdgpu.ids"; } | tee gpurun_out/r06_z14_gn_static_fwd.txt
