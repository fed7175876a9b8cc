#!/bin/bash
# whole-step A/B of the NT kernel's tile enumeration (experiments build): MERLOT_P8_CG = columns per group (0: row-major)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cg in 0 3 4 6 0; do
  echo "== MERLOT_P8_CG=$cg"
  MERLOT_P8_CG=$cg timeout 300 python scripts/bench_exp.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  nt %.3f  tn %.3f  fwd %.1f ms' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass']))"
done 2>&1 | tee gpurun_out/r03_o_cg_step.txt
