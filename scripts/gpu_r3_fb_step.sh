#!/bin/bash
# whole-step A/B of the fused attention backward + resident plain forward (experiments build, same box, one call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "0 0" "1 1" "1 0" "0 0" "1 1"; do
  set -- $cfg
  echo "== MERLOT_ATTN_FB=$1 MERLOT_ATTN_RESFWD=$2"
  MERLOT_ATTN_FB=$1 MERLOT_ATTN_RESFWD=$2 timeout 300 python scripts/bench_exp.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  nt %.3f  tn %.3f  fwd %.1f ms' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass']))"
done 2>&1 | tee gpurun_out/r03_k_attention_step_ab.txt
