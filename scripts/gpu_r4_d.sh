#!/bin/bash
# round 4, call D: attention LOG from the backward -- kernel tests, model test, step A/B (config key off / on in one call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -6
for m in off on on off; do
  echo "== attention_log_in_backward $m"
  MERLOT_LOG_BWD=$m timeout 300 python scripts/bench_logbwd.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  nt %.3f  tn %.3f  fwd %.1f ms' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass']))"
done ) 2>&1 | tee gpurun_out/r04_d_attention_log_bwd.txt | cut -c1-300
