#!/bin/bash
# round 5: same-box A/B of two product builds by the whole training step (libmerlot_hip_old.so = the build before), alternating, + the GEMM tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(
for l in old new old new old new; do
  f=merlot_amd/libmerlot_hip.so; [ $l = old ] && f=merlot_amd/libmerlot_hip_old.so
  echo "== step, $l"
  AB_LIB=$f timeout 300 python scripts/bench_lib.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  nt %.3f  tn %.3f  fwd %.1f ms' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass']))"
done
timeout 600 python -m pytest tests/test_gemm_persist_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or tn or wgrad" 2>&1 | tail -2
) 2>&1 | tee gpurun_out/${TAG:-r05_x_step_ab}.txt | cut -c1-300
