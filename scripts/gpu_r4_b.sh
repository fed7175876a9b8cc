#!/bin/bash
# round 4, call B: packed (v_pk_fma_f32, one dependent chain per pair) against scalar interleaved (8 chains of v_fmaak_f32) GELU epilogues
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_persist_gpu.py -x -q -m gpu 2>&1 | tail -3
for l in old new new old; do
  f=merlot_amd/libmerlot_hip.so; [ $l = old ] && f=merlot_amd/libmerlot_hip_pk.so
  echo "== $l"
  AB_LIB=$f timeout 300 python scripts/ab_lib.py 2>&1 | tail -1
done
for l in old new old new; do
  f=merlot_amd/libmerlot_hip.so; [ $l = old ] && f=merlot_amd/libmerlot_hip_pk.so
  echo "== step, $l"
  AB_LIB=$f timeout 300 python scripts/bench_lib.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  nt %.3f  tn %.3f  fwd %.1f ms' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass']))"
done ) 2>&1 | tee gpurun_out/r04_b_gelu_scalar_ab.txt | cut -c1-300
