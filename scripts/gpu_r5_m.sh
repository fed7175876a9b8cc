#!/bin/bash
# round 5: same-box A/B of the NT kernel with pinned fragment-read addresses: the product build before (libmerlot_hip_old.so) against the current one,
# on the step's eight NT shapes at 101 376 and 405 504 rows, then the whole training step, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(
timeout 300 python scripts/exp_p8_ktile.py 2>&1 | grep "cycles per K-tile" | head -4
for T in 101376 405504; do
for l in old new new old; do
  f=merlot_amd/libmerlot_hip.so; [ $l = old ] && f=merlot_amd/libmerlot_hip_old.so
  echo "== T $T $l"
  T=$T AB_LIB=$f timeout 300 python scripts/ab_lib.py 2>&1 | tail -1
done; done
for l in old new old new; do
  f=merlot_amd/libmerlot_hip.so; [ $l = old ] && f=merlot_amd/libmerlot_hip_old.so
  echo "== step, $l"
  AB_LIB=$f timeout 300 python scripts/bench_lib.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  nt %.3f  tn %.3f  fwd %.1f ms' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass']))"
done
timeout 600 python -m pytest tests/test_gemm_persist_gpu.py -x -q -m gpu 2>&1 | tail -2
) 2>&1 | tee gpurun_out/r05_x_nt_ab.txt | cut -c1-300
