#!/bin/bash
# round 4, call S: row-wise im2col3x3 kernel (no per-element integer division; measured level with the generic kernel and not kept -- the kernel is described in profiles/r04_s_im2col_rows.txt): stem tests, micro-benchmark old/new, stem step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_stem_kernels_gpu.py tests/test_stem_model_gpu.py -x -q -m gpu 2>&1 | grep -v Warn | tail -4
for l in old new; do
  f=merlot_amd/libmerlot_hip.so; [ $l = old ] && f=merlot_amd/libmerlot_hip_prev.so
  echo "== im2col / col2im / groupnorm forward, $l"
  AB_LIB=$f timeout 300 python scripts/exp_im2col.py 2>&1 | grep '^\['
done
for l in old new new old; do
  f=merlot_amd/libmerlot_hip.so; [ $l = old ] && f=merlot_amd/libmerlot_hip_prev.so
  echo "== resnet stem step, $l"
  AB_LIB=$f timeout 300 python scripts/bench_lib.py --resnet-stem --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  mfu %.3f nt %.3f  tn %.3f  fwd %.1f ms' % (r['value'], r['ms_per_step'], r['model_flops_utilization'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass']))"
done
) 2>&1 | tee gpurun_out/r04_s_im2col_rows.txt | cut -c1-300
