#!/bin/bash
# round 4, call T (run twice: forward + input gradient, then + weight gradient): the stem's 3x3 convolutions as implicit GEMMs (csrc/conv_gemm.hip): kernel + model tests, per-layer times against the explicit
# path, stem step A/B (same library, bench.py --resnet-stem with / without --explicit-conv)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_stem_kernels_gpu.py tests/test_stem_model_gpu.py -x -q -m gpu 2>&1 | grep -v Warn | tail -6
timeout 600 python scripts/exp_conv_implicit.py 2>&1 | grep '^\['
for l in explicit implicit implicit explicit; do
  fl=""; [ $l = explicit ] && fl="--explicit-conv"
  echo "== resnet stem step, $l"
  timeout 300 python bench.py --resnet-stem $fl --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  mfu %.3f nt %.3f  tn %.3f  fwd %.1f ms  peak %.0f GB' % (r['value'], r['ms_per_step'], r['model_flops_utilization'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass'], r['hbm']['peak_allocated_gb']))"
done
) 2>&1 | tee gpurun_out/r04_u_conv_wgrad_implicit.txt | cut -c1-300
