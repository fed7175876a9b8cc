#!/bin/bash
# round 4, call I: the lock-step persistent kernels retired (K = 64 -> 128x256 ring kernel, >= 4 GiB operands -> row ranges of the ping-pong
# kernel): GEMM + stem tests, stem and headline step A/B against the previous build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gemm_persist_gpu.py tests/test_stem_kernels_gpu.py tests/test_stem_model_gpu.py tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | grep -v Warn | tail -4
for l in old new new old; do
  f=merlot_amd/libmerlot_hip.so; [ $l = old ] && f=merlot_amd/libmerlot_hip_prev.so
  echo "== resnet stem step, $l"
  AB_LIB=$f timeout 300 python scripts/bench_lib.py --resnet-stem --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  mfu %.3f nt %.3f  tn %.3f  fwd %.1f ms' % (r['value'], r['ms_per_step'], r['model_flops_utilization'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass']))"
done
for ex in 48 64; do
echo "== resnet stem step, new, $ex examples"
timeout 300 python bench.py --resnet-stem --examples $ex --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  mfu %.3f' % (r['value'], r['ms_per_step'], r['model_flops_utilization']))"
done
) 2>&1 | tee gpurun_out/r04_i_retire_persist.txt | cut -c1-300
