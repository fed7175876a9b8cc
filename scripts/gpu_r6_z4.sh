#!/bin/bash
# round 6, call 39: the one-launch GroupNorm forward (residual read behind the wait, 16 positions per thread in every case) per shape; tests; the as-shipped step with the
# product default (one-launch forward, two-launch backward) against two launches in both directions, mirrored; the hybrid-stem line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 0 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 50 python -u scripts/exp_gn_fused_fwd.py $i 2>&1 | grep -v "Warning\|amdgpu.ids" ; [ ${PIPESTATUS[0]} = 124 ] && echo "shape $i: no result inside 50 s"; done | tee gpurun_out/r06_z4_gn_fused_fwd.txt
timeout 900 python -m pytest tests/test_stem_kernels_gpu.py tests/test_stem_model_gpu.py tests/test_native_yaml_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -4 | cut -c1-300 | tee gpurun_out/r06_z4_tests.txt
for mode in default two two default; do
  if [ $mode = two ]; then extra="--no-gn-fused"; else extra=""; fi
  timeout 240 python bench.py --native-yaml --no-cpu-baseline --steps 6 --warmup 3 $extra 2>/dev/null | tee gpurun_out/r06_z4_bench_native_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done 2>&1 | tee gpurun_out/r06_z4_native_ab.txt
for mode in default two; do
  if [ $mode = two ]; then extra="--no-gn-fused"; else extra=""; fi
  timeout 240 python bench.py --resnet-stem --no-cpu-baseline --steps 6 --warmup 3 $extra 2>/dev/null | tee gpurun_out/r06_z4_bench_stem_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resnet-stem $mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms')"
done 2>&1 | tee -a gpurun_out/r06_z4_native_ab.txt
