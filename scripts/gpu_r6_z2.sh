#!/bin/bash
# round 6, call 37: (1) the one-launch GroupNorm with its arrival counters 4 KiB apart (call 36: shapes 96x176x64 and 48x88x256 gave no result in 45 s with the counters 4 B apart), per
# shape under its own timeout, product library (whole-batch two-launch entries beside it), (2) GroupNorm per-shape sweep of the sample groups (experiments build; call 36's script stopped
# at an over-strict comparison), (3) the as-shipped step: two-launch | one-launch, (4) tests, the headline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 0 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 60 python -u scripts/exp_gn_fused.py $i 2>&1 | grep -v "Warning\|amdgpu.ids\|^N = " ; [ ${PIPESTATUS[0]} = 124 ] && echo "shape $i: no result inside 60 s"; done | tee gpurun_out/r06_z2_gn_fused_shapes.txt
timeout 600 python -u scripts/exp_gn_groups.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_z2_gn_groups.txt
timeout 600 python -m pytest tests/test_stem_kernels_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -6 | cut -c1-300 | tee gpurun_out/r06_z2_tests.txt
for mode in two fused fused two; do
  if [ $mode = fused ]; then extra="--gn-fused"; else extra=""; fi
  timeout 240 python bench.py --native-yaml --no-cpu-baseline --steps 6 --warmup 3 $extra 2>/dev/null | tee gpurun_out/r06_z2_bench_native_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_native_yaml_gpu.py tests/test_stem_model_gpu.py -q -m gpu -k "attention or native or shipped or stem" 2>&1 | grep -v Warning | tail -4 | cut -c1-300 | tee -a gpurun_out/r06_z2_tests.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_z2_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms')"
