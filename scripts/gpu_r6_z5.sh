#!/bin/bash
# round 6, call 41: the one-launch GroupNorm BACKWARD rebuilt on per-slice slots (ABI v11: write-through stores + one arrival per workgroup instead of 4C returning atomics): tests, per shape
# against the two-launch backward (one process per shape, own timeout), the as-shipped step default (one-launch forward only) | --gn-fused (both directions), mirrored
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_stem_kernels_gpu.py -q -m gpu -x 2>&1 | grep -v Warning | tail -5 | cut -c1-300 | tee gpurun_out/r06_z5_tests.txt
for i in 0 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 50 python -u scripts/exp_gn_fused_fwd.py $i bwd1 2>&1 | grep -v "Warning\|amdgpu.ids" ; [ ${PIPESTATUS[0]} = 124 ] && echo "shape $i: no result inside 50 s"; done | tee gpurun_out/r06_z5_gn_fused_bwd.txt
for mode in default both both default; do
  if [ $mode = both ]; then extra="--gn-fused"; else extra=""; fi
  timeout 240 python bench.py --native-yaml --no-cpu-baseline --steps 6 --warmup 3 $extra 2>/dev/null | tee gpurun_out/r06_z5_bench_native_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done 2>&1 | tee gpurun_out/r06_z5_native_ab.txt
timeout 600 python -m pytest tests/test_stem_model_gpu.py tests/test_native_yaml_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -3 | cut -c1-300 | tee -a gpurun_out/r06_z5_tests.txt
