#!/bin/bash
# round 6, call 34: GroupNorm in one launch per direction (ABI v10) -- kernel tests, per-shape A/B, the as-shipped step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stem_kernels_gpu.py -q -m gpu -x 2>&1 | grep -v Warning | tail -15 | cut -c1-300 | tee gpurun_out/r06_x_tests.txt
timeout 600 python scripts/exp_gn_fused.py 2>&1 | grep -v Warning | tee gpurun_out/r06_x_gn_fused.txt
timeout 900 python -m pytest tests/test_stem_model_gpu.py tests/test_native_yaml_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -8 | cut -c1-300 | tee -a gpurun_out/r06_x_tests.txt
for mode in fused two fused two; do
  if [ $mode = two ]; then extra="--no-gn-fused"; else extra=""; fi
  timeout 600 python bench.py --native-yaml --no-cpu-baseline --steps 6 --warmup 3 $extra 2>/dev/null | tee gpurun_out/r06_x_bench_native_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done
