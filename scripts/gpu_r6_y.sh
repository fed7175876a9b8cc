#!/bin/bash
# round 6, call 35: GroupNorm in one launch per direction WITHOUT device-scope fences (call 34: release / acquire per wave = an L2 write-back + invalidate each; the
# as-shipped step ran into its 600 s timeout), per-shape A/B first and under a short timeout; the fused attention backward's chunked arrival (CH): tests, A/B; step A/Bs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -u scripts/exp_gn_fused.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_y_gn_fused.txt
timeout 600 python -m pytest tests/test_stem_kernels_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -6 | cut -c1-300 | tee gpurun_out/r06_y_tests.txt
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_native_yaml_gpu.py tests/test_edge_cases_gpu.py -q -m gpu -k "attention or native or shipped or geometry" 2>&1 | grep -v Warning | tail -6 | cut -c1-300 | tee -a gpurun_out/r06_y_tests.txt
timeout 400 python -u scripts/exp_attn_fb_ch.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_y_attn_fb_ch.txt
for mode in fused two fused two; do
  if [ $mode = two ]; then extra="--no-gn-fused"; else extra=""; fi
  timeout 300 python bench.py --native-yaml --no-cpu-baseline --steps 6 --warmup 3 $extra 2>/dev/null | tee gpurun_out/r06_y_bench_native_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_y_bench$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms')"; done
