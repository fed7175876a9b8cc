#!/bin/bash
# round 6, call 25: the attention log from the tiled dK / dV kernel's own P (config #5's joint encoder: 12 column-sum launches of 2.85 ms per step gone); tests, config-#5 A/B, traffic file
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_kernels_gpu.py tests/test_f8_tn_gpu.py tests/test_fp8_gpu.py tests/test_edge_cases_gpu.py tests/test_model_gpu.py -q -m gpu 2>&1 | grep -v Warning | grep "passed\|failed\|Error\|error\|assert\|FAILED" | cut -c1-400 | tee gpurun_out/r06_w_tests.txt
for mode in default bf16 default bf16; do
  if [ $mode = bf16 ]; then extra="--bf16"; else extra=""; fi
  timeout 600 python bench.py --config 5 --no-cpu-baseline --steps 8 --warmup 4 $extra 2>/dev/null | tee -a gpurun_out/r06_w_bench5_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done
bash scripts/gpu_traffic.sh > /dev/null 2>&1; grep "hash" gpurun_out/r06_traffic.txt | head -3
