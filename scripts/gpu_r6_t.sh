#!/bin/bash
# round 6, call 21: dqkv's 8-bit copy from the tiled attention backward's own launches (merlot_attention_bwd_q8); tests; config-#5 A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_f8_tn_gpu.py tests/test_kernels_gpu.py -q -m gpu -s -k "attention_bwd_copy or fused_fp8_backward or attention" 2>&1 | grep -v Warning | grep "passed\|failed\|Error\|error\|assert\|FAILED" | cut -c1-500 | tee gpurun_out/r06_t_tests.txt
for mode in default bf16 default; do
  if [ $mode = bf16 ]; then extra="--bf16"; else extra=""; fi
  timeout 600 python bench.py --config 5 --no-cpu-baseline --steps 8 --warmup 4 $extra 2>/dev/null | tee -a gpurun_out/r06_t_bench5_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'), d['config'].get('fp8_backward'))"
done
