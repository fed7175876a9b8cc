#!/bin/bash
# round 6, call 12: the 8-bit weight gradients in the model (separate quantising passes, delayed scales): tests, config-#5 step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_f8_tn_gpu.py -q -m gpu -x -s 2>&1 | grep -v Warning | tail -15 | cut -c1-400 | tee gpurun_out/r06_k_f8_tests.txt
for mode in none w1,w2 w1,w2,wqkv,wproj; do
  if [ $mode = none ]; then extra=""; else extra="--fp8-bwd $mode"; fi
  timeout 600 python bench.py --config 5 --no-cpu-baseline --steps 8 --warmup 3 $extra 2>/dev/null | tee gpurun_out/r06_k_bench5_$mode.json | cut -c1-300
done
timeout 600 python bench.py --config 5 --bf16 --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tee gpurun_out/r06_k_bench5_bf16.json | cut -c1-300
