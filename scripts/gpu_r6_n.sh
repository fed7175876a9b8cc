#!/bin/bash
# round 6, call 15: producers' cost after the vector-ALU diet of the copy block; conversions beyond range; tests; step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python scripts/exp_f8_producers.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_n_f8_producers.txt
timeout 1200 python -m pytest tests/test_f8_tn_gpu.py -q -m gpu -s -k "epilogue or beyond" 2>&1 | grep -v Warning | grep "passed\|failed\|Error\|error\|assert\|e4m3:\|e5m2:\|FAILED" | cut -c1-600 | tee gpurun_out/r06_n_f8_tests.txt
for mode in none w1,w2,fuse,noa; do
  if [ $mode = none ]; then extra=""; else extra="--fp8-bwd $mode"; fi
  timeout 600 python bench.py --config 5 --no-cpu-baseline --steps 8 --warmup 4 $extra 2>/dev/null | tee gpurun_out/r06_n_bench5_$mode.json | cut -c1-330
done
