#!/bin/bash
# round 6, call 13: the 8-bit copies from the producing launches ('fuse'): kernel tests, model test, config-#5 step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_f8_tn_gpu.py -q -m gpu -s -k "not config5_geometry_fp8_backward_gradients" 2>&1 | grep -v Warning | grep "passed\|failed\|Error\|error\|assert\|fp8_backward\|FAILED" | cut -c1-400 | tee gpurun_out/r06_l_f8_tests.txt
for mode in none w1,w2,fuse w1,w2,fuse,noa w1,w2,wqkv,fuse,noa; do
  if [ $mode = none ]; then extra=""; else extra="--fp8-bwd $mode"; fi
  timeout 600 python bench.py --config 5 --no-cpu-baseline --steps 8 --warmup 4 $extra 2>gpurun_out/r06_l_err_$mode.txt | tee gpurun_out/r06_l_bench5_$mode.json | cut -c1-330
  tail -3 gpurun_out/r06_l_err_$mode.txt | cut -c1-300
done
