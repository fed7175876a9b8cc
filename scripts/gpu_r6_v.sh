#!/bin/bash
# round 6, call 23: the fused 8-bit producers at ragged row counts; the whole f8 test file; config #5 at 60 and 64 examples
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_f8_tn_gpu.py tests/test_fp8_gpu.py tests/test_edge_cases_gpu.py -q -m gpu -s 2>&1 | grep -v Warning | grep "passed\|failed\|Error\|error\|assert\|ragged rows\|FAILED" | cut -c1-400 | tee gpurun_out/r06_v_tests.txt
for ex in 48 60 64; do
  timeout 600 python bench.py --config 5 --no-cpu-baseline --steps 8 --warmup 4 --examples $ex 2>/dev/null | tee gpurun_out/r06_v_bench5_ex$ex.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('examples $ex', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'peak GB', round(d['hbm']['peak_allocated_gb'],1))"
done
