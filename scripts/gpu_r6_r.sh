#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_edge_cases_gpu.py tests/test_gemm_ln_gpu.py -q -m gpu -s -k "8bit_paths or fold_agree" 2>&1 | grep -v Warning | grep "passed\|failed\|Error\|error\|assert\|8-bit weight\|FAILED" | cut -c1-600 | tee gpurun_out/r06_r_tests.txt
timeout 600 python bench.py --config 5 --no-cpu-baseline --examples 60 2>/dev/null | tee gpurun_out/r06_r_bench5_ex60.json | cut -c1-200
timeout 600 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_r_bench5_default.json | cut -c1-200
