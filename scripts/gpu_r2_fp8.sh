#!/bin/bash
# fp8 path: parity tests, then fp8-vs-bf16 GEMM timing at the config-#2 and config-#5 shapes
mkdir -p gpurun_out
T=${TAG:-r2fp8}
timeout 600 python -m pytest tests/test_fp8_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/${T}_tests.log
cat gpurun_out/${T}_tests.log
timeout 300 python scripts/exp_fp8.py > gpurun_out/${T}_gemm.log 2>&1
cat gpurun_out/${T}_gemm.log
