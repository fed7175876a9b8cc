#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_native_yaml_gpu.py -x -q -m gpu -s 2>&1 | grep -v Warning | tail -25 | tee gpurun_out/r05_l_native_tests.txt
