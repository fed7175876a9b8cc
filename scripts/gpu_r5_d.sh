#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/scripts:$PWD/tests
timeout 1200 python scripts/exp_native_shapes.py > gpurun_out/r05_k_native_shapes.txt 2>&1
cat gpurun_out/r05_k_native_shapes.txt | grep -v Warning | tail -70
