#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python bench.py --native-yaml --steps 4 --warmup 2 --no-cpu-baseline 2> gpurun_out/r05_m_native_err.txt > gpurun_out/r05_m_bench_native_yaml.json
tail -3 gpurun_out/r05_m_native_err.txt; cut -c1-1500 gpurun_out/r05_m_bench_native_yaml.json
