#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/scripts:$PWD/tests
python scripts/exp_epi_bound.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_t_epilogue_bound.txt
