#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/scripts:$PWD/tests
python scripts/exp_attn_pp_contention.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_v_attn_pp_contention_claims.txt
