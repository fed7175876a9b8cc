#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/scripts:$PWD/tests
python scripts/exp_attn_pp_trace.py > gpurun_out/r05_i_attn_pp_trace_prio.txt 2>&1
cat gpurun_out/r05_i_attn_pp_trace_prio.txt | tail -30
