#!/bin/bash
# round 5: the persistent attention kernels' A/B lines only (scripts/exp_attn_pp.py), no test run
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/scripts:$PWD/tests
python scripts/exp_attn_pp.py > gpurun_out/${TAG:-r05_x_attn_pp_stagger}.txt 2>&1
grep "^bwd B  \|^bwd B   5\|^fwd B\|^masked fwd B   5" gpurun_out/${TAG:-r05_x_attn_pp_stagger}.txt
