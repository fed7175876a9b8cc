#!/bin/bash
# round 5, call a, b: the persistent forward -- correctness against the fp32 reference and the resident kernel, A/B timing
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/scripts:$PWD/tests
python scripts/exp_attn_pp.py > gpurun_out/r05_x_attn_pp_bwd_final.txt 2>&1
tail -20 gpurun_out/r05_x_attn_pp_bwd_final.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -5 | tee -a gpurun_out/r05_x_attn_pp_bwd_final.txt
