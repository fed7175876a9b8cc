#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python scripts/exp_attn_res.py > gpurun_out/r03_j_attention_res.txt 2>&1
tail -8 gpurun_out/r03_j_attention_res.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -3
