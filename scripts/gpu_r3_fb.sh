#!/bin/bash
# fused short-sequence attention backward: bit-identity + timing (experiments build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python scripts/exp_attn_fb.py > gpurun_out/r03_h_attention_fb.txt 2>&1
tail -15 gpurun_out/r03_h_attention_fb.txt
