#!/bin/bash
# round 6, call 10: the fused attention backward with one block per wave (twelve waves) for 257 .. 384 tokens: attention tests, A/B against the eight-wave kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_native_yaml_gpu.py -q -m gpu -k "attention or native or shipped" 2>&1 | grep -v Warning | tail -5 | cut -c1-300 | tee gpurun_out/r06_i_attn_tests.txt
timeout 600 python scripts/exp_attn_fb_bpw.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_i_attn_fb_bpw.txt
for i in 1 2; do timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_i_bench$i.json | cut -c1-200; done
