#!/bin/bash
# round 5: the fused attention backward before / after the rewrite of its dK / dV pass, same box, alternating (libmerlot_hip_exp_old.so = the build before)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/scripts:$PWD/tests
for i in 1 2; do
  EXP_LIB=$PWD/merlot_amd/libmerlot_hip_exp_old.so python scripts/exp_attn_fb_ab.py 2>&1 | grep -v amdgpu.ids
  python scripts/exp_attn_fb_ab.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05_x_attn_fb_ab.txt
cat gpurun_out/r05_x_attn_fb_ab.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -3 | tee -a gpurun_out/r05_x_attn_fb_ab.txt
