#!/bin/bash
# round 5, on the shipped build (ABI v7): the all-reduce-shaped CU contention of a training step emulated on one GPU (as profiles/r04_o), then BASELINE
# config #5 (bf16 / fp8 forward GEMMs) and the training sanity run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
MERLOT_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 900 python scripts/exp_dp_contention.py 2>&1 | grep -v "^\[W\|amdgpu.ids" > gpurun_out/r05_w_dp_contention.txt
cat gpurun_out/r05_w_dp_contention.txt
TAG=r05_z bash scripts/gpu_r5_extra.sh
