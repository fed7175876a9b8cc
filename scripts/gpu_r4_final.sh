#!/bin/bash
# round-4 closing run on ONE box: GPU tests, smoke, bench line (with the CPU baseline), rocprofv3 kernel stats + busy fraction,
# TCC traffic passes (hash-stamped), SQ counter pass, training sanity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TAG=${TAG:-r04_x}
TAG=$TAG bash scripts/gpu_r2_check.sh 2>&1 | cut -c1-900
TAG=$TAG bash scripts/gpu_r2_prof.sh > gpurun_out/${TAG}_prof_summary.txt 2>&1; tail -3 gpurun_out/${TAG}_prof_summary.txt
bash scripts/gpu_traffic.sh > /dev/null 2>&1; head -8 gpurun_out/r04_traffic.txt
bash scripts/gpu_pmc.sh > gpurun_out/${TAG}_pmc_sq.txt 2>&1; grep -A2 "gemm_nt_p8_kernel<0, false, false\|gemm_tn_p1" gpurun_out/${TAG}_pmc_sq.txt | cut -c1-400 | head -12
timeout 300 python scripts/train_sanity.py > gpurun_out/${TAG}_train_sanity.txt 2>&1; tail -3 gpurun_out/${TAG}_train_sanity.txt
