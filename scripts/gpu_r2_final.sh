#!/bin/bash
# round-2 closing run on ONE box: GPU tests, smoke, bench line, rocprofv3 kernel stats + busy fraction, TCC traffic passes,
# SQ counter pass, training sanity
export TAG=${TAG:-r2z}
bash scripts/gpu_r2_check.sh 2>&1 | cut -c1-600
bash scripts/gpu_r2_prof.sh 2>&1 | tail -3
bash scripts/gpu_traffic.sh > /dev/null 2>&1; head -7 gpurun_out/r02_traffic.txt
bash scripts/gpu_pmc.sh > gpurun_out/${TAG}_pmc_sq.txt 2>&1; grep -A2 "gemm_nt_p8_kernel<0, false, false>\|gemm_tn_p8" gpurun_out/${TAG}_pmc_sq.txt | cut -c1-400 | head -12
timeout 300 python scripts/train_sanity.py > gpurun_out/${TAG}_train_sanity.txt 2>&1; tail -3 gpurun_out/${TAG}_train_sanity.txt
