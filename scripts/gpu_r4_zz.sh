#!/bin/bash
# round 4, last call: the final build once more on one box -- whole GPU suite, smoke, default bench line (CPU baseline included), rocprofv3 kernel statistics of
# the same command, the stem line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TAG=r04_zz
TAG=$TAG bash scripts/gpu_r2_check.sh 2>&1 | cut -c1-700
TAG=$TAG bash scripts/gpu_r2_prof.sh > gpurun_out/${TAG}_prof_summary.txt 2>&1; head -3 gpurun_out/${TAG}_prof_summary.txt | cut -c1-200
timeout 600 python bench.py --resnet-stem --no-cpu-baseline 2>/dev/null > gpurun_out/${TAG}_bench_resnet_stem.json; cut -c1-200 gpurun_out/${TAG}_bench_resnet_stem.json
