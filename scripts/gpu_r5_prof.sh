#!/bin/bash
# round 5: rocprofv3 kernel statistics of the bench command (train steps only: the per-kernel shares of ONE training step)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TAG=${TAG:-r05_p}
bash scripts/gpu_r2_prof.sh > gpurun_out/${TAG}_prof_summary.txt 2>&1
head -45 gpurun_out/${TAG}_prof_summary.txt | cut -c1-190
