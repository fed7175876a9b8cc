#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/exp_p8.py > gpurun_out/${TAG}_p8_correct.log 2>&1; tail -2 gpurun_out/${TAG}_p8_correct.log
T=101376 CFGS=21,22,22,21 timeout 600 python scripts/exp_skew.py > gpurun_out/${TAG}_p8_ab.log 2>&1; tail -7 gpurun_out/${TAG}_p8_ab.log | cut -c1-420
timeout 300 python scripts/exp_p8_ablate.py > gpurun_out/${TAG}_p8_ablate.log 2>&1; tail -3 gpurun_out/${TAG}_p8_ablate.log
