#!/bin/bash
# round 2: ping-pong persistent NT kernel (id 22) -- correctness, then A/B timing against id 21 (mirrored order)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/exp_p8.py > gpurun_out/${TAG:-r2}_p8_correct.log 2>&1; tail -15 gpurun_out/${TAG:-r2}_p8_correct.log
for T in 101376 41984; do
  T=$T CFGS=21,22,22,21 timeout 600 python scripts/exp_skew.py > gpurun_out/${TAG:-r2}_p8_ab_$T.log 2>&1; tail -8 gpurun_out/${TAG:-r2}_p8_ab_$T.log
done
