#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${T:-101376} CFGS=${CFGS:-21,22,22,21} timeout 600 python scripts/exp_skew.py > gpurun_out/${TAG:-r2}_p8_ab.log 2>&1; tail -8 gpurun_out/${TAG:-r2}_p8_ab.log | cut -c1-400
