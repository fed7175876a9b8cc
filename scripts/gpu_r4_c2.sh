#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python scripts/exp_grad_parity.py config2x8 2>&1 | tail -140 ) > gpurun_out/r04_c_grad_config2x8.txt 2>&1
tail -8 gpurun_out/r04_c_grad_config2x8.txt
