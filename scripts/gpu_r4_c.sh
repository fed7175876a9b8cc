#!/bin/bash
# round 4, call C: gradient agreement by tensor class (HIP vs emulation vs oracle), the forced-RCCL step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python scripts/exp_grad_parity.py config1 2>&1 | tail -140 ) > gpurun_out/r04_c_grad_config1.txt 2>&1
( timeout 900 python scripts/exp_grad_parity.py config2x8 2>&1 | tail -140 ) > gpurun_out/r04_c_grad_config2x8.txt 2>&1
( timeout 600 python -m pytest tests/test_force_dist_gpu.py -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r04_c_force_dist.txt 2>&1
tail -8 gpurun_out/r04_c_grad_config1.txt; tail -8 gpurun_out/r04_c_grad_config2x8.txt; tail -15 gpurun_out/r04_c_force_dist.txt
