#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v Warning | grep "FAILED\|passed\|failed\|Error\|assert " | head -40 | cut -c1-400 | tee gpurun_out/r06_q_pytest_gpu.txt
