#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -12
cd scripts
CFGS=21,11 timeout 300 python exp_epi2.py 2>&1 | grep -v amdgpu.ids | tee ../gpurun_out/exp_epi_fastmath.txt
cd ..
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
