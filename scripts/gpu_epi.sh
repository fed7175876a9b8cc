#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
MERLOT_NT_CFG=23 timeout 300 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "gemm or linear" 2>&1 | tail -5
cd scripts
CFGS=21,23 timeout 300 python exp_epi2.py 2>&1 | grep -v amdgpu.ids | tee ../gpurun_out/exp_epi_pingpong.txt
cd ..
for id in 21 23; do MERLOT_NT_PERSIST_ID=$id timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230; done
