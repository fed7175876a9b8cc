#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_reference_shim_gpu.py -q --tb=short -p no:cacheprovider 2>&1 | tail -25
MERLOT_NT_CFG=21 timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "gemm or linear" 2>&1 | tail -8
MERLOT_NT_PERSIST_DYN=1 timeout 600 python -m pytest tests/test_model_gpu.py -q --tb=short -p no:cacheprovider 2>&1 | tail -5
timeout 300 python scripts/exp_persist_dyn.py 2>&1 | tee gpurun_out/exp_persist_dyn.txt | tail -20
for d in 0 1; do MERLOT_NT_PERSIST_DYN=$d timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400; done
