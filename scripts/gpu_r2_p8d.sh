#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/exp_p8.py > gpurun_out/${TAG}_p8_correct.log 2>&1; tail -2 gpurun_out/${TAG}_p8_correct.log
T=101376 CFGS=21,22,22,21 timeout 600 python scripts/exp_skew.py > gpurun_out/${TAG}_p8_ab.log 2>&1; tail -7 gpurun_out/${TAG}_p8_ab.log | cut -c1-420
T=41984 CFGS=21,22,22,21 timeout 600 python scripts/exp_skew.py > gpurun_out/${TAG}_p8_ab2.log 2>&1; tail -7 gpurun_out/${TAG}_p8_ab2.log | cut -c1-420
timeout 900 python -m pytest tests/test_gemm_persist_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-400
