#!/bin/bash
# round 6, call 6: the LayerNorm fold with a workspace block per row count and a pipelined row-block pass: kernel + model tests, launch A/B, timeline, step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_ln_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -12 | cut -c1-300 | tee gpurun_out/r06_f_ln_fold_tests.txt
timeout 600 python scripts/exp_ln_fold.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_f_ln_fold_ab.txt
timeout 300 python scripts/exp_ln_fold_trace.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_f_ln_fold_trace.txt
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_f_bench_fold_on$i.json | cut -c1-200
timeout 400 python bench.py --no-cpu-baseline --no-ln-fold 2>/dev/null | tee gpurun_out/r06_f_bench_fold_off$i.json | cut -c1-200
done
