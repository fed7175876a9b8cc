#!/bin/bash
# round 6, call 7: the LayerNorm-fold kernel with its epilogue operands from the kernel-argument segment (call 6 ran the build before that change: a host-pass
# compile error had left the older library in place), and the same-box step A/B of the two round-6 switches: LayerNorm fold on / off, weight-gradient column sums on / off
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_ln_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -4 | cut -c1-300 | tee gpurun_out/r06_g_ln_fold_tests.txt
timeout 600 python scripts/exp_ln_fold.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_g_ln_fold_ab.txt
timeout 300 python scripts/exp_ln_fold_trace.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_g_ln_fold_trace.txt
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline --no-ln-fold 2>/dev/null | tee gpurun_out/r06_g_bench_base$i.json | cut -c1-200
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_g_bench_fold_on$i.json | cut -c1-200
timeout 400 python bench.py --no-cpu-baseline --no-ln-fold --no-tn-colsum 2>/dev/null | tee gpurun_out/r06_g_bench_no_tn_colsum$i.json | cut -c1-200
done
