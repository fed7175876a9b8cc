#!/bin/bash
# round 6, call 2: the LayerNorm fold (merlot_gemm_bf16_nt_ln, ABI v8): kernel tests, same-box A/B of the launches, the model-level tests that run through it,
# the default bench line with the fold on / off, and the depth-12 gradient measurement (HIP / emulation / oracle) behind the new test's bounds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_ln_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -15 | tee gpurun_out/r06_b_ln_fold_tests.txt
timeout 600 python scripts/exp_ln_fold.py 2>&1 | grep -v Warning | tee gpurun_out/r06_b_ln_fold_ab.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_edge_cases_gpu.py tests/test_grad_classes_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/r06_b_model_tests.txt
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_b_bench_fold_on.json | cut -c1-300
timeout 400 python bench.py --no-cpu-baseline --no-ln-fold 2>/dev/null | tee gpurun_out/r06_b_bench_fold_off.json | cut -c1-300
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_b_bench_fold_on2.json | cut -c1-300
timeout 1500 python scripts/exp_grad_parity.py config2d12 2>&1 | grep -v Warning | tail -150 > gpurun_out/r06_c_grad_depth12.txt; tail -8 gpurun_out/r06_c_grad_depth12.txt
