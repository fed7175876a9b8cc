#!/bin/bash
# round 6, call 3: the LayerNorm fold after its row-block reads became compiler-visible sc1 buffer loads (call 2: NaN rows -- and a step that ran 13 % FASTER on
# the NaNs, see DESIGN 3.1 "power"), the weight-gradient launch that also sums its A operand, the depth-12 gradient test with its measured bounds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_ln_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -15 | tee gpurun_out/r06_c_ln_fold_tests.txt
timeout 600 python -m pytest tests/test_gemm_persist_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "sums_its_a or gemm_tn or alpha_is" 2>&1 | grep -v Warning | tail -5 | tee gpurun_out/r06_c_tn_cs_tests.txt
timeout 600 python scripts/exp_ln_fold.py 2>&1 | grep -v Warning | tee gpurun_out/r06_c_ln_fold_ab.txt
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_c_bench_fold_on.json | cut -c1-300
timeout 400 python bench.py --no-cpu-baseline --no-ln-fold 2>/dev/null | tee gpurun_out/r06_c_bench_fold_off.json | cut -c1-300
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_c_bench_fold_on2.json | cut -c1-300
timeout 400 python bench.py --no-cpu-baseline --no-ln-fold 2>/dev/null | tee gpurun_out/r06_c_bench_fold_off2.json | cut -c1-300
timeout 900 python -m pytest tests/test_config2_depth12_gpu.py tests/test_zz_full_depth_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/r06_c_depth12_tests.txt
timeout 600 python scripts/exp_config5_grad.py 2>&1 | grep -v Warning | tail -12 | tee gpurun_out/r06_c_config5_grad.txt
