#!/bin/bash
# round 6, call 8: the LayerNorm kernels with a grid-stride loop + the next row requested ahead: same-box A/B against the library built before the change, the
# LayerNorm tests, step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
  AB_LIB=merlot_amd/libmerlot_hip_old.so python scripts/exp_ln_prefetch.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a gpurun_out/r06_h_ln_prefetch.txt
  python scripts/exp_ln_prefetch.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a gpurun_out/r06_h_ln_prefetch.txt
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -q -m gpu -k "layernorm or ln_ or q8 or fp8" 2>&1 | grep -v Warning | tail -4 | cut -c1-300 | tee gpurun_out/r06_h_ln_tests.txt
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline --no-ln-fold 2>/dev/null | tee gpurun_out/r06_h_bench_new$i.json | cut -c1-200
done
