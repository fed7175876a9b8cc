#!/bin/bash
# round 6, call 4: the LayerNorm fold under the microscope -- which rows / elements differ from the two-launch composition (three shapes), the kernel tests, and the
# per-tile timeline of the fused launch (experiments build: arrival round trip per tile, LayerNorm pass per row block)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( M=65536 K=768 P=0.0 python scripts/exp_ln_fold_dbg.py; M=405504 K=768 P=0.1 python scripts/exp_ln_fold_dbg.py; M=167936 K=3072 P=0.1 python scripts/exp_ln_fold_dbg.py ) 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_d_dbg.txt | cut -c1-400
timeout 600 python -m pytest tests/test_gemm_ln_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -12 | cut -c1-300 | tee gpurun_out/r06_d_ln_fold_tests.txt
timeout 300 python scripts/exp_ln_fold_trace.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_d_ln_fold_trace.txt
