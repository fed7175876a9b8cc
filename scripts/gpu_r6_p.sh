#!/bin/bash
# round 6, call 17: regression of the whole GPU suite on the ABI v9 build, the headline line (unchanged kernels), the headline GEOMETRY with the 8-bit paths, training sanity on them
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -6 | cut -c1-300 | tee gpurun_out/r06_p_pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_p_bench.json | cut -c1-260
timeout 600 python bench.py --no-cpu-baseline --fp8 2>/dev/null | tee gpurun_out/r06_p_bench_c2_fp8fwd.json | cut -c1-330
timeout 600 python bench.py --no-cpu-baseline --fp8 --fp8-bwd w1,w2,fuse,noa,dgrad1 2>/dev/null | tee gpurun_out/r06_p_bench_c2_fp8all.json | cut -c1-400
timeout 600 python scripts/train_sanity.py 0.1 w1,w2,fuse,noa,dgrad1 8 2>&1 | grep -v "Warning\|amdgpu" | tail -16 | tee gpurun_out/r06_p_train_sanity_fp8.txt
timeout 600 python scripts/train_sanity.py 0.1 "" 8 2>&1 | grep -v "Warning\|amdgpu" | tail -5 | tee gpurun_out/r06_p_train_sanity_fp8fwd.txt
