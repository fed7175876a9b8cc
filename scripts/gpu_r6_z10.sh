#!/bin/bash
# round 6, call 47: the two-launch GroupNorm backward with the new rule for its sums pass (512 / 256 blocks): does the APPLY pass (no trailing atomics) want its own, finer slicing?
# then the as-shipped step and the hybrid-stem line on the product library (new rule) for the A/B against call 39's numbers on this box: --exp-lib with MERLOT_GN_BLOCKS=2048 = the old rule
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 896 224; do timeout 200 python -u scripts/exp_gn_blocks.py $n 256,512,1024,2048,4096 apply 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee gpurun_out/r06_z10_gn_blocks_apply.txt
for mode in new old old new; do
  if [ $mode = old ]; then export MERLOT_GN_BLOCKS=2048; else unset MERLOT_GN_BLOCKS; fi
  timeout 240 python bench.py --native-yaml --exp-lib --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | tee gpurun_out/r06_z10_bench_native_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode rule:', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done 2>&1 | tee gpurun_out/r06_z10_native_ab.txt
unset MERLOT_GN_BLOCKS
timeout 600 python -m pytest tests/test_stem_kernels_gpu.py tests/test_stem_model_gpu.py tests/test_native_yaml_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -3 | cut -c1-300 | tee gpurun_out/r06_z10_tests.txt
