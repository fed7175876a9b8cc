#!/bin/bash
# round 6, call 36: GroupNorm over groups of samples that fit the Infinity Cache (per-shape sweep, the as-shipped step A/B), the one-launch form per shape under its own
# timeout (which shapes stall), the fused attention backward's chunked arrival at more token counts (where the rule's threshold lies), tests of both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -u scripts/exp_gn_groups.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_z_gn_groups.txt
for i in 1 2 4 7 11 12; do timeout 45 python -u scripts/exp_gn_fused.py $i 2>&1 | grep -v "Warning\|amdgpu.ids\|^N = " ; [ ${PIPESTATUS[0]} = 124 ] && echo "shape $i: no result inside 45 s"; done | tee gpurun_out/r06_z_gn_fused_shapes.txt
timeout 400 python -u scripts/exp_attn_fb_ch.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_z_attn_fb_ch.txt
timeout 600 python -m pytest tests/test_stem_kernels_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -6 | cut -c1-300 | tee gpurun_out/r06_z_tests.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_native_yaml_gpu.py tests/test_stem_model_gpu.py -q -m gpu -k "attention or native or shipped or stem" 2>&1 | grep -v Warning | tail -6 | cut -c1-300 | tee -a gpurun_out/r06_z_tests.txt
for mb in 0 64 64 0; do
  MERLOT_GN_GROUP_MB=$mb timeout 300 python bench.py --native-yaml --exp-lib --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | tee gpurun_out/r06_z_bench_native_mb$mb.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('groups of $mb MB:', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done
timeout 300 python bench.py --native-yaml --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | tee gpurun_out/r06_z_bench_native_product.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product library:', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms')"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_z_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms')"
