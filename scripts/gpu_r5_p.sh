#!/bin/bash
# round 5: the two-half persistent forward without a mask (the as-shipped ViT length): attention tests, the as-shipped model tests, the as-shipped bench line,
# and the TCC traffic file re-measured (the attention sources changed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_native_yaml_gpu.py -x -q -m gpu -k "attention or native or shipped or frame" 2>&1 | grep -v Warning | tail -4 | tee gpurun_out/r05_x_ppm_unmasked.txt
timeout 300 python bench.py --native-yaml --no-cpu-baseline 2>/dev/null | tee gpurun_out/r05_x_bench_native_yaml.json | cut -c1-330
bash scripts/gpu_traffic.sh > /dev/null 2>&1; grep "HBM_MB.*attn\|hash" gpurun_out/r05_traffic.txt
