#!/bin/bash
# the side configurations after the attention work: config #5 (bf16 / fp8) and the ResNet-hybrid stem
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --config 5 --bf16 --no-cpu-baseline > gpurun_out/r03_p_bench_config5_bf16.json 2>/dev/null
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/r03_p_bench_config5_fp8.json 2>/dev/null
timeout 600 python bench.py --resnet-stem --no-cpu-baseline > gpurun_out/r03_p_bench_resnet_stem.json 2>/dev/null
python - <<'PY'
import json
for f in ('config5_bf16','config5_fp8','resnet_stem'):
    d=json.load(open('gpurun_out/r03_p_bench_%s.json'%f))
    print(f, round(d['value'],1), round(d['ms_per_step'],1), round(d['model_flops_utilization'],4), round(d['roofline']['frac'],3), d.get('roofline_fp8',{}).get('frac'))
PY
