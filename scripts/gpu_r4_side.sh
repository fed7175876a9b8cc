#!/bin/bash
# round 4: the side configurations on the closing build: config #5 (bf16 / fp8) and the ResNet-hybrid stem (+ its kernel statistics)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --config 5 --bf16 --no-cpu-baseline > gpurun_out/r04_y_bench_config5_bf16.json 2>/dev/null
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/r04_y_bench_config5_fp8.json 2>/dev/null
timeout 600 python bench.py --resnet-stem --no-cpu-baseline > gpurun_out/r04_y_bench_resnet_stem.json 2>/dev/null
python - <<'PY'
import json
for f in ('config5_bf16','config5_fp8','resnet_stem'):
    d=json.load(open('gpurun_out/r04_y_bench_%s.json'%f))
    print(f, round(d['value'],1), round(d['ms_per_step'],1), round(d['model_flops_utilization'],4), round(d['roofline']['frac'],3), d.get('roofline_fp8',{}).get('frac'))
PY
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/prof_stem
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stem -o bench -- python $R/bench.py --resnet-stem --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cp $(find /tmp/prof_stem -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r04_y_stem_kernel_stats.csv 2>/dev/null; head -12 $R/gpurun_out/r04_y_stem_kernel_stats.csv | cut -c1-160
