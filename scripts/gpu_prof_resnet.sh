#!/bin/bash
# rocprofv3 kernel stats of the ResNet-hybrid-stem variant of the bench (not the headline config)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --resnet-stem --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_resnet.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $R/gpurun_out/resnet_kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/resnet_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
