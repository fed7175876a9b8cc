#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --examples ${EX:-16} --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log | cut -c1-1500
cd /tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --examples ${EX:-16} --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_bench.log 2>&1
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-300
find /tmp/prof -type f | head
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
