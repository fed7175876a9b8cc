#!/bin/bash
# round 2: rocprofv3 kernel stats of the bench command (same flags as the bench line it accompanies)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_prof_bench.log 2>&1
tail -1 $R/gpurun_out/${TAG}_prof_bench.log | cut -c1-200
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/${TAG}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:34]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
