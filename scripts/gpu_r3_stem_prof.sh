#!/bin/bash
# rocprofv3 kernel stats of bench.py --resnet-stem
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp; rm -rf /tmp/prof_stem
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stem -o bench -- python $R/bench.py --resnet-stem --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r3_stem_prof.log 2>&1
tail -1 $R/gpurun_out/r3_stem_prof.log | cut -c1-300
f=$(find /tmp/prof_stem -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r3_stem_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/r3_stem_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:32]:
    print(f"{r['Name'][:110]:110s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
