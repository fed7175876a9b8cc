#!/bin/bash
# round 5, last call: the whole GPU suite + smoke on the final tree, then rocprofv3 kernel statistics of the as-shipped configuration (bench.py --native-yaml)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -8 > gpurun_out/r05_last_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r05_last_pytest_gpu.txt
cat gpurun_out/r05_last_pytest_gpu.txt
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/prof_ny
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ny -o bench -- python $R/bench.py --native-yaml --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r05_last_native_yaml_prof.log 2>&1
f=$(find /tmp/prof_ny -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r05_last_native_yaml_kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/r05_last_native_yaml_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('bench.py --native-yaml --steps 3 --warmup 2 (5 train steps + 4 forward passes); total kernel ms', round(tot / 1e6, 1))
for r in rows[:28]:
    print(f"{r['Name'][:105]:105s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
