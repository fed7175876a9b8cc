#!/bin/bash
# full GPU pass: all -m gpu tests, smoke, full bench, rocprof kernel stats of the same command
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py --steps ${STEPS:-6} --warmup 3 ${EXTRA:---no-cpu-baseline} > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log | cut -c1-3000
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_bench.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $R/gpurun_out/kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
