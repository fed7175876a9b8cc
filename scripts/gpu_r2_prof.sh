#!/bin/bash
# round 2: rocprofv3 kernel stats of the bench command (same flags as the bench line it accompanies) + GPU busy fraction
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/${TAG}_prof_bench.log 2>&1
tail -1 $R/gpurun_out/${TAG}_prof_bench.log | cut -c1-200
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
t=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/${TAG}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
# busy fraction over the last 5 training steps: window = from the 5th-last adamw kernel's end ... last adamw end
tr = list(csv.DictReader(open('$t')))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in tr)
adam = [e for e in ev if 'adamw_kernel' in e[2]]
if len(adam) >= 6:
    t0, t1 = adam[-6][1], adam[-1][1]
    busy, cur_s, cur_e, n = 0, None, None, 0
    for s, e, k in ev:
        if e <= t0 or s >= t1: continue
        n += 1
        s, e = max(s, t0), min(e, t1)
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f'window {1e-6 * (t1 - t0):.2f} ms over 5 steps, {n} kernels; GPU busy {1e-6 * busy:.2f} ms = {100.0 * busy / (t1 - t0):.1f} %; idle per step {1e-6 * (t1 - t0 - busy) / 5:.2f} ms')
PY
