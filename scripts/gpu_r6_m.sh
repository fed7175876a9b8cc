#!/bin/bash
# round 6, call 14: kernel statistics of the config-#5 step without / with the 8-bit backward ('fuse', 'fuse,noa'): which launches pay for the copies
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for mode in none w1,w2,fuse w1,w2,fuse,noa; do
  if [ $mode = none ]; then extra=""; else extra="--fp8-bwd $mode"; fi
  rm -rf /tmp/prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --config 5 --steps 3 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $R/gpurun_out/r06_m_prof_$mode.log 2>&1
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/r06_m_kernel_stats_$mode.csv 2>/dev/null
done
cd $R
python - <<PY
import csv
def load(m):
    rows = list(csv.DictReader(open('gpurun_out/r06_m_kernel_stats_%s.csv' % m)))
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs']) / 1e6) for r in rows}
base = load('none')
for m in ('w1,w2,fuse', 'w1,w2,fuse,noa'):
    cur = load(m)
    print('====', m, 'total kernel ms: %.1f -> %.1f' % (sum(v[1] for v in base.values()), sum(v[1] for v in cur.values())))
    names = sorted(set(base) | set(cur), key=lambda n: -abs(cur.get(n, (0, 0))[1] - base.get(n, (0, 0))[1]))
    for n in names[:26]:
        b, c = base.get(n, (0, 0.0)), cur.get(n, (0, 0.0))
        print(f'{n[:110]:110s} calls {b[0]:5d} -> {c[0]:5d}  ms {b[1]:8.2f} -> {c[1]:8.2f}  ({c[1] - b[1]:+8.2f})  avg us {1e3 * b[1] / max(b[0], 1):8.1f} -> {1e3 * c[1] / max(c[0], 1):8.1f}')
PY
