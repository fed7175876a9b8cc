#!/bin/bash
# round 6, call 55: why the LayerNorm backward on 512 / 256 blocks is faster in isolation and slower in the step -- its in-step kernel times under rocprofv3 for both caps (experiments library)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for cap in 2048 512 2048 512; do
  rm -rf /tmp/prof_ln
  MERLOT_LN_BWD_BLOCKS=$cap timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ln -o b -- python $R/bench.py --exp-lib --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > /tmp/prof_ln.log 2>&1
  python - <<PY
import csv, glob, json
f = glob.glob('/tmp/prof_ln/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
ln = [r for r in rows if 'ln_bwd_kernel' in r['Name']]
line = json.loads([l for l in open('/tmp/prof_ln.log') if l.startswith('{')][-1])
print('cap $cap: step %.1f ms; all kernels %.1f ms; ln_bwd: ' % (line['ms_per_step'], tot / 1e6) + ' | '.join('%s calls avg %.1f us total %.1f ms' % (r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6) for r in ln))
PY
done | tee $R/gpurun_out/r06_z15_ln_bwd_in_step.txt
