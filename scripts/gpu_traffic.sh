#!/bin/bash
# HBM traffic of the GEMM kernels from the TCC counters (separate passes, as MI355X_MICROARCH.md prescribes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o g -- python $R/scripts/pmc_gemm.py > $R/gpurun_out/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/g_counter_collection.csv $R/gpurun_out/pmc_$c.csv
done
python - <<PY
import csv, collections
for c in ['FETCH_SIZE', 'WRITE_SIZE']:
    rows = list(csv.DictReader(open('$R/gpurun_out/pmc_%s.csv' % c)))
    agg = collections.OrderedDict()
    for r in rows:
        k = r['Kernel_Name'][:80]
        if 'gemm' not in k and 'attn' not in k: continue
        d = agg.setdefault(k, [0.0, 0])
        d[0] += float(r['Counter_Value']); d[1] += 1
    for k, (v, n) in agg.items():
        print(c, k, 'per-launch KB:', v / n)
PY
