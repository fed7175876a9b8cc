#!/bin/bash
# HBM traffic of the implicit 3x3 convolution kernels against the explicit path, from the TCC counters (separate --pmc passes with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes).  Writes gpurun_out/r04_traffic_conv.txt.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcc_$c -o g -- python $R/scripts/pmc_conv.py > $R/gpurun_out/pmcc_$c.log 2>&1
  cp /tmp/pmcc_$c/g_counter_collection.csv $R/gpurun_out/pmcc_$c.csv
done
cd $R
python - <<'PY' > gpurun_out/r04_traffic_conv.txt
import csv, collections
print('# rocprofv3 --pmc <counter> --kernel-trace -- python scripts/pmc_conv.py   (one pass per counter; MI355X, gfx950); 1 024 frames; launches in order: shape (56^2, 64 -> 64) then (28^2, 256 -> 256)')
print('# KB = TCC counter value per launch as reported; HBM bytes = (2*FETCH + WRITE) * 1024 (the gfx950 correction of MI355X_MICROARCH.md)')
vals = collections.OrderedDict()
for c in ['FETCH_SIZE', 'WRITE_SIZE']:
    rows = list(csv.DictReader(open('gpurun_out/pmcc_%s.csv' % c)))
    seen = collections.Counter()
    for r in rows:
        k = r['Kernel_Name']
        if not any(t in k for t in ('conv3x3', 'gemm_', 'im2col3x3', 'conv_wgrad_reduce', 'tn_reduce')):
            continue
        k = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:80]
        seen[k] += 1
        shape = 'A' if seen[k] <= 2 and 'ring::Cfg<2, 4' not in k or False else None
        vals.setdefault(k, {}).setdefault(c, []).append(float(r['Counter_Value']))
for k, d in vals.items():
    f, w = d.get('FETCH_SIZE', []), d.get('WRITE_SIZE', [])
    for i in range(min(len(f), len(w))):
        print('%-72s launch %d | FETCH_KB %12.1f | WRITE_KB %12.1f | HBM_MB %9.1f' % (k, i, f[i], w[i], (2 * f[i] + w[i]) / 1024.0))
PY
cat gpurun_out/r04_traffic_conv.txt
