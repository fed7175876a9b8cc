#!/bin/bash
# round 6, call 56: HBM-side traffic of the GroupNorm forward, two launches against one (TCC counters, one --pmc pass per counter with --kernel-trace only), shape 5 of the as-shipped stem
# (48 x 88 x 128, 896 frames: x = y = 969 MB)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_gn_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_gn_$c -o g -- python $R/scripts/exp_gn_fused_fwd.py 5 > /tmp/pmc_gn_$c.log 2>&1
done
python - <<'PY' | tee $R/gpurun_out/r06_z16_gn_traffic.txt
import csv, glob, collections
print('# rocprofv3 --pmc <counter> --kernel-trace -- python scripts/exp_gn_fused_fwd.py 5   (48 x 88 x 128, 896 frames: one tensor = 969.0 MB); MB per launch = counter KiB * 1024 / 1e6;')
print('# FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950 (16-B-per-lane streaming reads are tallied at half their bytes)')
vals = collections.OrderedDict()
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('/tmp/pmc_gn_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gn_' not in k:
            continue
        k = k.replace('(anonymous namespace)::', '').split('(')[0][:60]
        d = vals.setdefault(k, {}).setdefault(c, [0.0, 0])
        d[0] += float(r['Counter_Value']); d[1] += 1
for k, v in vals.items():
    fe = v.get('FETCH_SIZE', [0, 1]); wr = v.get('WRITE_SIZE', [0, 1])
    print('%-62s launches %3d  read %8.1f MB  written %8.1f MB' % (k, fe[1], 2 * fe[0] / fe[1] * 1024 / 1e6, wr[0] / wr[1] * 1024 / 1e6))
PY
