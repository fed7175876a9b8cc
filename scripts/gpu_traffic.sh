#!/bin/bash
# HBM traffic of the dominant kernels from the TCC counters -- separate --pmc passes with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes -- over the PRODUCT library.  Writes gpurun_out/${TRAFFIC_OUT:-r06_traffic.txt}; copy it to profiles/.
# The file records the hashes of the GEMM sources AND of the attention sources (bench.py refuses the rows of a family whose hash differs
# from the tree's: round 4's file was measured before the attention kernels last changed and nothing said so) and the sha256 of the
# measured libmerlot_hip.so.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o g -- python $R/scripts/pmc_gemm.py > $R/gpurun_out/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/g_counter_collection.csv $R/gpurun_out/pmc_$c.csv
done
cd $R
python - <<'PY' > gpurun_out/r06_traffic.txt
import csv, collections, hashlib, os, sys
sys.path.insert(0, os.getcwd())
import bench
print('# rocprofv3 --pmc <counter> --kernel-trace -- python scripts/pmc_gemm.py   (one pass per counter; MI355X, gfx950)')
print('# KB = TCC counter value per launch as reported (FETCH_SIZE / WRITE_SIZE in KiB); HBM bytes = (2*FETCH + WRITE) * 1024')
print('#   (the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64-B units as if they were 32-B)')
print('gemm_source_hash', bench.gemm_source_hash())
print('attention_source_hash', bench.source_hash(bench.ATTENTION_SOURCES))
print('libmerlot_hip.so sha256', hashlib.sha256(open('merlot_amd/libmerlot_hip.so', 'rb').read()).hexdigest())
vals = {}
for c in ['FETCH_SIZE', 'WRITE_SIZE']:
    rows = list(csv.DictReader(open('gpurun_out/pmc_%s.csv' % c)))
    agg = collections.OrderedDict()
    for r in rows:
        k = r['Kernel_Name']
        if not any(t in k for t in ('gemm', 'attn', 'quantize', 'amax', 'tn_reduce', 'ln_fwd', 'ln_bwd')):
            continue
        k = k.replace('(anonymous namespace)::', '').split('(')[0][:90]
        d = agg.setdefault(k, [0.0, 0])
        d[0] += float(r['Counter_Value']); d[1] += 1
    for k, (v, n) in agg.items():
        vals.setdefault(k, {})[c] = v / n
        print('%s | %s | launches %d | KB_per_launch %.1f' % (c, k, n, v / n))
print('# kernel | HBM MB per launch = (2*FETCH + WRITE) KiB / 1024')
for k, d in vals.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        print('HBM_MB | %s | %.1f' % (k, (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) / 1024.0))
PY
cat gpurun_out/r06_traffic.txt
