#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp; rm -rf /tmp/pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc -o g -- python $R/scripts/pmc_gemm.py > $R/gpurun_out/pmc.log 2>&1
ls /tmp/pmc
cp /tmp/pmc/g_counter_collection.csv $R/gpurun_out/pmc_counters.csv 2>/dev/null
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$R/gpurun_out/pmc_counters.csv')))
print(rows[0].keys())
agg = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name'][:70]
    if 'gemm' not in k and 'attn' not in k: continue
    d = agg.setdefault(k, collections.defaultdict(float))
    d[r['Counter_Name']] += float(r['Counter_Value'])
    d['_n'] += 1
for k, d in agg.items():
    wc = d['SQ_WAVE_CYCLES'] or 1
    print(k)
    print('   ', {n: f"{v:.3g}" for n, v in d.items()})
    print(f"    wait_any/wave {d['SQ_WAIT_ANY']/wc:.2f} wait_inst_any {d['SQ_WAIT_INST_ANY']/wc:.2f} wait_lds {d['SQ_WAIT_INST_LDS']/wc:.2f} active {d['SQ_ACTIVE_INST_ANY']/wc:.2f} bankconf/wave {d['SQ_LDS_BANK_CONFLICT']/wc:.3f} mfma_busy/busy {d['SQ_VALU_MFMA_BUSY_CYCLES']/(d['SQ_BUSY_CYCLES'] or 1):.3f}")
PY
