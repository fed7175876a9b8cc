#!/bin/bash
# round 4, call V: stem after the implicit convolutions: the 64-filter weight-gradient tile, the batch sweep again (the patch matrices are gone: 168 GB at
# 64 examples), kernel statistics of the stem step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_stem_kernels_gpu.py -x -q -m gpu 2>&1 | grep -v Warn | tail -2
FRAMES=1024 timeout 300 python scripts/exp_conv_implicit.py 2>&1 | grep '^\[' | cut -c1-400
for ex in 64 80 96; do
echo "== resnet stem step, $ex examples"
timeout 400 python bench.py --resnet-stem --examples $ex --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{\|OutOfMemory\|out of memory' | head -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    r=json.loads(t)
    print('value %.1f seg/s  %.1f ms/step  mfu %.3f  peak %.0f GB' % (r['value'], r['ms_per_step'], r['model_flops_utilization'], r['hbm']['peak_allocated_gb']))
except Exception: print('no result:', t[:200])"
done
) 2>&1 | tee gpurun_out/r04_v_stem_batch.txt | cut -c1-400
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/prof_stem
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stem -o bench -- python $R/bench.py --resnet-stem --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cp $(find /tmp/prof_stem -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r04_v_stem_kernel_stats.csv 2>/dev/null; head -30 $R/gpurun_out/r04_v_stem_kernel_stats.csv | cut -c1-150
