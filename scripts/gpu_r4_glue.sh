#!/bin/bash
# round 4: torch glue off the step (ClsAvgPoolFn's second output, one reduction per period in GatherAddFn, slice-cast instead of gather + cast,
# SplitHiddenFn): model tests, the per-op table of one step again, step time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reference_shim_gpu.py tests/test_edge_cases_gpu.py tests/test_grad_classes_gpu.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
timeout 600 python scripts/exp_glue_sites.py 2>&1 | grep -v Warn | grep "one step\|x aten" | head -30
for i in 1 2; do timeout 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('value %.1f  %.1f ms  mfu %.4f' % (r['value'], r['ms_per_step'], r['model_flops_utilization']))"; done
) 2>&1 | tee gpurun_out/r04_glue.txt | cut -c1-230
