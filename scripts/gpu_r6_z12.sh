#!/bin/bash
# round 6, call 50: the LayerNorm backward on 512 / 256 blocks (new) against <= 2 048 (MERLOT_LN_BWD_BLOCKS=2048, the cap of rounds 1 - 6): headline step, experiments library, mirrored; LayerNorm tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mode in new old old new; do
  if [ $mode = old ]; then export MERLOT_LN_BWD_BLOCKS=2048; else unset MERLOT_LN_BWD_BLOCKS; fi
  timeout 240 python bench.py --exp-lib --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_z12_bench_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode cap:', round(d['value'],1), 'seg/s', round(d['ms_per_step'],2), 'ms', 'loss', d['config'].get('final_loss'))"
done 2>&1 | tee gpurun_out/r06_z12_ab.txt
unset MERLOT_LN_BWD_BLOCKS
timeout 900 python -m pytest tests -q -m gpu -k "layernorm or ln_ or layer_norm or grad_classes or config2" 2>&1 | grep -v Warning | tail -3 | cut -c1-300 | tee gpurun_out/r06_z12_tests.txt
