#!/bin/bash
# round 6, call 58: soak of the product default (one-launch GroupNorm forward) inside the training loop: 80 timed steps of the as-shipped model and of the hybrid-stem line, per-step times must stay flat
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --native-yaml --no-cpu-baseline --no-kernel-timing --steps 80 --warmup 3 2>/dev/null | tee gpurun_out/r06_z17_soak_native.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('as-shipped, 80 steps:', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
timeout 600 python bench.py --resnet-stem --no-cpu-baseline --no-kernel-timing --steps 80 --warmup 3 2>/dev/null | tee gpurun_out/r06_z17_soak_stem.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hybrid stem, 80 steps:', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
