#!/bin/bash
# fp8 path at the model level: parity tests, then bench.py --config 5: fp8 (QKV + fc1), + fc2, + attention, all-bf16 -- same box
mkdir -p gpurun_out
T=${TAG:-r2fp8b}
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/${T}_tests.log
cat gpurun_out/${T}_tests.log
python scripts/exp_fp8_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_attn.log
for mode in "" "--fp8-fc2" "--fp8-attn" "--bf16" ""; do
  n=$(echo "x$mode" | tr -d ' -')
  timeout 600 python bench.py --config 5 $mode --steps 4 --warmup 2 --no-kernel-timing > gpurun_out/${T}_bench_$n.json 2> gpurun_out/${T}_bench_$n.err
  tail -2 gpurun_out/${T}_bench_$n.err | grep -v amdgpu.ids; python -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench_$n.json').read().strip().splitlines()[-1]); print('$mode', d['metric'], round(d['value'],1), 'seg/s', round(d['ms_per_step'],2), 'ms', d['config']['final_loss'])"
done
