#!/bin/bash
# round 6, call 20: the QKV input gradient on dqkv's 8-bit copy ('dgradqkv': one quantising pass, two consumers), weights quantised once per step; the fp8 attention forward removed
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_f8_tn_gpu.py tests/test_fp8_gpu.py -q -m gpu -s -k "fused_fp8_backward or test_fp8_gpu" 2>&1 | grep -v Warning | grep "passed\|failed\|Error\|error\|assert\|fp8_backward\|FAILED" | cut -c1-500 | tee gpurun_out/r06_s_f8_tests.txt
for mode in none w1,w2,fuse,noa,dgrad1 w1,w2,wqkv,fuse,noa,dgrad1,dgradqkv bf16 w1,w2,fuse,noa,dgrad1 w1,w2,wqkv,fuse,noa,dgrad1,dgradqkv; do
  if [ $mode = bf16 ]; then extra="--bf16"; else extra="--fp8-bwd $mode"; fi
  timeout 600 python bench.py --config 5 --no-cpu-baseline --steps 8 --warmup 4 $extra 2>/dev/null | tee -a gpurun_out/r06_s_bench5_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done
timeout 600 python bench.py --no-cpu-baseline --fp8 --fp8-bwd w1,w2,wqkv,fuse,noa,dgrad1,dgradqkv 2>/dev/null | tee gpurun_out/r06_s_bench_c2_fp8all.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config-2 geometry, all 8-bit paths', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms')"
timeout 600 python bench.py --no-cpu-baseline --fp8 --fp8-bwd w1,w2,fuse,noa,dgrad1 2>/dev/null | tee gpurun_out/r06_s_bench_c2_fp8mlp.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config-2 geometry, MLP 8-bit paths', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms')"
