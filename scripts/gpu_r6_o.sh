#!/bin/bash
# round 6, call 16: fc1's input gradient on du's 8-bit copy ('dgrad1'); tests; config-#5 step A/B with the all-bf16 line on the same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_f8_tn_gpu.py -q -m gpu -s -k "f8_nt_with or fused_fp8_backward" 2>&1 | grep -v Warning | grep "passed\|failed\|Error\|error\|assert\|fp8_backward\|FAILED" | cut -c1-500 | tee gpurun_out/r06_o_f8_tests.txt
for mode in bf16 none w1,w2,fuse,noa w1,w2,fuse,noa,dgrad1 w1,w2,wqkv,fuse,noa,dgrad1 bf16 none w1,w2,fuse,noa,dgrad1; do
  if [ $mode = none ]; then extra=""; elif [ $mode = bf16 ]; then extra="--bf16"; else extra="--fp8-bwd $mode"; fi
  timeout 600 python bench.py --config 5 --no-cpu-baseline --steps 8 --warmup 4 $extra 2>/dev/null | tee -a gpurun_out/r06_o_bench5_$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value'],1), 'seg/s', round(d['ms_per_step'],1), 'ms', 'loss', d['config'].get('final_loss'))"
done
