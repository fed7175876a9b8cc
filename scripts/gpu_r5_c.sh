#!/bin/bash
# round 5, call c: the attention tests on the product build, then the whole training step with the persistent attention kernels off / on
# (experiments build, same box, A B A B)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/scripts:$PWD/tests
OUT=gpurun_out/r05_s_attn_pp_step_ab.txt
: > $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -3 | tee -a $OUT
for pp in 0 1 0 1; do
  echo "MERLOT_ATTN_PP=$pp" >> $OUT
  MERLOT_ATTN_PP=$pp timeout 600 python scripts/bench_exp.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-260 >> $OUT
done
cat $OUT
