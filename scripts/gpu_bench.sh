#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --examples ${EX:-16} --steps ${STEPS:-8} --warmup 3 ${EXTRA} > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log | cut -c1-3000
