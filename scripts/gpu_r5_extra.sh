#!/bin/bash
# round 5: what the closing run did not cover on the final build -- BASELINE config #5 (all-bf16 and fp8 forward GEMMs, same box) and the training sanity run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${TAG:-r05_y}
timeout 600 python bench.py --config 5 --bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5_bf16.json 2>/dev/null; cut -c1-260 gpurun_out/${TAG}_bench_config5_bf16.json
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5_fp8.json 2>/dev/null; cut -c1-260 gpurun_out/${TAG}_bench_config5_fp8.json
timeout 400 python scripts/train_sanity.py > gpurun_out/${TAG}_train_sanity.txt 2>&1; tail -3 gpurun_out/${TAG}_train_sanity.txt
