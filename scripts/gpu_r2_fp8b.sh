#!/bin/bash
# fp8 path at the model level: the config-#5 parity test, then bench.py --config 5 with fp8 forward GEMMs and all-bf16
mkdir -p gpurun_out
T=${TAG:-r2fp8b}
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -k config5 2>&1 | tail -25 > gpurun_out/${T}_tests.log
cat gpurun_out/${T}_tests.log
timeout 600 python bench.py --config 5 --steps 4 --warmup 2 > gpurun_out/${T}_bench_fp8.json 2> gpurun_out/${T}_bench_fp8.err
tail -3 gpurun_out/${T}_bench_fp8.err; cat gpurun_out/${T}_bench_fp8.json
timeout 600 python bench.py --config 5 --bf16 --steps 4 --warmup 2 > gpurun_out/${T}_bench_bf16.json 2> gpurun_out/${T}_bench_bf16.err
tail -3 gpurun_out/${T}_bench_bf16.err; cat gpurun_out/${T}_bench_bf16.json
