#!/bin/bash
# round 4, closing check of the final build (ABI v6: implicit convolutions, GroupNorm rewrite): the whole GPU suite, smoke, the default bench line (with the CPU
# baseline and the new `hbm` fields), the stem bench line + its kernel statistics.  The headline's kernels are those of profiles/r04_x_* (GEMM source hash unchanged).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_z_pytest_gpu.txt; cat gpurun_out/r04_z_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r04_z_bench.json 2>gpurun_out/r04_z_bench.err; tail -2 gpurun_out/r04_z_bench.err; cut -c1-600 gpurun_out/r04_z_bench.json
timeout 600 python bench.py --resnet-stem --no-cpu-baseline > gpurun_out/r04_y_bench_resnet_stem.json 2>/dev/null; cut -c1-300 gpurun_out/r04_y_bench_resnet_stem.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/prof_stem
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stem -o bench -- python $R/bench.py --resnet-stem --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cp $(find /tmp/prof_stem -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r04_y_stem_kernel_stats.csv 2>/dev/null; head -4 $R/gpurun_out/r04_y_stem_kernel_stats.csv | cut -c1-150
