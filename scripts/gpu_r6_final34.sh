#!/bin/bash
# round 6, closing run of the final tree (final3: ABI v10 build; final4: ABI v11 -- GroupNorm over Infinity-Cache-sized groups of samples, the fused attention backward's chunked arrival from 320 tokens up) on ONE box:
# the whole GPU suite + smoke, the default bench line (CPU baseline included), the driver's own command line, rocprofv3 kernel statistics of the bench command, the TCC traffic
# passes (hash-stamped for the current sources), the SQ counter pass, the stem / native-yaml lines with the native-yaml kernel statistics, config #5 (all-bf16 and the default)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TAG=${TAG:-r06_final7}
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | grep "passed\|failed\|FAILED\|Error" | head -5 > gpurun_out/${TAG}_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/${TAG}_pytest_gpu.txt
cat gpurun_out/${TAG}_pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -2 gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_driver_cmd.json 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_bench_driver_cmd.json
bash scripts/gpu_r2_prof.sh > gpurun_out/${TAG}_prof_summary.txt 2>&1; head -3 gpurun_out/${TAG}_prof_summary.txt | cut -c1-200
bash scripts/gpu_traffic.sh > /dev/null 2>&1; grep "HBM_MB\|hash" gpurun_out/r06_traffic.txt | head -12
bash scripts/gpu_pmc.sh > gpurun_out/${TAG}_pmc_sq.txt 2>&1; tail -6 gpurun_out/${TAG}_pmc_sq.txt | cut -c1-220
timeout 600 python bench.py --resnet-stem --no-cpu-baseline > gpurun_out/${TAG}_bench_resnet_stem.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_resnet_stem.json
timeout 600 python bench.py --native-yaml --no-cpu-baseline > gpurun_out/${TAG}_bench_native_yaml.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_native_yaml.json
timeout 600 python bench.py --config 5 --bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5_bf16.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_config5_bf16.json
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5_fp8.json 2>/dev/null; cut -c1-260 gpurun_out/${TAG}_bench_config5_fp8.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/profn
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profn -o bench -- python $GRAFT_REPO_ROOT/bench.py --native-yaml --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_profn.log 2>&1
cp $(find /tmp/profn -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_native_yaml_kernel_stats.csv 2>/dev/null
cd $GRAFT_REPO_ROOT
