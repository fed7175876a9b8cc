#!/bin/bash
# round 6, closing run of a build on ONE box: the whole GPU suite + smoke, the default bench line (CPU baseline included), the driver's own command line,
# rocprofv3 kernel statistics of the bench command, the TCC traffic passes (hash-stamped; now with the joint / text-only attention and the LayerNorm kernels),
# the SQ counter pass, the stem / native-yaml / config-5 lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TAG=${TAG:-r06_final}
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -6 > gpurun_out/${TAG}_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/${TAG}_pytest_gpu.txt
cat gpurun_out/${TAG}_pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -2 gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_driver_cmd.json 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_bench_driver_cmd.json
bash scripts/gpu_r2_prof.sh > gpurun_out/${TAG}_prof_summary.txt 2>&1; head -3 gpurun_out/${TAG}_prof_summary.txt | cut -c1-200
bash scripts/gpu_traffic.sh > /dev/null 2>&1; grep "HBM_MB\|hash" gpurun_out/r06_traffic.txt
bash scripts/gpu_pmc.sh > gpurun_out/${TAG}_pmc_sq.txt 2>&1; tail -12 gpurun_out/${TAG}_pmc_sq.txt | cut -c1-220
timeout 600 python bench.py --resnet-stem --no-cpu-baseline > gpurun_out/${TAG}_bench_resnet_stem.json 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_bench_resnet_stem.json
timeout 600 python bench.py --native-yaml --no-cpu-baseline > gpurun_out/${TAG}_bench_native_yaml.json 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_bench_native_yaml.json
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5_fp8.json 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_bench_config5_fp8.json
timeout 600 python bench.py --config 5 --bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5_bf16.json 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_bench_config5_bf16.json
