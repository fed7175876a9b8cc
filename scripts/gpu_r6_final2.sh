#!/bin/bash
# round 6, SECOND closing run (the build with the 8-bit backward, ABI v9) on ONE box: the whole GPU suite + smoke, the default bench line (CPU baseline included), the driver's
# own command line, rocprofv3 kernel statistics of the bench command, the TCC traffic passes (hash-stamped), the SQ counter pass, the stem / native-yaml lines, config #5
# (all-bf16, fp8 forward only, the default with the 8-bit backward), the headline geometry on the 8-bit paths, 200 training steps on one batch with and without the 8-bit backward
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TAG=${TAG:-r06_final2}
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
timeout 600 python bench.py --config 5 --fp8-bwd none --no-cpu-baseline > gpurun_out/${TAG}_bench_config5_fp8fwd.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_config5_fp8fwd.json
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5_fp8.json 2>/dev/null; cut -c1-260 gpurun_out/${TAG}_bench_config5_fp8.json
timeout 600 python bench.py --fp8 --fp8-bwd w1,w2,wqkv,fuse,noa,dgrad1,dgradqkv --no-cpu-baseline > gpurun_out/${TAG}_bench_c2geom_8bit.json 2>/dev/null; cut -c1-260 gpurun_out/${TAG}_bench_c2geom_8bit.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 4 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof5.log 2>&1
cp $(find /tmp/prof5 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_config5_kernel_stats.csv 2>/dev/null
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/train_sanity.py 0.1 w1,w2,wqkv,fuse,noa,dgrad1,dgradqkv 8 200 2>&1 | grep -v "Warning\|amdgpu" | tail -16 > gpurun_out/${TAG}_train200_8bit.txt; tail -3 gpurun_out/${TAG}_train200_8bit.txt
timeout 900 python scripts/train_sanity.py 0.1 "" 8 200 2>&1 | grep -v "Warning\|amdgpu" | tail -16 > gpurun_out/${TAG}_train200_fp8fwd.txt; tail -3 gpurun_out/${TAG}_train200_fp8fwd.txt
timeout 900 python scripts/train_sanity.py 0.1 bf16 8 200 2>&1 | grep -v "Warning\|amdgpu" | tail -16 > gpurun_out/${TAG}_train200_bf16.txt; tail -3 gpurun_out/${TAG}_train200_bf16.txt
