#!/bin/bash
# round 5, the very last call: the whole GPU suite + smoke + the default bench line on the final tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | grep "passed\|failed\|Error\|^E " | tail -6 > gpurun_out/r05_end_pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r05_end_pytest_gpu.txt
cat gpurun_out/r05_end_pytest_gpu.txt
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r05_end_bench.json 2>/dev/null; cut -c1-330 gpurun_out/r05_end_bench.json
