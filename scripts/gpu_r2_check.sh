#!/bin/bash
# full GPU check: tests, smoke, bench line
mkdir -p gpurun_out
T=${TAG:-r2}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${T}_pytest_gpu.log
cat gpurun_out/${T}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err
tail -2 gpurun_out/${T}_bench.err; cat gpurun_out/${T}_bench.log
