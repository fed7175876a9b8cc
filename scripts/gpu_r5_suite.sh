#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TAG=${TAG:-r05_n}
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -15 > gpurun_out/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/${TAG}_pytest_gpu.txt
cat gpurun_out/${TAG}_pytest_gpu.txt
