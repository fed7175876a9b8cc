#!/bin/bash
# product build: full GPU suite + smoke + bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r03_k_bench.json 2> gpurun_out/r03_k_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_k_bench.json'))
print(d['value'], d['ms_per_step'], d['model_flops_utilization'], d['roofline']['frac'], d['roofline_wgrad']['frac'], d['forward_only']['ms_per_pass'])
PY
