#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
