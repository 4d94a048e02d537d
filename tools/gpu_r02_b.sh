#!/bin/bash
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60) > $OUT/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > $OUT/smoke.log
(timeout 500 python bench.py --no-train-configs 2>/dev/null | tail -1) > $OUT/bench.json
tail -40 $OUT/pytest_gpu.log | cut -c1-300; tail -2 $OUT/smoke.log; cut -c1-300 $OUT/bench.json
