#!/bin/bash
# regenerate the tune table with the new configurations, then bench with it
TAG=${1:-r02g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(W2L_AUTOTUNE=1 timeout 900 python tools/make_tune_table.py --out $OUT/tune_table.json 2>&1 | tail -30) > $OUT/make_tune_table.log
(W2L_TUNE_TABLE=$PWD/$OUT/tune_table.json timeout 400 python bench.py --no-cpu-baseline --profile-layers 2>$OUT/layers.log | tail -1) > $OUT/bench.json
(W2L_TUNE_TABLE=$PWD/$OUT/tune_table.json timeout 400 python bench.py --no-cpu-baseline --pipeline 3 2>/dev/null | tail -1) > $OUT/bench_p3.json
(W2L_TUNE_TABLE=$PWD/$OUT/tune_table.json timeout 400 python bench.py --no-cpu-baseline --pipeline 1 2>/dev/null | tail -1) > $OUT/bench_p1.json
tail -3 $OUT/make_tune_table.log; for f in bench bench_p3 bench_p1; do cut -c1-330 $OUT/$f.json; done; grep -v amdgpu $OUT/layers.log | cut -c1-130
