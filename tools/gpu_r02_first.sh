#!/bin/bash
# Round-2 first GPU session: regenerate the tune table, run the GPU suite on the default (table-less -> heuristic) path,
# bench with heuristic configurations and with the fresh table.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(W2L_AUTOTUNE=1 timeout 900 python tools/make_tune_table.py --out $OUT/tune_table.json 2>&1 | tail -30) > $OUT/make_tune_table.log
(W2L_TUNE_TABLE=0 timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > $OUT/pytest_gpu_heuristic.log
(W2L_TUNE_TABLE=0 timeout 400 python bench.py --no-train-configs --profile-layers 2>$OUT/layers_heuristic.log | tail -1) > $OUT/bench_heuristic.json
(W2L_TUNE_TABLE=$PWD/$OUT/tune_table.json timeout 400 python bench.py --no-cpu-baseline --profile-layers 2>$OUT/layers_table.log | tail -1) > $OUT/bench_table.json
(W2L_TUNE_TABLE=$PWD/$OUT/tune_table.json timeout 400 python bench.py --no-cpu-baseline --autotune 2>/dev/null | tail -1) > $OUT/bench_autotune.json
(W2L_TUNE_TABLE=$PWD/$OUT/tune_table.json timeout 600 python -m pytest tests/test_determinism_gpu.py tests/test_train_gpu.py tests/test_models_gpu.py -m gpu -q 2>&1 | tail -15) > $OUT/pytest_gpu_table.log
tail -5 $OUT/make_tune_table.log; tail -12 $OUT/pytest_gpu_heuristic.log; tail -5 $OUT/pytest_gpu_table.log
for f in heuristic table autotune; do cut -c1-400 $OUT/bench_$f.json; done
