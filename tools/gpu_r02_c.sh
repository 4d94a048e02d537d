#!/bin/bash
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "second_generation or every_tile_config or autotuned" 2>&1 | tail -25) > $OUT/pytest_wino2.log
(timeout 600 python -m pytest tests/test_golden_datapath_gpu.py tests/test_train_gpu.py -m gpu -q -k "training_loops or hq_step_with_one" 2>&1 | tail -25) > $OUT/pytest_fix.log
(timeout 600 python tools/conv_sweep.py --wino 2>&1 | grep wino) > $OUT/wino_sweep.txt
(timeout 300 python tools/conv_sweep.py --cinsweep --tile 8 2>&1 | tail -8) > $OUT/cinsweep.txt
tail -12 $OUT/pytest_wino2.log | cut -c1-250; tail -6 $OUT/pytest_fix.log | cut -c1-250; cat $OUT/wino_sweep.txt
