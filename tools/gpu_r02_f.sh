#!/bin/bash
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "second_generation or every_tile_config or autotuned or fused_1x1_head" 2>&1 | tail -25) > $OUT/pytest_wino2.log
(timeout 600 python tools/conv_sweep.py --wino 2>&1 | grep wino) > $OUT/wino_sweep.txt
tail -12 $OUT/pytest_wino2.log | cut -c1-250; cat $OUT/wino_sweep.txt
