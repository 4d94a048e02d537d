#!/bin/bash
TAG=${1:-r02i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "fused_phase or every_tile_config or autotuned" 2>&1 | tail -25) > $OUT/pytest_tp2.log
(timeout 600 python tools/conv_sweep.py --convt 2>&1 | grep convt) > $OUT/convt_sweep.txt
tail -12 $OUT/pytest_tp2.log | cut -c1-250; cat $OUT/convt_sweep.txt
