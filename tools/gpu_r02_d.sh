#!/bin/bash
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for m in 0 1 2; do
  (W2L_WINO2_STAGGER=$m timeout 300 python tools/conv_sweep.py --wino --only-tile 8 2>&1 | grep wino | sed "s/^/stagger$m /") >> $OUT/wino2_stagger.txt
done
(W2L_WINO2_STAGGER=1 timeout 300 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "second_generation" 2>&1 | tail -3) > $OUT/pytest_wino2.log
cat $OUT/wino2_stagger.txt; tail -3 $OUT/pytest_wino2.log
