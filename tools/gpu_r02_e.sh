#!/bin/bash
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for g in 256 512 768; do
  (W2L_DEBUG=1 W2L_WINO2_STAGGER=0 W2L_WINO2_GRID=$g timeout 300 python tools/conv_sweep.py --wino --only-tile 8 2>&1 | grep "wino\|w2l" | sed "s/^/grid$g /") >> $OUT/wino2_grid.txt
done
(W2L_WINO2_STAGGER=1 W2L_WINO2_GRID=512 timeout 300 python tools/conv_sweep.py --wino --only-tile 8 2>&1 | grep "wino" | sed "s/^/grid512s1 /") >> $OUT/wino2_grid.txt
cat $OUT/wino2_grid.txt
