#!/bin/bash
# session k1: experiments - wino4 tile-block shapes on the 24x24 / 12x12 layers
mkdir -p gpurun_out/r02k1
for rep in 1 2; do
for blk in "" "2,2,6" "2,3,4" "3,2,4" "2,4,3" "1,4,6"; do
  W2L_WINO4_BLOCK=$blk timeout 100 python tools/conv_sweep.py --one 256 256 24 24 --tile 11 2>&1 | grep "one" | sed "s/^/blk=$blk /"
done
done > gpurun_out/r02k1/blocks.txt
for blk in "" "2,2,6" "2,3,4" "1,2,10" "1,1,15"; do
  W2L_WINO4_BLOCK=$blk timeout 100 python tools/conv_sweep.py --one 384 384 12 12 --tile 11 2>&1 | grep "one" | sed "s/^/blk=$blk /"
done >> gpurun_out/r02k1/blocks.txt
cat gpurun_out/r02k1/blocks.txt
