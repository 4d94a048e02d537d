#!/bin/bash
# session k1: quarter-split conv_wino2 shape: parity, then the 32-cout layers against the other configurations
mkdir -p gpurun_out/r02k1
timeout 600 python -m pytest tests/test_conv_gpu.py -q -x -k "quarter or head or (every_tile_config and 12)" 2>&1 | tail -5 > gpurun_out/r02k1/tests.txt
cat gpurun_out/r02k1/tests.txt
for rep in 1 2; do
for t in 4 9 12; do
  timeout 100 python tools/conv_sweep.py --one 80 32 96 96 --tile $t 2>&1 | grep "one"
  timeout 100 python tools/conv_sweep.py --one 32 32 48 48 --tile $t 2>&1 | grep "one"
  timeout 100 python tools/conv_sweep.py --one 32 32 80 16 --tile $t 2>&1 | grep "one"
done
for t in 8 12; do
  timeout 100 python tools/conv_sweep.py --one 64 64 24 24 --tile $t 2>&1 | grep "one"
  timeout 100 python tools/conv_sweep.py --one 64 64 96 96 --tile $t 2>&1 | grep "one"
done
done > gpurun_out/r02k1/q.txt
cat gpurun_out/r02k1/q.txt
