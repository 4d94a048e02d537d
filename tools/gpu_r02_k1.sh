#!/bin/bash
# session k1: kernel experiments - A/B on one box
mkdir -p gpurun_out/r02k1
timeout 300 python -m pytest tests/test_conv_gpu.py -q -x -k "f4x4" 2>&1 | tail -3 > gpurun_out/r02k1/tests.txt
cat gpurun_out/r02k1/tests.txt
for rep in 1 2; do
for v in "" _old; do
  for shape in "64 64 96 96" "128 128 48 48" "256 256 24 24"; do
    W2L_HIP_LIB=$PWD/wav2lip_amd/lib/libw2l_hip$v.so timeout 100 python tools/conv_sweep.py --one $shape --tile 11 2>&1 | grep "one"
  done
done
done > gpurun_out/r02k1/variants.txt
cat gpurun_out/r02k1/variants.txt
