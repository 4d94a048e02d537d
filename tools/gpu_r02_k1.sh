#!/bin/bash
# session k1: kernel experiments
mkdir -p gpurun_out/r02k1
timeout 600 python -m pytest tests/test_conv_gpu.py -q -x -k "second_generation or f4x4 or head" 2>&1 | tail -3 > gpurun_out/r02k1/tests.txt
cat gpurun_out/r02k1/tests.txt
timeout 300 python tools/conv_sweep.py --wino 2>&1 | grep "wino2\|wino4" > gpurun_out/r02k1/sweep.txt
cat gpurun_out/r02k1/sweep.txt
