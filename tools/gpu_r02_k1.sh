#!/bin/bash
# session k1: first runs of the F(4x4,3x3) kernel: parity tests, then time vs cin (fixed cost per work item vs cost per K-step)
mkdir -p gpurun_out/r02k1
timeout 900 python -m pytest tests/test_conv_gpu.py -q -x -k "f4x4 or (every_tile_config and 11)" 2>&1 | tail -15 > gpurun_out/r02k1/tests.txt
cat gpurun_out/r02k1/tests.txt
timeout 600 python tools/conv_sweep.py --cinsweep --tile 11 2>&1 | grep -v "^$" > gpurun_out/r02k1/cinsweep.txt
timeout 600 python tools/conv_sweep.py --cinsweep --tile 8 2>&1 | grep -v "^$" >> gpurun_out/r02k1/cinsweep.txt
cat gpurun_out/r02k1/cinsweep.txt
