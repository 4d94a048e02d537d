#!/bin/bash
# First GPU session of tools/experiments/conv_wino4w.hip (an A/B against the production F(4x4) kernel; see EXPERIMENTS.md).
# Build the variants HERE (hipcc cross-compiles; the .so files travel with the snapshot), then run this on the GPU box:
#   bash tools/build_variant.sh w4w  ../../tools/experiments/conv_wino4w.hip "-DW4W_RING=3" conv_wino4
#   bash tools/build_variant.sh w4w6 ../../tools/experiments/conv_wino4w.hip "-DW4W_RING=6" conv_wino4
#   bash tools/build_variant.sh w4x  ../../tools/experiments/conv_wino4w.hip "-DW4W_HALF=1" conv_wino4     (two 16-tile workgroups per CU)
#   gpurun --timeout 600 -- 'bash tools/gpu_session.sh w4w "bash tools/experiments/w4w_session.sh"'
# Order: parity first (every wino4 test of the conv suite through the variant library), then per-layer times of configuration
# 11 on the decoder shapes for production / ring 3 / ring 6, then the bench step with the faster variant.
set -u
V3=wav2lip_amd/lib/libw2l_hip_w4w.so
V6=wav2lip_amd/lib/libw2l_hip_w4w6.so
VX=wav2lip_amd/lib/libw2l_hip_w4x.so
for V in $V3 $V6 $VX; do
    [ -f $V ] || { echo "missing $V: build the variants first"; exit 1; }
    echo "== parity through $V"
    W2L_HIP_LIB=$V timeout 200 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "wino4" 2>&1 | tail -3
done
echo "== layer times, configuration 11 (F(4x4)): production, ring 3, ring 6, two half workgroups"
timeout 120 python tools/conv_sweep.py --wino --only-tile 11 2>&1 | grep wino
W2L_HIP_LIB=$V3 timeout 120 python tools/conv_sweep.py --wino --only-tile 11 2>&1 | grep wino
W2L_HIP_LIB=$V6 timeout 120 python tools/conv_sweep.py --wino --only-tile 11 2>&1 | grep wino
W2L_HIP_LIB=$VX timeout 120 python tools/conv_sweep.py --wino --only-tile 11 2>&1 | grep wino
echo "== bench step: production, ring 3"
timeout 200 python bench.py --no-cpu-baseline --no-train-configs --steps 20 --windows 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('production', d['value'], d['roofline']['frac'], d['roofline']['dominant_kernel']['frac'])"
W2L_HIP_LIB=$V3 timeout 200 python bench.py --no-cpu-baseline --no-train-configs --steps 20 --windows 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w4w ring 3', d['value'], d['roofline']['frac'], d['roofline']['dominant_kernel']['frac'], d.get('parity'))"
