timeout 300 python -m pytest tests/test_bf16_conv_gpu.py -m gpu -q -x -k "every_tile" 2>&1 | tail -2
for L in "res128 @48" "res256 @24" "res384 @12" "res512 @6" "convT 512->256 @12" "convT 320->128 @24" "res64 @96"; do
timeout 200 python tools/bf16_sweep.py --fwd --tiles --only "$L" 2>&1 | grep -v amdgpu | grep -v totals | cut -c1-250
done
