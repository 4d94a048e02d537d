for i in 1 2; do
timeout 200 python tools/bf16_sweep.py --fwd 2>&1 | grep -v amdgpu | cut -c1-60
W2L_HIP_LIB=wav2lip_amd/lib/libw2l_hip_spread.so timeout 200 python tools/bf16_sweep.py --fwd 2>&1 | grep -v amdgpu | cut -c1-60 | sed 's/^/SPREAD /'
done
W2L_HIP_LIB=wav2lip_amd/lib/libw2l_hip_spread.so timeout 300 python -m pytest tests/test_bf16_conv_gpu.py -m gpu -q -x 2>&1 | tail -2
