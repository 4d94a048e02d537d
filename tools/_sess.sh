timeout 300 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "f4x4 or wino" 2>&1 | tail -2
for i in 1 2; do
W2L_HIP_LIB=wav2lip_amd/lib/libw2l_hip_nochain.so timeout 120 python tools/conv_sweep.py --wino --only-tile 11 2>&1 | grep "dec[456]"
timeout 120 python tools/conv_sweep.py --wino --only-tile 11 2>&1 | grep "dec[456]"
done
timeout 300 python bench.py --no-cpu-baseline --no-train-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain', d['value'], d['windows'], d['roofline']['frac'], d['roofline']['dominant_kernel']['frac'], d['roofline']['sustained_clock_mhz'])"
W2L_HIP_LIB=wav2lip_amd/lib/libw2l_hip_nochain.so timeout 300 python bench.py --no-cpu-baseline --no-train-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nochain', d['value'], d['windows'], d['roofline']['frac'], d['roofline']['dominant_kernel']['frac'], d['roofline']['sustained_clock_mhz'])"
