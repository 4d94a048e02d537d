timeout 300 python -m pytest tests/test_bf16_conv_gpu.py -m gpu -q -k "box or statistics" 2>&1 | tail -15
