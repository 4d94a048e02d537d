timeout 300 python -m pytest tests/test_bf16_conv_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "bf16" 2>&1 | tail -3
for i in 1 2; do
W2L_CONVB_BOX=0 timeout 300 python tools/train_bench.py --precision bf16 --cfg 3 4 5 --steps 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('gemm', d['cfg'], d['ms_per_step'], d['frac'])"
timeout 300 python tools/train_bench.py --precision bf16 --cfg 3 4 5 --steps 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('box ', d['cfg'], d['ms_per_step'], d['frac'])"
done
for L in "res64 @96"; do
W2L_CONVB_BOX=0 timeout 200 python tools/bf16_sweep.py --fwd --only "$L" 2>&1 | grep -v amdgpu | grep "res64" | cut -c1-60 | sed 's/^/GEMM /'
timeout 200 python tools/bf16_sweep.py --fwd --only "$L" 2>&1 | grep -v amdgpu | grep "res64" | cut -c1-60 | sed 's/^/BOX  /'
done
