#!/bin/bash
# bench.py over (HIP graphs on/off) x (batch sizes), one batch at a time: usage bash tools/bench_matrix.sh "1 8 128"
for hg in 0 1; do
  for b in ${1:-1 8 128}; do
    W2L_HIP_GRAPHS=$hg timeout 200 python bench.py --batch $b --steps 50 --no-cpu-baseline --pipeline 1 2>/dev/null | tail -1 > /tmp/bm.json
    python -c "
import json
try:
    r = json.load(open('/tmp/bm.json')); print('graphs=$hg batch=$b frames/s', r['value'], 'ms/step', r['ms_per_step'])
except Exception as e:
    print('graphs=$hg batch=$b FAILED', e)
"
  done
done
