#!/bin/bash
# session k1: encoders on two streams vs one, with four batches in flight (default bench)
mkdir -p gpurun_out/r02k1
for rep in 1 2; do
for g in 1 0; do
  W2L_TWO_STREAMS=$g timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_streams $g', d['value'], d['ms_per_step'], d['windows'])"
done
done > gpurun_out/r02k1/streams.txt 2>&1
for p in 6 8; do
  timeout 200 python bench.py --pipeline $p --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline $p', d['value'], d['ms_per_step'], d['windows'])"
done >> gpurun_out/r02k1/streams.txt 2>&1
cat gpurun_out/r02k1/streams.txt
