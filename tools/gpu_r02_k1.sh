#!/bin/bash
# session k1: experiments - batches in flight
mkdir -p gpurun_out/r02k1
for p in 2 3 4 2 3; do
  timeout 200 python bench.py --pipeline $p --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline $p', d['value'], d['ms_per_step'], d['windows'])"
done > gpurun_out/r02k1/pipeline.txt 2>&1
cat gpurun_out/r02k1/pipeline.txt
