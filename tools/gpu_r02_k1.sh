#!/bin/bash
# session k1: HIP-graph replay of the generator's launch sequence vs plain launches, default bench
mkdir -p gpurun_out/r02k1
for rep in 1 2; do
for g in 0 1; do
  W2L_HIP_GRAPHS=$g timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hip_graphs $g', d['value'], d['ms_per_step'], d['windows'])"
done
done > gpurun_out/r02k1/graphs.txt 2>&1
cat gpurun_out/r02k1/graphs.txt
