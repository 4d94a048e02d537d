#!/bin/bash
# session k1: bench vs batch size (tune-table shapes: 1 8 16 32 64 128 256)
mkdir -p gpurun_out/r02k1
for b in 1 8 16 32 64 128 256 512; do
  timeout 200 python bench.py --batch $b --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('batch $b', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', 'executed frac', r['frac'], 'nominal TF', r['nominal_tflops'], d['config']['launch_configs'][:40])"
done > gpurun_out/r02k1/batch.txt 2>&1
cat gpurun_out/r02k1/batch.txt
