#!/bin/bash
# same-box comparison of the round-5 tree (_r5/, commit 9c9634a) and the current tree on the bf16 training steps and the bench
export TMPDIR=/tmp
mkdir -p gpurun_out/r7_rounds
for r in 1 2 3; do
  for which in r5 r6; do
    d=.; [ $which = r5 ] && d=_r5
    (cd $d && timeout -s KILL 300 python tools/train_bench.py --cfg 3 4 5 --precision bf16 --steps 10 --warmup 3 2>/dev/null) | python -c "
import sys, re
for l in sys.stdin:
    m = re.search(r'\"cfg\": (\d).*?\"ms_per_step\": ([\d.]+)', l)
    if m: print('$which', 'cfg', m.group(1), m.group(2))
" | tee -a gpurun_out/r7_rounds/ab.log
  done
done
for which in r5 r6; do
  d=.; [ $which = r5 ] && d=_r5
  (cd $d && timeout -s KILL 300 python bench.py --no-cpu-baseline --no-train-configs --windows 5 2>/dev/null) | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d = json.loads(l); print('$which', 'bench', d['value'], d['windows']['sustained_value'])
" | tee -a gpurun_out/r7_rounds/ab.log
done
