#!/bin/bash
# Same-box comparison of an EARLIER tree and the current tree on the bf16 training steps and the bench.  Prepare the earlier tree in
# _r5/ first (it travels to the GPU box with the snapshot; delete it afterwards):
#   mkdir -p _r5 && git archive <commit> | tar -x -C _r5 && make -C _r5/wav2lip_amd/csrc -j4
#   gpurun --timeout 1800 -- 'bash tools/ab_rounds.sh'
# (round 6 used commit 9c9634a, the end of round 5: profiles/r06/m_same_box_round5_tree_against_round6_tree.log)
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
