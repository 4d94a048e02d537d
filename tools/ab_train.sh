#!/bin/bash
# Same-box A/B of two builds of libw2l_hip on the bf16 training steps: alternates W2L_HIP_LIB between the two libraries REPS times.
#   gpurun --timeout 900 -- 'bash tools/ab_train.sh <tag> [base_lib] [new_lib]'      (CFGS="3 4 5" REPS=3 STEPS=10)
# ENVA / ENVB: extra environment of the two sides ("W2L_BWD_PRUNE=0 ..."), for switches that live in the Python layer.
TAG=${1:?tag}
A=${2:-wav2lip_amd/lib/libw2l_hip_base.so}
B=${3:-wav2lip_amd/lib/libw2l_hip.so}
export OUT=gpurun_out/$TAG TMPDIR=/tmp
mkdir -p $OUT
for r in $(seq 1 ${REPS:-3}); do
  for which in A B; do
    lib=$A; [ $which = B ] && lib=$B
    extra=$ENVA; [ $which = B ] && extra=$ENVB
    env $extra W2L_HIP_LIB=$PWD/$lib timeout -s KILL 300 python tools/train_bench.py --cfg ${CFGS:-3 4 5} --precision ${PREC:-bf16} --steps ${STEPS:-10} --warmup 3 2>/dev/null \
      | python -c "
import sys, re
for l in sys.stdin:
    m = re.search(r'\"cfg\": (\d).*?\"ms_per_step\": ([\d.]+)', l)
    if m: print('$which', '$(basename $lib)', 'cfg', m.group(1), m.group(2))
" | tee -a $OUT/ab.log
  done
done
python - <<PY
import collections
d = collections.defaultdict(list)
for l in open("$OUT/ab.log"):
    w, lib, _, cfg, ms = l.split()
    d[(cfg, w)].append(float(ms))
for cfg in sorted({k[0] for k in d}):
    a, b = d[(cfg, "A")], d[(cfg, "B")]
    ma, mb = sum(a) / len(a), sum(b) / len(b)
    print("cfg %s: A %.3f  B %.3f  (%+.2f %%)" % (cfg, ma, mb, 100 * (mb - ma) / ma))
PY
