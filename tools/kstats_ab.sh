#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of a cfg-4 bf16 training run under two builds of the library.
#   gpurun --timeout 900 -- 'bash tools/kstats_ab.sh <tag> [base_lib] [new_lib]'
TAG=${1:?tag}
A=${2:-wav2lip_amd/lib/libw2l_hip_base.so}
B=${3:-wav2lip_amd/lib/libw2l_hip.so}
ROOT=$PWD
export OUT=$ROOT/gpurun_out/$TAG TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
for which in A B; do
  lib=$A; [ $which = B ] && lib=$B
  W2L_HIP_LIB=$ROOT/$lib timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$which -o s -- python $ROOT/tools/train_bench.py --cfg ${CFGS:-4} --precision bf16 --steps 6 --warmup 2 > $OUT/$which.log 2>&1
  cp $(find $OUT/$which -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$which.csv
  rm -rf $OUT/$which
done
cd $ROOT
python - <<PY
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"].split("(")[0][:70]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3)
    return d
a, b = load("$OUT/kernel_stats_A.csv"), load("$OUT/kernel_stats_B.csv")
print("%-72s %6s %9s %9s | %6s %9s %9s" % ("kernel", "calls", "ms", "avg us", "calls", "ms", "avg us"))
for k in sorted(set(a) | set(b), key=lambda k: -max(a.get(k, (0, 0, 0))[1], b.get(k, (0, 0, 0))[1]))[:45]:
    x, y = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    print("%-72s %6d %9.3f %9.1f | %6d %9.3f %9.1f" % (k, x[0], x[1], x[2], y[0], y[1], y[2]))
print("total A %.2f ms  B %.2f ms" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
PY
