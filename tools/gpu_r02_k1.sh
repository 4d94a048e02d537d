#!/bin/bash
# session k1: output stores of conv_wino4 with cache-policy bits (plain / sc1 = dropped from L2 / nt)
mkdir -p gpurun_out/r02k1
export TMPDIR=/tmp
for rep in 1 2; do
for v in "" _sc1 _nt; do
  for shape in "64 64 96 96" "128 128 48 48" "256 256 24 24"; do
    W2L_HIP_LIB=$PWD/wav2lip_amd/lib/libw2l_hip$v.so timeout 100 python tools/conv_sweep.py --one $shape --tile 11 2>&1 | grep "one"
  done
done
done > gpurun_out/r02k1/aux.txt
cat gpurun_out/r02k1/aux.txt
ROOT=$PWD
cd /tmp
for v in "" _sc1; do
  W2L_HIP_LIB=$ROOT/wav2lip_amd/lib/libw2l_hip$v.so timeout 200 rocprofv3 --output-format csv --pmc FETCH_SIZE WRITE_SIZE -d $ROOT/gpurun_out/r02k1/pmc$v -o pmc -- python $ROOT/tools/conv_sweep.py --one 64 64 96 96 --tile 11 --reps 2 > /dev/null 2>&1
  python - "$ROOT/gpurun_out/r02k1/pmc$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = {}
for r in rows:
    if "wino4_f32" in r["Kernel_Name"]:
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(sys.argv[1].split("/")[-1], k, "per launch (KiB): min %.0f max %.0f n %d" % (min(v), max(v), len(v)))
PY
done >> $ROOT/gpurun_out/r02k1/aux.txt 2>&1
tail -4 $ROOT/gpurun_out/r02k1/aux.txt
find $ROOT/gpurun_out/r02k1 -name "*.db" -delete
