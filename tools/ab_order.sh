#!/bin/bash
# Same-box A/B of workgroup orders (igemm_block_coords, conv_wino's m_fastest) on a -DW2L_ORDER_ENV build of conv_igemm.hip and
# conv_wino.hip (libw2l_hip_orderenv.so): frames/s, per-layer serial timings and HBM fetch bytes per condition.
# usage (GPU box): [CONDS="name:igemm_order:wino_mfast:phase_blocked_group ..."] bash tools/ab_order.sh <tag>     ("-" = the launcher's own rule)
TAG=${1:-order}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD
CONDS=${CONDS:-"old:0:0:- new:-:-:- nocm:2:-:- r16:-:-:16 r24:-:-:24 r48:-:-:48 r64:-:-:64"}
(timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -2) > $OUT/pytest.log
export W2L_HIP_LIB=$ROOT/wav2lip_amd/lib/libw2l_hip_orderenv.so
B="python $ROOT/bench.py --no-cpu-baseline --no-train-configs --steps 20 --warmup 3 --sustained-seconds 1.5"
setc() { IFS=: read name o m r <<< "$1"; unset W2L_IGEMM_ORDER W2L_WINO_MFAST W2L_IGEMM_R; [ "$r" != "-" ] && export W2L_IGEMM_R=$r; [ "$o" != "-" ] && export W2L_IGEMM_ORDER=$o; [ "$m" != "-" ] && export W2L_WINO_MFAST=$m; }
for i in 1 2 3; do for c in $CONDS; do setc $c
  timeout 200 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name value %.0f sustained %.0f serial %.3f ms' % (d['value'], d['windows']['sustained_value'], d['roofline']['serial_ms_per_step']))"
done; done > $OUT/ab.log
for c in $CONDS; do setc $c
  timeout 200 $B --profile-layers 2>$OUT/layers_$name.log >/dev/null
  (cd /tmp; timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $ROOT/$OUT/pmc3_$name -o pmc3 -- $B --pipeline 1 --steps 2 --warmup 1 --windows 1 --sustained-seconds 0 > $ROOT/$OUT/pmc3_$name.log 2>&1)
  python tools/pmc_summary.py $OUT/pmc3_$name/pmc3_counter_collection.csv > $OUT/fetch_$name.txt
  rm -rf $OUT/pmc3_$name
done
cat $OUT/pytest.log $OUT/ab.log; for f in $OUT/fetch_*.txt; do echo $f; tail -n 1 $f; done
