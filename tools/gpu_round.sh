#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (+A/B variants), rocprofv3 kernel stats and PMC passes.
# usage (from the repo root, on the GPU box via gpurun):  [PROFILE=1] [SKIP_TESTS=1] [VARIANTS="a b"] bash tools/gpu_round.sh [tag]
# Launch configurations are the committed tune table + heuristic in every run (no autotune launches anywhere), so the bench,
# the A/B variants and the rocprofv3 passes all execute the same kernels with the same configurations.
# AUTOTUNE=1 makes the first bench run stopwatch-tune instead and share its choices through $OUT/tune.json.
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
STEPS=${STEPS:-20}
if [ -z "$SKIP_TESTS" ]; then
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $OUT/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > $OUT/smoke.log
fi
python -c "import bench; print(bench.source_fingerprint())" > $OUT/source_fingerprint.txt
TUNE=""
[ -n "$AUTOTUNE" ] && TUNE="--autotune --tune-cache $OUT/tune.json"
(timeout 500 python bench.py --steps $STEPS --warmup 3 $TUNE --profile-layers $BENCH_ARGS 2>$OUT/layers.log | tail -1) > $OUT/bench.json
for v in $VARIANTS; do
  (W2L_HIP_LIB=$PWD/wav2lip_amd/lib/libw2l_hip_$v.so timeout 300 python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --profile-layers 2>$OUT/layers_$v.log | tail -1) > $OUT/bench_$v.json
done
if [ -n "$PROFILE" ]; then
  PB="python $ROOT/bench.py --no-cpu-baseline $BENCH_ARGS"
  [ -n "$AUTOTUNE" ] && PB="$PB --tune-cache $ROOT/$OUT/tune.json"
  cd /tmp
  (timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_stats -o stats -- $PB --steps $STEPS --warmup 3 --windows 1 > $ROOT/$OUT/prof_stats.log 2>&1)
  # the same kernel statistics with ONE batch in flight: per-kernel average durations are then serial launch durations (with four
  # batches in flight kernels of different streams overlap and the averages are concurrency-inflated)
  (timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_stats_p1 -o stats -- $PB --pipeline 1 --steps $STEPS --warmup 3 --windows 1 --sustained-seconds 0 > $ROOT/$OUT/prof_stats_p1.log 2>&1)
  pmc() {  # name, counters...
    local name=$1; shift
    # --pipeline 1: one batch at a time, so that "the dispatches between two datagen_pack launches" are exactly one step
    # --sustained-seconds 0: the counter CSV holds one row per dispatch and counter (the 2.5 s window would make it 27 000 dispatches)
    (timeout 400 rocprofv3 --output-format csv --pmc "$@" -d $ROOT/$OUT/prof_$name -o $name -- $PB --pipeline 1 --steps 2 --warmup 1 --windows 1 --sustained-seconds 0 > $ROOT/$OUT/prof_$name.log 2>&1)
  }
  pmc pmc1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES
  pmc pmc2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
  pmc pmc3 FETCH_SIZE
  pmc pmc4 WRITE_SIZE
  cd $ROOT
  # keep the merge small: drop everything but csv/txt summaries
  find $OUT -type f \( -name "*.db" -o -name "*.pftrace" \) -delete
  find $OUT -type f -size +8M -delete
fi
tail -3 $OUT/pytest_gpu.log 2>/dev/null; tail -2 $OUT/smoke.log 2>/dev/null; cat $OUT/bench.json; for v in $VARIANTS; do cat $OUT/bench_$v.json; done
du -sh $OUT
