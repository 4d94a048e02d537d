#!/bin/bash
# One GPU-box session for the training path: parity tests, step timings for BASELINE configs 3-5, rocprofv3 kernel stats.
# usage (repo root, via gpurun):  [SKIP_TESTS=1] [CFGS="3 4 5"] [PROFILE=1] bash tools/gpu_train_round.sh [tag]
TAG=${1:-t01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
CFGS=${CFGS:-"3 4 5"}
if [ -z "$SKIP_TESTS" ]; then
(timeout -s KILL 600 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider ${PYTEST_ARGS} 2>&1 | tail -40) > $OUT/pytest_gpu.log
fi
(timeout -s KILL 500 python tools/train_bench.py --cfg $CFGS --steps ${STEPS:-5} --warmup 2 --profile-nodes $BENCH_ARGS 2>$OUT/train_nodes.log | tail -12) > $OUT/train_bench.log
if [ -n "$PROFILE" ]; then
  cd /tmp
  for c in $CFGS; do
    (timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_cfg$c -o stats -- python $ROOT/tools/train_bench.py --cfg $c --steps 3 --warmup 1 $BENCH_ARGS > $ROOT/$OUT/prof_cfg$c.log 2>&1)
  done
  cd $ROOT
  find $OUT -type f \( -name "*.db" -o -name "*.pftrace" -o -name "*kernel_trace.csv" \) -delete
  find $OUT -type f -size +8M -delete
fi
tail -8 $OUT/pytest_gpu.log 2>/dev/null; cat $OUT/train_bench.log
du -sh $OUT
