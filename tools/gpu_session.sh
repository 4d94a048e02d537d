#!/bin/bash
# One parametrised GPU-box session (replaces the per-session scratch scripts of round 2).
#   gpurun --timeout 900 -- 'bash tools/gpu_session.sh <tag> "<shell snippet>"'
# The snippet runs from the repo root with $OUT = gpurun_out/<tag> (merged back by gpurun), TMPDIR=/tmp, and its whole
# stdout/stderr kept in $OUT/session.log; the last lines are echoed so that gpurun's tail shows them.
TAG=${1:?tag}
shift
export OUT=gpurun_out/$TAG TMPDIR=/tmp ROOT=$PWD
mkdir -p $OUT
( eval "$@" ) > $OUT/session.log 2>&1
echo "exit $?" >> $OUT/session.log
find $OUT -type f \( -name "*.db" -o -name "*.pftrace" \) -delete
find $OUT -type f -size +8M -delete
tail -${TAIL:-40} $OUT/session.log | cut -c1-${COLS:-240}
