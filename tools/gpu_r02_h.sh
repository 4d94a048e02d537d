#!/bin/bash
# Full evidence session: regenerate the tune table IN PLACE (the copy that comes back is the one to commit), then the
# inference round (tests, smoke, bench, rocprofv3 stats + PMC passes) and the training round.
TAG=${1:-r02h}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
(W2L_AUTOTUNE=1 timeout 900 python tools/make_tune_table.py --out wav2lip_amd/tune_table.json 2>&1 | tail -30) > gpurun_out/$TAG/make_tune_table.log
cp wav2lip_amd/tune_table.json gpurun_out/$TAG/tune_table.json
bash tools/gpu_final_round.sh $TAG
