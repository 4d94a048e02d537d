#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (+A/B variants), rocprofv3 kernel stats and PMC passes.
# usage (from the repo root, on the GPU box via gpurun):  bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $OUT/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > $OUT/smoke.log
fi
(timeout 400 python bench.py --steps 20 --warmup 3 --profile-layers $BENCH_ARGS 2>$OUT/layers.log | tail -1) > $OUT/bench.json
for v in $VARIANTS; do
  (W2L_HIP_LIB=$PWD/wav2lip_amd/lib/libw2l_hip_$v.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-layers 2>$OUT/layers_$v.log | tail -1) > $OUT/bench_$v.json
done
if [ -n "$PROFILE" ]; then
  cd /tmp
  (timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_stats -o stats -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/prof_stats.log 2>&1)
  (timeout 200 rocprofv3 -L > $OLDPWD/$OUT/counters_list.txt 2>&1)
  (timeout 400 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES -d $OLDPWD/$OUT/prof_pmc1 -o pmc1 -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_pmc1.log 2>&1)
  (timeout 400 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OLDPWD/$OUT/prof_pmc2 -o pmc2 -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_pmc2.log 2>&1)
  (timeout 400 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OLDPWD/$OUT/prof_pmc3 -o pmc3 -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_pmc3.log 2>&1)
  (timeout 400 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OLDPWD/$OUT/prof_pmc4 -o pmc4 -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_pmc4.log 2>&1)
  cd $OLDPWD
  # keep the merge small: drop everything but csv/txt summaries
  find $OUT -type f \( -name "*.db" -o -name "*.pftrace" \) -delete
  find $OUT -type f -size +8M -delete
fi
tail -3 $OUT/pytest_gpu.log; cat $OUT/smoke.log | tail -2; cat $OUT/bench.json; for v in $VARIANTS; do cat $OUT/bench_$v.json; done
du -sh $OUT
