#!/bin/bash
# PMC passes over ONE layer of tools/conv_sweep.py (GPU box):  bash tools/kprof.sh <tag> <cin> <cout> <H> <W> <tile>
TAG=$1; shift
CIN=$1; COUT=$2; H=$3; W=$4; TILE=$5
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
ROOT=$PWD
export TMPDIR=/tmp
CMD="python $ROOT/tools/conv_sweep.py --one $CIN $COUT $H $W --tile $TILE --reps 3"
$CMD > $OUT/time.log 2>&1
cd /tmp
pmc() { local name=$1; shift; (timeout 300 rocprofv3 --output-format csv --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1); }
pmc p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES
pmc p2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA
pmc p3 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
pmc p4 TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pmc p5 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
cd $ROOT
find $OUT -type f \( -name "*.db" -o -name "*.pftrace" \) -delete
cat $OUT/time.log | tail -2
python $ROOT/tools/kprof_summary.py $OUT
