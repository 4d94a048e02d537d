#!/bin/bash
# Everything the round's evidence needs, in one GPU-box session (about 4 GPU-minutes).  usage: bash tools/gpu_final_round.sh <tag>
TAG=${1:-final}
export TMPDIR=/tmp
ROOT=$PWD
PROFILE=1 bash tools/gpu_round.sh ${TAG}_inf > gpurun_out/${TAG}_inf.out 2>&1
SKIP_TESTS=1 PROFILE=1 bash tools/gpu_train_round.sh ${TAG}_tr > gpurun_out/${TAG}_tr.out 2>&1
mkdir -p gpurun_out/${TAG}_bf16
(timeout 400 python tools/train_bench.py --precision bf16 --cfg 3 4 5 --steps 5 --profile-nodes 2>gpurun_out/${TAG}_bf16/train_nodes.log | tail -3) > gpurun_out/${TAG}_bf16/train_bench.log
(timeout 300 python tools/s3fd_bench.py 2>&1 | tail -1) > gpurun_out/${TAG}_inf/s3fd_bench.log
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/${TAG}_inf/bench_torchrun1.json
cd /tmp
for P in f32 bf16; do
  (timeout 400 rocprofv3 --output-format csv --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES -d $ROOT/gpurun_out/${TAG}_tr/pmc_$P -o pmc -- python $ROOT/tools/train_bench.py --precision $P --cfg 4 --steps 2 --warmup 2 > $ROOT/gpurun_out/${TAG}_tr/pmc_$P.log 2>&1)
  f=$(find $ROOT/gpurun_out/${TAG}_tr/pmc_$P -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $ROOT/tools/pmc_by_kernel.py $f > $ROOT/gpurun_out/${TAG}_tr/pmc_${P}_by_kernel.txt
done
cd $ROOT
find gpurun_out/${TAG}_tr -name "*counter_collection.csv" -delete
find gpurun_out -type f \( -name "*.db" -o -name "*.pftrace" \) -delete
tail -4 gpurun_out/${TAG}_inf.out | cut -c1-250; tail -4 gpurun_out/${TAG}_tr.out | cut -c1-250; cat gpurun_out/${TAG}_bf16/train_bench.log | cut -c1-250
cat gpurun_out/${TAG}_inf/s3fd_bench.log gpurun_out/${TAG}_inf/bench_torchrun1.json | cut -c1-250
head -12 gpurun_out/${TAG}_tr/pmc_f32_by_kernel.txt | cut -c1-260
