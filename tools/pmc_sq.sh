#!/bin/bash
# SQ counters of the split 3x3 kernels (MFMA busy, wave cycles, wait buckets, LDS conflicts): separate rocprofv3 passes, summary text
set -u
TAG=${1:-r01k}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pq$i -o run -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --overlap 0 > $OUT/pmc_sq_$i.log 2>&1
  python $ROOT/tools/pmc_summary.py $(find /tmp/pq$i -name "*.db" | head -1) "bx3|k_conv3_ws|k_conv3_sp|thin_h" >> $OUT/pmc_sq_conv3.txt 2>> $OUT/pmc_sq_err.log
done
cd $ROOT; tail -60 $OUT/pmc_sq_conv3.txt
