#!/bin/bash
# SQ counters (wave cycles, wait buckets, MFMA busy, instruction mix, LDS conflicts) of a kernel-name pattern: separate rocprofv3 passes
set -u
PAT=${1:-k_conv_mfma}
TAG=${2:-r02sq}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
: > $OUT/pmc_sq_$PAT.txt
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pq$i -o run -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --overlap 0 > $OUT/pmc_sq2_$i.log 2>&1
  python $ROOT/tools/pmc_summary.py $(find /tmp/pq$i -name "*.db" | head -1) "$PAT" >> $OUT/pmc_sq_$PAT.txt 2>> $OUT/pmc_sq_err.log
done
cd $ROOT; cat $OUT/pmc_sq_$PAT.txt
