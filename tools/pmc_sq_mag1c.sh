#!/bin/bash
# SQ counters of the mag1c kernels (k_mag1c_tile) over tools/bench_mag1c.py: separate rocprofv3 passes (counters never combined with
# trace domains other than the kernel trace), then the derived shares (tools/pmc_sq_reading.py) against a kernel trace of the same build
set -u
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -f $OUT/pmc_sq_mag1c.txt
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm$i -o run -- python $ROOT/tools/bench_mag1c.py > $OUT/pmc_sq_mag1c_$i.log 2>&1
  python $ROOT/tools/pmc_summary.py $(find /tmp/pm$i -name "*.db" | head -1) "k_mag1c" >> $OUT/pmc_sq_mag1c.txt 2>> $OUT/pmc_sq_err.log < /dev/null
done
rm -rf /tmp/pmt
timeout 300 rocprofv3 --kernel-trace -d /tmp/pmt -o run -- python $ROOT/tools/bench_mag1c.py > /dev/null 2>&1
cd $ROOT
python tools/prof_summary.py $(find /tmp/pmt -name "*.db" | head -1) < /dev/null > $OUT/mag1c_trace_for_sq.txt
python tools/pmc_sq_reading.py $OUT/pmc_sq_mag1c.txt $OUT/mag1c_trace_for_sq.txt < /dev/null
