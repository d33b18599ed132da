#!/bin/bash
# SQ counters of the fused inverted-residual training kernels (conv_irt.hip) over tools/bench_irt.py: separate rocprofv3 passes
set -u
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -f $OUT/pmc_sq_irt.txt
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pi$i
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pi$i -o run -- python $ROOT/tools/bench_irt.py --reps 3 > $OUT/pmc_sq_irt_$i.log 2>&1
  python $ROOT/tools/pmc_summary.py $(find /tmp/pi$i -name "*.db" | head -1) "k_irt" >> $OUT/pmc_sq_irt.txt 2>> $OUT/pmc_sq_err.log
done
rm -rf /tmp/pt
rocprofv3 --kernel-trace -d /tmp/pt -o run -- python $ROOT/tools/bench_irt.py --reps 5 > /dev/null 2>&1
python $ROOT/tools/prof_summary.py $(find /tmp/pt -name "*.db" | head -1) | grep -E "k_irt|total" > $OUT/trace_irt.txt
cd $ROOT; cat $OUT/trace_irt.txt; cat $OUT/pmc_sq_irt.txt
