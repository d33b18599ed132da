#!/bin/bash
# SQ counters of the pointwise families (conv_pw3.hip: k_pw3 / k_pw3_wgrad; conv_mfma.hip: k_conv_mfma<1>, k_conv1_ksplit, k_wgrad_mfma<1>):
# separate rocprofv3 passes over a serial bench run with every pointwise launch on the new family (STARCOP_PW3=all) and with it off
set -u
TAG=${1:-r03g}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in all 0; do
  i=0
  for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAVES"; do
    i=$((i+1)); rm -rf /tmp/pq$i
    STARCOP_PW3=$mode timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pq$i -o run -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --overlap 0 > $OUT/pmc_sq_pw_$i.log 2>&1
    python $ROOT/tools/pmc_summary.py $(find /tmp/pq$i -name "*.db" | head -1) "k_pw3|k_conv_mfma<1|k_conv1_ksplit|k_wgrad_mfma<1" >> $OUT/pmc_sq_pointwise_pw3_$mode.txt 2>> $OUT/pmc_sq_err.log
  done
  rm -rf /tmp/pt
  STARCOP_PW3=$mode rocprofv3 --kernel-trace -d /tmp/pt -o run -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --overlap 0 > /dev/null 2>&1
  python $ROOT/tools/prof_summary.py $(find /tmp/pt -name "*.db" | head -1) > $OUT/trace_serial_pw3_$mode.txt
  python $ROOT/tools/pmc_sq_reading.py $OUT/pmc_sq_pointwise_pw3_$mode.txt $OUT/trace_serial_pw3_$mode.txt > /dev/null
done
cd $ROOT; tail -30 $OUT/pmc_sq_pointwise_pw3_all.txt
