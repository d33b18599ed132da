#!/bin/bash
# SQ counters of the ENCODER / BatchNorm families as they run in the default step (the last table for them was
# profiles/r02_pmc_sq_pointwise.txt): pointwise forward / data gradient (k_conv_mfma<1>, k_pw3, k_conv1_ksplit), pointwise weight
# gradient (k_wgrad_mfma<1>, k_pw3_wgrad), depthwise (k_dw_*), the fused block (k_irt_*), BatchNorm reductions.  Separate rocprofv3
# passes (two counters each) over a serial bench run + a serial kernel trace of the same build for the durations.
# usage (GPU box, repo root): bash tools/pmc_sq_enc.sh <tag>
set -u
TAG=${1:-r06a}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
PAT='k_conv_mfma<1|k_pw3|k_conv1_ksplit|k_wgrad_mfma<1|k_dw_|k_irt_|k_bn_bwd_reduce|k_bn_bwd_small'
cd /tmp && export TMPDIR=/tmp
rm -f $OUT/pmc_sq_encoder.txt
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1)); rm -rf /tmp/pe$i
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pe$i -o run -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --overlap 0 > $OUT/pmc_sq_enc_$i.log 2>&1
  python $ROOT/tools/pmc_summary.py $(find /tmp/pe$i -name "*.db" | head -1) "$PAT" >> $OUT/pmc_sq_encoder.txt 2>> $OUT/pmc_sq_err.log
done
rm -rf /tmp/pet
timeout 300 rocprofv3 --kernel-trace -d /tmp/pet -o run -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --overlap 0 > /dev/null 2>&1
python $ROOT/tools/prof_summary.py $(find /tmp/pet -name "*.db" | head -1) > $OUT/trace_serial_for_sq.txt
python $ROOT/tools/pmc_sq_reading.py $OUT/pmc_sq_encoder.txt $OUT/trace_serial_for_sq.txt > /dev/null
cd $ROOT; tail -40 $OUT/pmc_sq_encoder.txt
