#!/bin/bash
# LDS bank conflicts of k_wgrad3_bx3 attributed by elimination builds (tools/build_exp_wgrad3.sh) under the SQ counters:
# one rocprofv3 --pmc pass per build over tools/exp_wgrad3.py + an un-profiled timing run.   usage: bash tools/exp_wgrad3_pmc.sh <tag> [builds...]
set -u
TAG=${1:-r06b}; shift
BUILDS=${*:-"0 1 2 3 4 5 6"}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
RES=$OUT/wgrad3_lds_elimination.txt
echo "# k_wgrad3_bx3 elimination builds: 0 shipped | 1 no input LDS stores | 2 no dy LDS stores | 3 no A (dy) reads | 4 no B (input) reads | 5 no second B read | 6 no constant reads" > $RES
cd /tmp && export TMPDIR=/tmp
for b in $BUILDS; do
  LIB=$ROOT/starcop_amd/libstarcop_hip.so; [ "$b" != "0" ] && LIB=$ROOT/starcop_amd/libstarcop_hip_wg$b.so
  echo "## build $b ($(basename $LIB))" >> $RES
  STARCOP_HIP_LIB=$LIB python $ROOT/tools/exp_wgrad3.py 10 >> $RES 2>&1
  rm -rf /tmp/pw$b
  STARCOP_HIP_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pw$b -o run -- python $ROOT/tools/exp_wgrad3.py 2 > $OUT/wg_pmc_$b.log 2>&1
  python $ROOT/tools/pmc_summary.py $(find /tmp/pw$b -name "*.db" | head -1) "k_wgrad3_bx3" >> $RES 2>> $OUT/wg_pmc_err.log
done
cd $ROOT; cat $RES
