#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats (eager two-stream and serial) + PMC HBM traffic of k_conv3_bx3.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01h}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_eager /tmp/p_serial /tmp/p_fetch /tmp/p_write
rocprofv3 --kernel-trace -d /tmp/p_eager -o run -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras > $OUT/bench_eager.log 2>&1
rocprofv3 --kernel-trace -d /tmp/p_serial -o run -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras --overlap 0 > $OUT/bench_serial.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o run -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --graph 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o run -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --graph 0 > $OUT/pmc_write.log 2>&1
cd $ROOT
python tools/prof_summary.py $(find /tmp/p_eager -name "*.db" | head -1) > $OUT/kernel_trace_stats_bench_b16.txt
python tools/prof_summary.py $(find /tmp/p_serial -name "*.db" | head -1) > $OUT/kernel_trace_stats_bench_b16_serial.txt
python tools/gap_analysis.py $(find /tmp/p_eager -name "*.db" | head -1) 0.3 0.7 > $OUT/gap_analysis_eager.txt
python tools/pmc_traffic.py $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) "k_conv3_bx3|k_conv3_ws|k_conv3_sp" $OUT/pmc_traffic_conv3_bx3.json > /dev/null
grep "^{" $OUT/bench_eager.log > $OUT/bench_line_under_rocprof.json
python bench.py > $OUT/bench_full.log 2>&1; grep "^{" $OUT/bench_full.log > $OUT/bench_line.json
ls -la $OUT
