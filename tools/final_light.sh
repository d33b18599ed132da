#!/bin/bash
# the light end-of-round evidence set (kernels unchanged since the last tools/final_round.sh run): GPU test suite, smoke, full bench line,
# rocprofv3 kernel traces (eager + serial) with the gap / phase / tail analyses.   usage: bash tools/final_light.sh <tag>
set -u
TAG=${1:-r04t}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_eager /tmp/p_serial
rocprofv3 --kernel-trace -d /tmp/p_eager -o run -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras > $OUT/bench_eager.log 2>&1
rocprofv3 --kernel-trace -d /tmp/p_serial -o run -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras --overlap 0 > $OUT/bench_serial.log 2>&1
cd $ROOT
E=$(find /tmp/p_eager -name "*.db" | head -1); S=$(find /tmp/p_serial -name "*.db" | head -1)
python tools/prof_summary.py $E > $OUT/kernel_trace_stats_bench_b16.txt
python tools/prof_summary.py $S > $OUT/kernel_trace_stats_bench_b16_serial.txt
python tools/gap_analysis.py $E 0.3 0.7 > $OUT/gap_analysis_eager.txt
python tools/phase_gaps.py $E > $OUT/phase_gaps_eager.txt
python tools/trace_tail.py $E 16 > $OUT/trace_tail_eager.txt
python bench.py > $OUT/bench_full.log 2>&1; grep "^{" $OUT/bench_full.log > $OUT/bench_line.json
ls $OUT
