#!/bin/bash
# everything the round's evidence needs, in one GPU call: full GPU test suite, smoke, bench (with extras + CPU baseline), rocprofv3
# kernel traces (eager + serial), gap analysis, PMC traffic of k_conv3_bx3, SQ counters of the split kernels, mag1c kernel trace
set -u
TAG=${1:-r02f}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/profile_round.sh $TAG > /dev/null 2>&1
bash tools/pmc_sq.sh $TAG > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_m
rocprofv3 --kernel-trace -d /tmp/p_m -o run -- python $ROOT/tools/bench_mag1c.py > $OUT/mag1c_bench_under_rocprof.log 2>&1
cd $ROOT; python tools/prof_summary.py $(find /tmp/p_m -name "*.db" | head -1) > $OUT/mag1c_kernel_trace.txt
python tools/bench_mag1c.py > $OUT/mag1c_bench.log 2>&1
for a in "--precision bf16 --batch 64" "--precision fp32 --batch 64" "--precision fp32-x3"; do
  python bench.py $a --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > "$OUT/bench_line_$(echo $a | tr -d ' -').json"
done
ls $OUT
