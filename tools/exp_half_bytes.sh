#!/bin/bash
# the half-bytes prize, measured.  Alternating same-box runs of the shipped library and the elimination build of
# tools/build_exp_half.sh at batch 16 (fp32) and batch 64 (fp32, bf16 math) + the per-layer tables of the full-resolution group.
set -u
TAG=${1:-r06d}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
A=$ROOT/starcop_amd/libstarcop_hip.so; H=$ROOT/starcop_amd/libstarcop_hip_half.so
R=$OUT/half_bytes.txt; : > $R
run() { STARCOP_HIP_LIB=$1 python bench.py --no-cpu-baseline --no-extras "${@:2}" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$(basename $1)', '${*:2}', d['value'], 'tiles/s', d['ms_per_step'], 'ms/step')"; }
for rep in 1 2; do for L in $A $H; do run $L --batch 16 --steps 40 --warmup 8 >> $R; done; done
for rep in 1 2; do for L in $A $H; do run $L --batch 64 --steps 12 --warmup 3 >> $R; done; done
for rep in 1 2; do for L in $A $H; do run $L --batch 64 --steps 12 --warmup 3 --precision bf16 >> $R; done; done
for L in $A $H; do
  echo "== per layer, batch 16, $(basename $L)" >> $R
  STARCOP_HIP_LIB=$L python tools/bench_layers.py --reps 6 2>/dev/null | grep -E " d3| d4|logits|^total" >> $R
done
for L in $A $H; do
  echo "== per layer, batch 64, $(basename $L)" >> $R
  STARCOP_HIP_LIB=$L python tools/bench_layers.py --reps 3 --batch 64 2>/dev/null | grep -E " d3| d4|logits|^total" >> $R
done
cat $R
