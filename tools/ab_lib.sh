#!/bin/bash
# same-box A/B of two builds of the library (STARCOP_HIP_LIB): alternating bench runs, tiles/s + ms/step + the serial family table entries
# that moved.   usage: bash tools/ab_lib.sh <libA.so> <libB.so> [pairs=2] [bench args]
A=$1; B=$2; PAIRS=${3:-2}; shift; shift; shift
for rep in $(seq $PAIRS); do for L in $A $B; do
  STARCOP_HIP_LIB=$L python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['roofline']['families_ms_per_step']
print('$(basename $L)', d['value'], d['ms_per_step'], {k.split(' ')[0]: v for k, v in f.items()})"
done; done
