#!/bin/bash
# same-box A/B of an environment variable: alternating bench runs.   usage: bash tools/ab_env.sh VAR "v1 v2 v3" [pairs=2] [bench args]
VAR=$1; VALS=$2; PAIRS=${3:-2}; shift; shift; shift
for rep in $(seq $PAIRS); do for v in $VALS; do
  env $VAR=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['roofline']['families_ms_per_step']
print('$VAR=$v', d['value'], d['ms_per_step'])"
done; done
