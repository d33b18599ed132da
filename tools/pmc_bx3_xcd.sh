#!/bin/bash
# FETCH_SIZE of k_conv3_bx3 with and without the XCD-aware work-group numbering + timing of both
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_xcd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2; do
  rm -rf /tmp/px$v
  STARCOP_BX3_XCDMAP=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/px$v -o run -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --overlap 0 > $OUT/f$v.log 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/px$v/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
namecol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
n, s = c.execute(f"select count(distinct dispatch_id), sum(value) from counters_collection where kernel_name like '%k_conv3_bx3%' and {namecol}='FETCH_SIZE'").fetchone()
print("xcdmap=$v: k_conv3_bx3 launches", n, "FETCH_SIZE raw per launch MB", s * 1024 / n / 1e6)
PY
  cd $ROOT; STARCOP_BX3_XCDMAP=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcdmap=$v', d['value'], d['ms_per_step'], d['roofline']['families_ms_per_step']['k_conv3_bx3 (fwd+dgrad)'])"
  cd /tmp
done
