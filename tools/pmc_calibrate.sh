#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of kernels with a KNOWN byte count (tools/calib_workload.py): separate rocprofv3 passes, summary json
set -u
TAG=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_f /tmp/pc_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pc_f -o run -- python $ROOT/tools/calib_workload.py > $OUT/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pc_w -o run -- python $ROOT/tools/calib_workload.py > $OUT/calib_write.log 2>&1
cd $ROOT
python - <<PY > $OUT/pmc_calibration.json
import json, sqlite3, glob
def per(db, counter, pat):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
    n, v = c.execute(f"select count(distinct dispatch_id), sum(value) from counters_collection where kernel_name like ? and {namecol} = ?", (f"%{pat}%", counter)).fetchone()
    return n, (v or 0) * 1024.0 / max(n, 1)
fd = glob.glob("/tmp/pc_f/**/*.db", recursive=True)[0]; wd = glob.glob("/tmp/pc_w/**/*.db", recursive=True)[0]
known = 2 ** 30
out = {"known_bytes_per_launch_each_way": known}
for name, pat in (("k_add_srcs (4 B/lane loads + stores)", "k_add_srcs"), ("torch copy (16 B/lane)", "elementwise")):
    nf, f = per(fd, "FETCH_SIZE", pat); nw, w = per(wd, "WRITE_SIZE", pat)
    out[name] = {"launches": nf, "FETCH_SIZE_bytes": f, "fetch_reported_over_known": f / known, "WRITE_SIZE_bytes": w, "write_reported_over_known": w / known}
print(json.dumps(out, indent=1))
PY
cat $OUT/pmc_calibration.json
