"""HBM traffic per launch of a kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; units of KiB).
gfx950 note (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reports wide (16 B/lane) coalesced streams by 2x; these kernels
issue 4 B/lane loads, for which the counter is uncalibrated -> reported raw, with the 2x figure alongside."""
import json, sqlite3, sys

fetch_db, write_db, pat, out = sys.argv[1:5]


def per_launch(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
    pats = pat.split("|")               # "k_conv3_bx3|k_conv3_ws": launches of either kernel template
    where = " or ".join(["kernel_name like ?"] * len(pats))
    rows = c.execute(f"select count(distinct dispatch_id), sum(value) from counters_collection where ({where}) and {namecol} = ?",
                     (*[f"%{x}%" for x in pats], counter)).fetchone()
    return rows[0], rows[1] * 1024.0 / rows[0]


def steps_in(db):
    """training steps in the profiled run = dispatches of the loss kernel (one per step)"""
    c = sqlite3.connect(db)
    return c.execute("select count(distinct dispatch_id) from counters_collection where kernel_name like '%k_bce%'").fetchone()[0]


nf, fb = per_launch(fetch_db, "FETCH_SIZE")
nw, wb = per_launch(write_db, "WRITE_SIZE")
nsteps = steps_in(fetch_db)
res = {"kernel_pattern": pat, "launches": nf, "fetch_bytes_per_launch_raw": fb, "fetch_bytes_per_launch_x2": 2 * fb,
       "write_bytes_per_launch": wb, "hbm_bytes_per_launch_raw": fb + wb,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --steps 3 --warmup 1 --graph 0`"}
# calibrated figure (profiles/r02_pmc_calibration.json: what FETCH_SIZE reports of a KNOWN 1 GiB stream in a 4 B/lane access pattern)
import os
cal_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r02_pmc_calibration.json")
if os.path.exists(cal_path):
    cal = json.load(open(cal_path))
    f = [v["fetch_reported_over_known"] for v in cal.values() if isinstance(v, dict) and "fetch_reported_over_known" in v][0]
    res["fetch_calibration"] = {"FETCH_SIZE_reported_over_known": f, "file": "profiles/r02_pmc_calibration.json"}
    res["hbm_bytes_per_launch_corrected"] = fb / f + wb
    # A layer's data gradient may be several physical launches (up-sampled / skip channels): per STEP the figure is independent of how the
    # work is cut into launches -- bench.py divides it by its LOGICAL launches per step (one per layer and direction), the unit of its
    # algorithmic bytes.
    if nsteps:
        res["steps"] = nsteps
        res["launches_per_step"] = nf / nsteps
        res["hbm_bytes_per_step_corrected"] = (fb / f + wb) * nf / nsteps
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
