"""Per-kernel PMC counter sums from a rocprofv3 --pmc results.db:  python tools/pmc_summary.py db [kernel-substring[|substring...]]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
namecol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
valcol = "value" if "value" in cols else [x for x in cols if "value" in x][0]
q = f"select kernel_name, {namecol}, count(distinct dispatch_id), sum({valcol}) from counters_collection group by kernel_name, {namecol}"
agg = {}
for k, n, d, v in c.execute(q):
    if any(x in k for x in pat.split('|')):
        agg.setdefault(k, {})[n] = (d, v)
for k, d in agg.items():
    print(k[:110])
    nd = max(x[0] for x in d.values())
    for n, (dd, v) in sorted(d.items()):
        print(f"    {n:32s} {v/nd:16.1f}  per dispatch ({nd} dispatches)")
