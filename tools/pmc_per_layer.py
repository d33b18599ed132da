"""per-(kernel, grid) FETCH_SIZE of a rocprofv3 --pmc results.db: which layers of a kernel family over-fetch"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "k_conv3_bx3"
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
print("columns:", cols)
namecol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
gcols = [x for x in cols if "grid" in x.lower()]
key = ", ".join(["kernel_name"] + gcols)
q = f"select {key}, count(distinct dispatch_id), sum(value) from counters_collection where kernel_name like ? and {namecol}='FETCH_SIZE' group by {key} order by sum(value) desc"
for row in c.execute(q, (f"%{pat}%",)):
    name = row[0][:60]; g = row[1:-2]; n, v = row[-2], row[-1]
    print(f"{name:60s} grid {g}  launches {n:3d}  FETCH raw/launch {v*1024/n/1e6:9.1f} MB  x2 = {2*v*1024/n/1e6:9.1f} MB")
