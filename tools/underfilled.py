"""launches that do not fill the chip: per (kernel, grid) of a rocprofv3 --kernel-trace results.db the work-group count and the mean
duration, sorted by time, for launches with fewer than `limit` work-groups (default 512).  usage: underfilled.py results.db [limit]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
g = [x for x in cols if x.lower() in ("grid_x", "grid_size_x", "grid_size")]
w = [x for x in cols if x.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")]
print("columns:", cols)
gx = "grid_x" if "grid_x" in cols else g[0]
wx = "workgroup_x" if "workgroup_x" in cols else w[0]
extra = ""
if "grid_y" in cols:
    extra = ", grid_y, grid_z, workgroup_y, workgroup_z"
rows = c.execute(f"select {name_col}, {gx}, {wx}, end - start{extra} from kernels").fetchall()
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    wgs = r[1] // max(r[2], 1)
    if extra:
        wgs *= (r[4] // max(r[6], 1)) * (r[5] // max(r[7], 1))
    k = (r[0][:100], wgs, r[2])
    agg[k][0] += r[3]; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"total kernel time {tot/1e6:.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if k[1] < limit and v[0] / tot > 0.002:
        print(f"{v[0]/tot*100:5.2f} %  {v[0]/v[1]/1e3:8.1f} us x{v[1]:4d}  WGs {k[1]:5d} x {k[2]:4d} thr  {k[0]}")
