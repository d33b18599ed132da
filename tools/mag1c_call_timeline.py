"""timeline of the LAST call in a rocprofv3 --kernel-trace results.db of tools/mag1c_call_profile.py: every dispatch's start offset,
duration and the idle gap before it.  usage: mag1c_call_timeline.py results.db [dispatches_per_call]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select start, end, {name_col} from kernels order by start").fetchall()
# a call ends with the status reduction: find the last two occurrences of the tile kernel and print everything between them
tiles = [i for i, r in enumerate(rows) if "k_mag1c_tile" in r[2] and r[1] - r[0] > 100_000]
a, b = tiles[-2], tiles[-1]
t0 = rows[a][1]
print(f"previous call's filter kernel ends at 0; this call's ends at {(rows[b][1] - t0) / 1e3:.1f} us")
prev_end = t0
for r in rows[a + 1:b + 4]:
    print(f"  +{(r[0] - t0) / 1e3:8.1f} us  gap {(r[0] - prev_end) / 1e3:7.1f}  dur {(r[1] - r[0]) / 1e3:7.1f}  {r[2][:90]}")
    prev_end = max(prev_end, r[1])
