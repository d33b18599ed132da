"""Summarise a rocprofv3 --kernel-trace results.db into a per-kernel stats table (text)."""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace summary of {db}")
print(f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for n, cnt, s, a, mn, mx in rows[:40]:
    short = n[:88]
    print(f"{short:90s} {cnt:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
