"""Busy / idle analysis of a rocprofv3 --kernel-trace results.db: union of kernel intervals over the steady-state steps,
idle time between consecutive kernels, and the share of tiny kernels.  usage: gap_analysis.py results.db [first_fraction [last_fraction]]"""
import sqlite3
import sys

db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
stop = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
qcol = "queue_id" if "queue_id" in cols else None
rows = c.execute(f"select start, end, {name_col}{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
rows = rows[int(len(rows) * skip):int(len(rows) * stop)]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for r in rows[1:]:
    if r[0] > cur_e:
        busy += cur_e - cur_s
        gaps.append(r[0] - cur_e)
        cur_s, cur_e = r[0], r[1]
    else:
        cur_e = max(cur_e, r[1])
busy += cur_e - cur_s
ksum = sum(r[1] - r[0] for r in rows)
tiny = [r for r in rows if r[1] - r[0] < 10_000]
print(f"dispatches {len(rows)}  span {span/1e6:.2f} ms  device busy (union) {busy/1e6:.2f} ms = {100*busy/span:.1f}%  "
      f"sum of kernel durations {ksum/1e6:.2f} ms (overlap {100*(ksum-busy)/span:.1f}% of span)")
print(f"idle gaps: {len(gaps)} totalling {sum(gaps)/1e6:.2f} ms; >2us: {sum(1 for g in gaps if g > 2000)}, "
      f">5us: {sum(1 for g in gaps if g > 5000)}, >20us: {sum(1 for g in gaps if g > 20000)}")
print(f"kernels shorter than 10 us: {len(tiny)} ({100*len(tiny)/len(rows):.0f}% of dispatches), {sum(r[1]-r[0] for r in tiny)/1e6:.2f} ms")
if qcol:
    qs = {}
    for r in rows:
        qs.setdefault(r[3], [0, 0]); qs[r[3]][0] += 1; qs[r[3]][1] += r[1] - r[0]
    for q, (n, t) in qs.items():
        print(f"  queue {q}: {n} dispatches, {t/1e6:.2f} ms")
