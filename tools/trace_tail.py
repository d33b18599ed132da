"""last kernels of one eager step in a rocprofv3 --kernel-trace results.db (queue, start us, duration us, kernel) and the time each
queue alone / both / none were busy: python tools/trace_tail.py results.db [n_last]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1]); nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = c.execute("select start, end, name, stream_id from kernels order by start").fetchall()
rows = rows[len(rows) // 3:]
first = [i for i, r in enumerate(rows) if "k_stem_fwd" in r[2]]      # a step = [its stem forward, the next step's stem forward)
k = next(i for i in range(2, len(first) - 1) if len(set(r[3] for r in rows[first[i]:first[i + 1]])) > 1)      # an eager (two-stream) step
first = first[k - 2:]
seg = rows[first[2]:first[3]]; t0 = seg[0][0]
ev = sorted([(s, q, 1) for s, e, n, q in seg] + [(e, q, -1) for s, e, n, q in seg])
act, last, acc = {}, t0, {}
for t, q, d in ev:
    key = tuple(sorted(k for k, v in act.items() if v > 0)); acc[key] = acc.get(key, 0) + t - last; last = t
    act[q] = act.get(q, 0) + d
print("step %.3f ms;" % ((rows[first[3]][0] - t0) / 1e6), "; ".join(f"streams {k or 'none'}: {v / 1e6:.3f} ms" for k, v in sorted(acc.items())))
nm = lambda t: (re.search(r"(k_\w+)", t) or [t[:30]])[0]
for s, e, n, q in seg[-nlast:]:
    print(f"  s{q} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {nm(n)}")
