"""Forward / backward split of the eager step in a rocprofv3 --kernel-trace results.db: span, busy union and idle time of each phase
(forward = from the stem convolution up to the loss kernel, backward + update + filter packs = the rest), and the largest idle gaps with the
kernels on either side.  usage: phase_gaps.py results.db"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select start, end, {name_col}, stream_id from kernels order by start").fetchall()
rows = rows[len(rows) // 3:]
adam = [i - 1 for i, r in enumerate(rows) if "k_stem_fwd" in r[2] and i > 0]      # last kernel before each step's stem forward
loss = [i for i, r in enumerate(rows) if "k_bce(" in r[2]]
def union(seg):
    busy, gaps, cs, ce = 0, [], seg[0][0], seg[0][1]
    prev = seg[0]
    for r in seg[1:]:
        if r[0] > ce:
            busy += ce - cs; gaps.append((r[0] - ce, prev[2][:90], r[2][:90])); cs, ce = r[0], r[1]
        else:
            ce = max(ce, r[1])
        if r[1] >= ce: prev = r
    return busy + ce - cs, gaps
fw = bw = fwb = bwb = 0; allg = []; n = 0
for a0, a1 in zip(adam[:-1], adam[1:]):
    ls = [i for i in loss if a0 < i < a1]
    if not ls: continue
    l = ls[0]
    if len(set(r[3] for r in rows[a0 + 1:a1 + 1])) < 2: continue      # bench.py's instrumented serial steps (one stream): not the eager step
    f, b = rows[a0 + 1:l + 1], rows[l + 1:a1 + 1]
    fb, fg = union(f); bb, bg = union(b)
    fw += f[-1][1] - f[0][0]; bw += max(r[1] for r in b) - b[0][0]; fwb += fb; bwb += bb; allg += [("fwd",) + g for g in fg] + [("bwd",) + g for g in bg]; n += 1
print(f"{n} steps: forward span {fw/n/1e6:.3f} ms busy {fwb/n/1e6:.3f} idle {(fw-fwb)/n/1e6:.3f} | backward+update span {bw/n/1e6:.3f} ms busy {bwb/n/1e6:.3f} idle {(bw-bwb)/n/1e6:.3f}")
import collections
agg = collections.defaultdict(lambda: [0, 0])
for ph, g, a, b in allg:
    nm = lambda t: (re.search(r"(k_\w+|__amd\w+|\w+_kernel)", t) or [t[:40]])[0]
    k = (ph, nm(a), nm(b)); agg[k][0] += g; agg[k][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {v[0]/n/1e3:7.1f} us/step  x{v[1]/n:5.1f}  {k[0]}  {k[1]}  ->  {k[2]}")
