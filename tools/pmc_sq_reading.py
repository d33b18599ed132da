"""Append the derived shares to a pmc_sq summary: python tools/pmc_sq_reading.py <pmc_sq txt> <serial kernel-trace summary txt>
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x average dispatch duration x 2.4 GHz); wave-cycle shares = counter /
SQ_WAVE_CYCLES (both in quad-cycles): issuing = SQ_ACTIVE_INST_ANY, parked = SQ_WAIT_ANY (s_waitcnt / barrier), issue-stalled =
SQ_WAIT_INST_ANY."""
import re, sys
sq, trace = sys.argv[1:3]
dur = {}
for l in open(trace):
    m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l.rstrip())
    if m: dur[m.group(1).strip()[:90]] = float(m.group(4))
ctr, cur = {}, None
for l in open(sq):
    if l.startswith("#"): continue
    if not l.startswith(" ") and l.strip(): cur = l.strip()[:90]; ctr.setdefault(cur, {})
    elif cur and l.strip():
        p = l.split(); ctr[cur][p[0]] = float(p[1])
out = ["", "# reading (tools/pmc_sq_reading.py): MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x average dispatch duration x 2.4 GHz);",
       "# durations from the serial kernel trace of the same build; wave-cycle shares = counter / SQ_WAVE_CYCLES"]
for k, c in ctr.items():
    d = next((v for kk, v in dur.items() if kk[:80] == k[:80]), None)
    if d is None or "SQ_WAVE_CYCLES" not in c: continue
    wc = c["SQ_WAVE_CYCLES"]
    busy = 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * d * 1e-6 * 2.4e9)
    lds = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0
    short = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:44]
    out.append(f"# {short:44s} avg {d:7.1f} us  MFMA pipe busy {busy:5.1f}%  issuing {100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.1f}%  "
               f"parked {100*c.get('SQ_WAIT_ANY',0)/wc:5.1f}%  issue-stalled {100*c.get('SQ_WAIT_INST_ANY',0)/wc:5.1f}%  LDS conflict / active {lds:.2f}")
open(sq, "a").write("\n".join(out) + "\n")
print("\n".join(out))
