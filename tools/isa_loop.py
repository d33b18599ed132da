"""instruction mix of the innermost loops of one kernel: python tools/isa_loop.py <file.s> '<demangled substring>'"""
import re, subprocess, sys
s = open(sys.argv[1]).read(); want = sys.argv[2]
for nm in re.findall(r'^(_Z\S+):', s, re.M):
    dn = subprocess.run(['c++filt', nm], capture_output=True, text=True).stdout.strip()
    if want not in dn: continue
    body = s[s.index(nm + ':'):]; body = body[:body.index('.end_amdhsa_kernel')]
    lines = body.split('\n')
    # loops: from "Loop Header" label to the backward branch to it
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\S+):.*Loop Header: Depth=(\d+)', l)
        if not m: continue
        lab = m.group(1)
        end = max((j for j in range(i, len(lines)) if re.search(r's_cbranch\S+\s+' + re.escape(lab) + r'\b|s_branch\s+' + re.escape(lab) + r'\b', lines[j])), default=None)
        if end is None: continue
        b = [x.strip().split(';')[0] for x in lines[i:end + 1]]; b = [x for x in b if x and not x.startswith('.')]
        c = lambda pat: sum(1 for x in b if re.match(pat, x))
        print(f"{dn[:60]} loop {lab} depth {m.group(2)}: n={len(b)} mfma={c('v_mfma')} valu={c(r'v_(?!mfma)')} salu={c(r's_(?!waitcnt|barrier|nop|cbranch|branch)')} "
              f"branch={c(r's_c?branch')} ds_read={c('ds_read')} ds_write={c('ds_write')} vmem={c(r'(global|buffer)_(load|store)')} wait={c('s_waitcnt')} barrier={c('s_barrier')}")
