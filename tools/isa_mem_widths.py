"""global load / store instruction widths per kernel of a HIP source (a kernel that moves its bytes 4 per lane issues four times the
requests of one that moves 16): python tools/isa_mem_widths.py starcop_amd/csrc/conv_valu.hip [pattern]"""
import re, subprocess, sys, os, tempfile, collections
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = os.path.dirname(os.path.abspath(src)); out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".mw.s")
extra = ["-fno-slp-vectorize"] if any(k in src for k in ("bx3", "conv_mfma", "conv_sp", "conv_pw3", "conv_irt", "conv_irb")) else []
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{d}/../../include", f"-I{d}", "-Wno-unused-result",
                *extra, "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    dn = re.sub(r'\(.*', '', dn)[:70]
    if pat not in dn:
        continue
    c = collections.Counter(re.findall(r'\b((?:global|buffer|flat)_(?:load|store)_(?:dwordx4|dwordx3|dwordx2|dword|short|ushort|ubyte|byte|b\d+|lds_dword\w*)\w*)', body))
    if c:
        print(f"{dn:70s} " + "  ".join(f"{k.replace('global_', 'g_').replace('buffer_', 'b_')}:{v}" for k, v in sorted(c.items())))
