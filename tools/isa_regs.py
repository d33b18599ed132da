"""register / LDS / scratch use per kernel of a HIP source: python tools/isa_regs.py starcop_amd/csrc/conv_bx3.hip [pattern]"""
import re, subprocess, sys, os, tempfile
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = os.path.dirname(os.path.abspath(src)); out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
extra = ["-fno-slp-vectorize"] if ("bx3" in src or "conv_mfma" in src) else []
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{d}/../../include", f"-I{d}", "-Wno-unused-result",
                *extra, "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(k + r'\s+(\S+)', body) or [None, '?'])[1]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")[:64]
    if pat in dn:
        print(f"{dn:64s} vgpr {g('.amdhsa_next_free_vgpr'):>4s} accum_off {g('.amdhsa_accum_offset'):>4s} lds {g('.amdhsa_group_segment_fixed_size'):>6s} scratch {g('.amdhsa_private_segment_fixed_size')}")
