#!/bin/bash
# Elimination builds of k_irb (conv_irb.hip): ONE ingredient removed, results wrong, timings valid.  -> starcop_amd/libstarcop_hip_irb{N}.so
#   1 no stencil phase (P2)   2 no expansion MFMAs   3 no projection MFMAs   4 no filter / constant loads inside the chunk loop
#   5 no barriers in the chunk loop   6 no expansion epilogue (BN, mask, s_e stores)   7 no MFMAs at all
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/starcop_amd/csrc"
make -s
python3 - <<'PY'
s = open("conv_irb.hip").read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (a, s.count(a))
    s = s.replace(a, b)
rep("            irb_mfma6(a, be[q][ks], acc, acc1);", "            if (IRB_EXP != 2 && IRB_EXP != 7) irb_mfma6(a, be[q][ks], acc, acc1); else asm volatile(\"\" :: \"v\"(a[0]), \"v\"(a[1]), \"v\"(a[2]), \"v\"(be[q][ks][0]), \"v\"(be[q][ks][1]), \"v\"(be[q][ks][2]));")
rep("          irb_mfma6(a, bp[0][ks], accp[0], accp1[0]);", "          if (IRB_EXP != 3 && IRB_EXP != 7) irb_mfma6(a, bp[0][ks], accp[0], accp1[0]); else asm volatile(\"\" :: \"v\"(a[0]), \"v\"(a[1]), \"v\"(a[2]), \"v\"(bp[0][ks][0]), \"v\"(bp[0][ks][1]), \"v\"(bp[0][ks][2]));")
rep("    // ---- P2: depthwise 3x3 + BN_d + ReLU6 + split for (output pixel opx, channels okg*8 .. +7)\n    {", "    // ---- P2\n    if (IRB_EXP != 1) {")
rep("    p_request(c);\n    k_request(c + 1);", "    if (IRB_EXP != 4) { p_request(c); k_request(c + 1); }")
rep("    e_request(c + 1);\n    __builtin_amdgcn_sched_barrier(0);\n    __syncthreads();", "    if (IRB_EXP != 4) e_request(c + 1);\n    __builtin_amdgcn_sched_barrier(0);\n    if (IRB_EXP != 5) __syncthreads();")
rep("    k_store(buf ^ 1);          // (past the last chunk: the same values again, never read)\n    __syncthreads();", "    k_store(buf ^ 1);\n    if (IRB_EXP != 5) __syncthreads();")
rep("#pragma unroll\n        for (int i = 0; i < 16; ++i) {\n          const int px = emb[q] * 32 + 8 * (i >> 2) + 4 * lhi + (i & 3);\n          const float v = __builtin_amdgcn_fmed3f(fmaf(acc[i], sc, sh), 0.f, 6.f);",
    "#pragma unroll\n        for (int i = 0; i < (IRB_EXP == 6 ? 1 : 16); ++i) {\n          const int px = emb[q] * 32 + 8 * (i >> 2) + 4 * lhi + (i & 3);\n          const float v = __builtin_amdgcn_fmed3f(fmaf(acc[i], sc, sh), 0.f, 6.f);")
open("/tmp/conv_irb_exp.hip", "w").write("#ifndef IRB_EXP\n#define IRB_EXP 0\n#endif\n" + s)
PY
cp sc_common.h /tmp/
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -fno-slp-vectorize -DIRB_EXP=$n -c /tmp/conv_irb_exp.hip -o /tmp/conv_irb_exp$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_mfma.o conv_bx3.o conv_sp.o conv_spw.o conv_pw3.o conv_irt.o /tmp/conv_irb_exp$n.o conv_valu.o elementwise.o mag1c.o features.o validation.o host_io.o -o ../libstarcop_hip_irb$n.so
done
ls ../libstarcop_hip_irb*.so
