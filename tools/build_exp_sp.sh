#!/bin/bash
# Elimination builds of conv_sp.hip (DESIGN.md section 10 method: the same source with ONE ingredient removed; results are wrong by
# construction, timings valid):  tools/build_exp_sp.sh 1 2 3 ... -> starcop_amd/libstarcop_hip_sp{1,2,3}.so  (STARCOP_HIP_LIB=...)
#   1 no MFMAs (operand reads kept alive)   2 no operand reads, no MFMAs   3 no global patch loads   4 no prologue / split / patch stores
#   5 no barriers in the K loop              6 no filter loads / stores
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/starcop_amd/csrc"
make -s
python3 - <<'PY'
s = open("conv_sp.hip").read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("            acc[pp][px] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[px][b][t == 1 ? 1 : 0], B[o][t == 0 ? 1 : 0], acc[pp][px], 0, 0, 0);",
    "#if SP_EXP == 1\n            asm volatile(\"\" :: \"v\"(A[px][b][t == 1 ? 1 : 0]), \"v\"(B[o][t == 0 ? 1 : 0]));\n#else\n            acc[pp][px] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[px][b][t == 1 ? 1 : 0], B[o][t == 0 ? 1 : 0], acc[pp][px], 0, 0, 0);\n#endif")
rep("    compute(kc & 1);\n    __syncthreads();", "#if SP_EXP != 2\n    compute(kc & 1);\n#endif\n#if SP_EXP != 5\n    __syncthreads();\n#endif")
rep("      for (int j = 0; j < 4; ++j) xv[r][j] = xb[(size_t)(j < jmax ? j : jmax) * plane + o];",
    "#if SP_EXP == 3\n      for (int j = 0; j < 4; ++j) xv[r][j] = (float)(o & 7);\n#else\n      for (int j = 0; j < 4; ++j) xv[r][j] = xb[(size_t)(j < jmax ? j : jmax) * plane + o];\n#endif")
rep("    uint2* const sp2 = reinterpret_cast<uint2*>(s_p + (size_t)buf * NT * 2 * NPX);\n#pragma unroll\n    for (int r = 0; r < NR; ++r) {",
    "    uint2* const sp2 = reinterpret_cast<uint2*>(s_p + (size_t)buf * NT * 2 * NPX);\n#pragma unroll\n    for (int r = 0; r < (SP_EXP == 4 ? 0 : NR); ++r) {")
rep("    for (int j = 0; j < NWV; ++j) wv[j] = wsrc[tid + 512 * j];", "    for (int j = 0; j < (SP_EXP == 6 ? 0 : NWV); ++j) wv[j] = wsrc[tid + 512 * j];")
rep("    for (int j = 0; j < NWV; ++j) s_w[buf * WST + tid + 512 * j] = wv[j];", "    for (int j = 0; j < (SP_EXP == 6 ? 0 : NWV); ++j) s_w[buf * WST + tid + 512 * j] = wv[j];")
open("/tmp/conv_sp_exp.hip", "w").write("#ifndef SP_EXP\n#define SP_EXP 0\n#endif\n" + s)
PY
cp conv_sp_pack.h /tmp/
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -fno-slp-vectorize -DSP_EXP=$n -c /tmp/conv_sp_exp.hip -o /tmp/conv_sp_exp$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_mfma.o conv_bx3.o /tmp/conv_sp_exp$n.o conv_pw3.o conv_irt.o conv_valu.o elementwise.o mag1c.o features.o validation.o host_io.o -o ../libstarcop_hip_sp$n.so
done
ls ../libstarcop_hip_sp*.so
