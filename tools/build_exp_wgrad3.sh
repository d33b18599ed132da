#!/bin/bash
# Elimination builds of k_wgrad3_bx3 (conv_bx3.hip) for attributing its LDS bank conflicts: the same source with ONE
# LDS access class removed -- results wrong by construction, counters / timings valid.
#   tools/build_exp_wgrad3.sh 1 2 3 4 5 -> starcop_amd/libstarcop_hip_wg{N}.so   (STARCOP_HIP_LIB=...)
#   1 no input (x) LDS stores   2 no dy LDS stores   3 no A (dy) operand reads   4 no B (input) operand reads   5 no second B read (X1)
#   6 no reads of the prologue constants s_ca / s_cb in the store items
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/starcop_amd/csrc"
make -s
python3 - <<'PY'
s = open("conv_bx3.hip").read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, cnt)
# inside k_wgrad3_bx3 only (the first occurrences after its header)
i = s.index("void k_wgrad3_bx3(const WgradXP p)")
j = s.index("void k_wgrad_thin_h(const WgradXP p)")
head, body, tail = s[:i], s[i:j], s[j:]
s = body
rep("        for (int c = 0; c < NT; ++c) s_x[c][d] = t[c];", "        for (int c = 0; c < (WG_EXP == 1 ? 0 : NT); ++c) s_x[c][d] = t[c];")
rep("        for (int c = 0; c < NT; ++c) s_dy[buf][c][d] = t[c];", "        for (int c = 0; c < (WG_EXP == 2 ? 0 : NT); ++c) s_dy[buf][c][d] = t[c];")
rep("        for (int t = 0; t < NT; ++t) A[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uintx4*>(&s_dy[buf][t][da]));",
    "        for (int t = 0; t < NT; ++t) A[t] = WG_EXP == 3 ? __builtin_bit_cast(bf16x8, (uintx4){(unsigned)da, 1u, 2u, 3u}) : __builtin_bit_cast(bf16x8, *reinterpret_cast<const uintx4*>(&s_dy[buf][t][da]));")
rep("          const uintx4 X0 = *reinterpret_cast<const uintx4*>(&s_x[t][dx]);",
    "          const uintx4 X0 = WG_EXP == 4 ? (uintx4){(unsigned)dx, 5u, 6u, 7u} : *reinterpret_cast<const uintx4*>(&s_x[t][dx]);")
rep("          const unsigned X1 = (*reinterpret_cast<const uintx4*>(&s_x[t][dx + 4]))[0];",
    "          const unsigned X1 = (WG_EXP == 4 || WG_EXP == 5) ? (unsigned)(dx + t) : (*reinterpret_cast<const uintx4*>(&s_x[t][dx + 4]))[0];")
rep("      const float4 c0 = *reinterpret_cast<const float4*>(&s_ca[col_l * SC_CST]);\n      const float c4 = s_ca[col_l * SC_CST + 4];",
    "      const float4 c0 = WG_EXP == 6 ? make_float4(1.f, 0.5f, 0.25f, 2.f) : *reinterpret_cast<const float4*>(&s_ca[col_l * SC_CST]);\n      const float c4 = WG_EXP == 6 ? 0.1f : s_ca[col_l * SC_CST + 4];")
rep("      const float4 c = *reinterpret_cast<const float4*>(&s_cb[cil * 4]);", "      const float4 c = WG_EXP == 6 ? make_float4(1.f, 0.f, 0.f, 60000.f) : *reinterpret_cast<const float4*>(&s_cb[cil * 4]);")
open("/tmp/conv_bx3_wg.hip", "w").write("#ifndef WG_EXP\n#define WG_EXP 0\n#endif\n" + head + s + tail)
PY
cp conv_sp_pack.h sc_common.h /tmp/
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -fno-slp-vectorize -DWG_EXP=$n -c /tmp/conv_bx3_wg.hip -o /tmp/conv_bx3_wg$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_mfma.o /tmp/conv_bx3_wg$n.o conv_sp.o conv_spw.o conv_pw3.o conv_irt.o conv_valu.o elementwise.o mag1c.o features.o validation.o host_io.o -o ../libstarcop_hip_wg$n.so
done
ls ../libstarcop_hip_wg*.so
