#!/bin/bash
# "Half bytes" elimination build: the kernels of the full-resolution group (planes of >= 256^2 pixels: decoder.blocks.3 /
# .4, head, BatchNorm-backward reductions, the 3x3 weight gradients there) address column x >> 1 instead of x when they load or store
# an ACTIVATION or GRADIENT tensor -- two neighbouring lanes hit the same dword, a wave instruction touches HALF the sectors, a tensor
# pass moves half the bytes through L2 / HBM -- with the instruction stream, the tiles and the launch geometry unchanged.  Results are
# wrong by construction; timings bound what 2-byte storage could buy those kernels WITHOUT restructuring them (packed two-pixel lanes
# would also halve the load instructions: a separate, larger item).   -> starcop_amd/libstarcop_hip_half.so (STARCOP_HIP_LIB=...)
# Not covered: the sub-pixel kernels of decoder.blocks.3.conv1 (k_conv3_sp / k_conv3_spd) and the stem.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/starcop_amd/csrc"
make -s
python3 - <<'PY'
import re
HDR = "#define HSX(x, hw) ((hw) >= 65536 ? ((x) >> 1) : (x))\n"
def patch(src, dst, reps):
    s = open(src).read()
    for a, b, cnt in reps:
        assert s.count(a) == cnt, (src, a, s.count(a))
        s = s.replace(a, b)
    open(dst, "w").write(HDR + s)
# ---- conv_bx3.hip: k_conv3_bx3 / k_conv3_ws loads + stores, k_conv3_thin_h loads + stores, k_wgrad3_bx3 / k_wgrad_thin_h loads
patch("conv_bx3.hip", "/tmp/conv_bx3_half.hip", [
    ("      off0[r] = ok ? (unsigned)((y >> up0) * Ws0 + (x >> up0)) : 0u;\n      off1[r] = ok ? (unsigned)((y >> up1) * Ws1 + (x >> up1)) : 0u;",
     "      off0[r] = ok ? (unsigned)((y >> up0) * Ws0 + HSX(x >> up0, (H >> up0) * Ws0)) : 0u;\n      off1[r] = ok ? (unsigned)((y >> up1) * Ws1 + HSX(x >> up1, (H >> up1) * Ws1)) : 0u;", 2),
    ("loff[pp] = (unsigned)(4 * lhi) * hw32 + (unsigned)(okp[pp] ? oy * W + ox : 0);",
     "loff[pp] = (unsigned)(4 * lhi) * hw32 + (unsigned)(okp[pp] ? oy * W + HSX(ox, H * W) : 0);", 3),
    ("          const size_t opix = (size_t)oy * W + ox;", "          const size_t opix = (size_t)oy * W + HSX(ox, H * W);", 2),
    # thin forward / data gradient
    ("    const unsigned off = ok ? (unsigned)((y >> up) * Ws + (x >> up)) : 0u;", "    const unsigned off = ok ? (unsigned)((y >> up) * Ws + ((x >> up) >> 1)) : 0u;", 1),
    ("      if (ok) *reinterpret_cast<float*>(reinterpret_cast<char*>(outn) + ((unsigned)co * HWu + (unsigned)(oy * W + ox)) * 4u) = v;",
     "      if (ok) *reinterpret_cast<float*>(reinterpret_cast<char*>(outn) + ((unsigned)co * HWu + (unsigned)(oy * W + (ox >> 1))) * 4u) = v;", 2),
    # 3x3 weight gradients: gradient rows (8-byte loads stay 8-byte aligned) and input rows
    ("        const unsigned off = dyo[k] + (unsigned)(((y >= 0 && y < H) ? y : 0) * W + ((x < W) ? x : 0)) * 4u;",
     "        const unsigned off = dyo[k] + (unsigned)(((y >= 0 && y < H) ? y : 0) * W + (HSX((x < W) ? x : 0, H * W) & ~1)) * 4u;", 1),
    ("    const unsigned off = dyo[k] + (unsigned)(((y >= 0 && y < H) ? y : 0) * W + ((x < W) ? x : 0)) * 4u;",
     "    const unsigned off = dyo[k] + (unsigned)(((y >= 0 && y < H) ? y : 0) * W + ((((x < W) ? x : 0) >> 1) & ~1)) * 4u;", 1),
    ("        xr[k][0] = *reinterpret_cast<const float*>(xch[k] + (ro + (unsigned)(xa >> up) * 4u));\n        xr[k][1] = *reinterpret_cast<const float*>(xch[k] + (ro + (unsigned)(xb >> up) * 4u));",
     "        xr[k][0] = *reinterpret_cast<const float*>(xch[k] + (ro + (unsigned)HSX(xa >> up, (H >> up) * (W >> up)) * 4u));\n        xr[k][1] = *reinterpret_cast<const float*>(xch[k] + (ro + (unsigned)HSX(xb >> up, (H >> up) * (W >> up)) * 4u));", 1),
    ("    xr[k][0] = *reinterpret_cast<const float*>(base + (ro + (unsigned)(xa >> up) * 4u));\n    xr[k][1] = *reinterpret_cast<const float*>(base + (ro + (unsigned)(xb >> up) * 4u));",
     "    xr[k][0] = *reinterpret_cast<const float*>(base + (ro + (unsigned)HSX(xa >> up, (H >> up) * Ws) * 4u));\n    xr[k][1] = *reinterpret_cast<const float*>(base + (ro + (unsigned)HSX(xb >> up, (H >> up) * Ws) * 4u));", 1),
])
# ---- conv_valu.hip: head forward (input rows), head backward (input rows, gradient stores)
patch("conv_valu.hip", "/tmp/conv_valu_half.hip", [
    ("          va[u] = row[oka ? xa : 0];\n          vb[u] = row[okb ? xb : 0];", "          va[u] = row[(oka ? xa : 0) >> 1];\n          vb[u] = row[(okb ? xb : 0) >> 1];", 1),
    ("    const float* xb = in.x + ((size_t)n * CIN + c0) * HW + (xok ? x : 0);\n    float* gb = gin + ((size_t)n * CIN + c0) * HW + x;",
     "    const float* xb = in.x + ((size_t)n * CIN + c0) * HW + ((xok ? x : 0) >> 1);\n    float* gb = gin + ((size_t)n * CIN + c0) * HW + (x >> 1);", 1),
])
# ---- elementwise.hip: the streaming BatchNorm-backward reduction
patch("elementwise.hip", "/tmp/elementwise_half.hip", [
    ("      const float4 yv = *reinterpret_cast<const float4*>(y + base + i);\n      const float4 gv = *reinterpret_cast<const float4*>(g + base + i);\n      const float ya[4] = {yv.x, yv.y, yv.z, yv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};\n#pragma unroll\n      for (int k = 0; k < 4; ++k) {\n        const float yh = fmaf(ya[k], scale, shift);\n        const float gb = (yh > lo && yh < hi) ? ga[k] : 0.f;\n        s1 += gb;\n        s2 = fmaf(gb, (ya[k] - mean) * invstd, s2);\n        mx = fmaxf(mx, fabsf(gb));\n        ax = fmaxf(ax, fabsf(yh));",
     "      const int ih = HW >= 65536 ? ((i >> 1) & ~3) : i;\n      const float4 yv = *reinterpret_cast<const float4*>(y + base + ih);\n      const float4 gv = *reinterpret_cast<const float4*>(g + base + ih);\n      const float ya[4] = {yv.x, yv.y, yv.z, yv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};\n#pragma unroll\n      for (int k = 0; k < 4; ++k) {\n        const float yh = fmaf(ya[k], scale, shift);\n        const float gb = (yh > lo && yh < hi) ? ga[k] : 0.f;\n        s1 += gb;\n        s2 = fmaf(gb, (ya[k] - mean) * invstd, s2);\n        mx = fmaxf(mx, fabsf(gb));\n        ax = fmaxf(ax, fabsf(yh));", 1),
])
PY
cp sc_common.h conv_sp_pack.h /tmp/
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result"
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c /tmp/conv_bx3_half.hip -o /tmp/conv_bx3_half.o &
/opt/rocm/bin/hipcc $F -c /tmp/conv_valu_half.hip -o /tmp/conv_valu_half.o &
/opt/rocm/bin/hipcc $F -c /tmp/elementwise_half.hip -o /tmp/elementwise_half.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_mfma.o /tmp/conv_bx3_half.o conv_sp.o conv_spw.o conv_pw3.o conv_irt.o /tmp/conv_valu_half.o /tmp/elementwise_half.o mag1c.o features.o validation.o host_io.o -o ../libstarcop_hip_half.so
ls -la ../libstarcop_hip_half.so
