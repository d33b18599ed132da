// Decoder conv1 of smp.Unet as a SUB-PIXEL convolution (starcop/models/model_module.py:244-251: every DecoderBlock runs
// conv3x3(cat([F.interpolate(prev, scale_factor=2, mode="nearest"), skip]))).
//
// For the up-sampled channels, conv3x3(nearest_up2(x)) is exactly four phase-specific 2x2 convolutions on the LOW-resolution x:
//   y[co][2i+py][2j+px] = sum_{a,b in {0,1}} sum_ci  Wph[py][px][a][b][co][ci] * x[ci][i + a-1+py][j + b-1+px]
//   Wph[py][px][a][b] = sum_{kh in S(py,a)} sum_{kw in S(px,b)} w[kh][kw],   S(0,0)={0}  S(0,1)={1,2}  S(1,0)={0,1}  S(1,1)={2}
// (zero padding maps 1:1: an up-sampled pixel outside the image is a low-resolution pixel outside the image).  The 3x3 form
// (k_conv3_bx3 with src.up address arithmetic) loads, prologues and splits every low-resolution value once per high-resolution
// copy and multiplies nine taps where four suffice: 2.25x the MFMAs and, per output, 2-4x the staging of this kernel.
// The skip channels (full resolution) join the SAME launch as low-resolution "parity planes" (qy,qx) -- pixels (2i+qy, 2j+qx) of a
// skip channel: slot (py,px,a,b) reads a parity plane at the same offset (a-1+py, b-1+px), with the single tap
// kh = 2a+py+qy-1, kw = 2b+px+qx-1 as its filter (zero when that is outside the 3x3 window), so one code path serves both kinds
// of 16-channel chunk (conv_sp_pack.h).
//
// Arithmetic: the two-fp16-term split of conv_bx3.hip (a*s = h0 + h1, products h0*g1 + h1*g0 + h0*g0, fp32 accumulation, exact
// power-of-two operand scales divided out in the epilogue).  The phase filters are sums of up to four taps, so their scale is
// 2^6 instead of 2^8 (|w| < 255 keeps |sum| * 2^6 < 65504).
//
// GEMM view per phase (py,px) and tap (a,b):  D[co][pixel] += Wph[co][ci] * patch[ci][pixel + (a-1+py, b-1+px)],  K step = 16 channels
//   A (32 x 16): lane l -> Wph[co = l&31][ci = 8*(l>>5) .. +7]   (one 16-byte LDS read per term)
//   B (16 x 32): lane l -> patch[ci = 8*(l>>5) .. +7][pixel l&31]   (one 16-byte LDS read per term)
// Work-group = 8 waves; low-resolution tile = 8 groups of 32 pixels (TW = 32: 8 rows x 32 columns; TW = 16: 16 rows x 16 columns,
// a group = two rows) x 32 output channels x 4 phases = 16 x 64 (32 x 32) output pixels.  Wave w: phase ROW py = w >> 2, groups
// 2(w&3), 2(w&3)+1, both px: 4 accumulators -- which leaves the registers to request every operand one 12-MFMA step ahead (the
// first version, 4 waves x 8 accumulators, had none and waited for LDS after every second MFMA).  The two waves of a SIMD share
// ONE patch and one filter stage in LDS: every low-resolution value is staged once for all four phases.
// Pipeline: patch and filters double-buffered, ONE barrier per 16-channel chunk; chunk k+1 is converted / stored at the top of
// chunk k (its global loads were issued a whole chunk earlier) and the loads of chunk k+2 are re-issued into the same registers
// right behind ("refill": DESIGN.md section 11 rule 2); all loads unconditional (clamped chunk index), so the waits stay exact.
#include "sc_common.h"
#include "conv_sp_pack.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) unsigned int uintx4;
typedef __attribute__((ext_vector_type(8))) _Float16 halfx8;
typedef __attribute__((ext_vector_type(2))) _Float16 halfx2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr float SP_SX = 2.f, SP_HMAX = 65504.f;
#ifndef SP_INTERLEAVE
#define SP_INTERLEAVE 4
#endif

__device__ __forceinline__ void sp_split2h(float a, float b, unsigned& t0, unsigned& t1) {
  const floatx2 v = {a, b};
  const halfx2 h0 = __builtin_convertvector(v, halfx2);
  t0 = __builtin_bit_cast(unsigned, h0);
  float ra, rb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(t0), "v"(v[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(t0), "v"(v[1]));
  const halfx2 h1 = __builtin_convertvector(floatx2{ra, rb}, halfx2);
  t1 = __builtin_bit_cast(unsigned, h1);
}

// activation operand scale from the sources' bounds (h_act_scale of conv_bx3.hip: the default 2 unless a bound says 2 M > 32752)
__device__ __forceinline__ float sp_act_scale(const float* xb0, const float* xb1) {
  float M = fmaxf(xb0 ? *xb0 : 0.f, xb1 ? *xb1 : 0.f);
  if (!(M * SP_SX > 32752.f)) return SP_SX;
  M = fminf(M, 3.0e38f);
  int e;
  (void)frexpf(32752.f / M, &e);
  e = e - 1 < -120 ? -120 : e - 1;
  return ldexpf(1.f, e);
}

struct ConvSPP {
  SrcD s0, s1;             // s0: low-resolution source [N][C0][Hl][Wl]; s1: skip source [N][C1][2 Hl][2 Wl] (C1 = 0: none); AFFINE or RAW
  const uintx4* wpk;       // sc_pack_weights_sp layout
  int N, Hl, Wl, Cout;     // output is [N][Cout][2 Hl][2 Wl]
  float* out;
  float* stats;            // [rows = N * tiles][Cout][2] or NULL
  const float* xb0; const float* xb1;      // activation bounds of the two sources (device floats or NULL)
};

constexpr int SP_SMEM_W = 2 * SP_WST * 16;               // filter stages, double-buffered (64 KB)
// patch pitch in 16-byte entries: TW + 2 pixels, padded for the 16-wide tile (SP_PC16) -- its operand reads take two 16-entry row pieces
// per half-wave, which collide in the LDS banks at the natural pitch 18 (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.35)
#ifndef SP_PC16
#define SP_PC16 18
#endif
template <int TW> constexpr int sp_pc() { return TW == 16 ? SP_PC16 : TW + 2; }
template <int TW> constexpr int sp_npx() { return (256 / TW + 2) * sp_pc<TW>(); }
template <int TW> constexpr int sp_smem_bytes() { return SP_SMEM_W + 2 * 2 * 2 * sp_npx<TW>() * 16 + 8 * 32 * 2 * 4; }

// BF: the "bf16" precision mode -- ONE bf16 term per operand (round to nearest even, no range scale), a third of the MFMAs and half
// the operand reads; the skeleton, LDS layout (term 0 only) and pipeline are the same
// NPP: pixel groups per wave.  2: the 256-pixel tile above.  1: a 128-pixel tile (8 x 16 or 4 x 32), one group per wave -- for launches
// whose 256-pixel tiles do not fill the chip (decoder.blocks.0 at batch 16: 16 planes of 16 x 16 = 128 work-groups for 256 CUs); twice
// the work-groups, each with half the MFMAs per staged filter chunk.
template <int TW, bool BF, int NPP = 2>
__global__ __launch_bounds__(512, 2) void k_conv3_sp(const ConvSPP p) {
  static_assert(NPP == 1 || NPP == 2, "one or two pixel groups per wave");
  constexpr int NT = BF ? 1 : 2;
  constexpr int NM = BF ? 4 : 12;            // MFMAs per step
  constexpr int TH = 128 * NPP / TW;
  constexpr int PC = sp_pc<TW>(), NPX = sp_npx<TW>();      // (NPX: the buffer pitch, sized for the two-group tile)
  constexpr int NPXU = (TH + 2) * PC;        // patch entries of this tile
  constexpr int NR = (NPXU + 127) / 128;
  constexpr int WST = SP_WST / 2 * NT;       // 16-byte filter entries per chunk
  constexpr int NWV = WST / 512;
  static_assert(NPXU <= NPX && NPXU <= 128 * NR, "staging rounds of 128 threads per channel quarter");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uintx4* const s_w = reinterpret_cast<uintx4*>(smem);                                   // [2][WST]
  uintx4* const s_p = reinterpret_cast<uintx4*>(smem + SP_SMEM_W);                        // [2 buf][NT][2 halves][NPX]
  float* const s_red = reinterpret_cast<float*>(smem + SP_SMEM_W + 2 * NT * 2 * NPX * 16);  // [8][32][2]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int py = wave >> 2, wq = wave & 3;
  const int Hl = p.Hl, Wl = p.Wl;
  const int tiles_x = (Wl + TW - 1) / TW, tiles_y = (Hl + TH - 1) / TH;
  const int ncot = (p.Cout + 31) >> 5;
  int n, cot, tile;
  {
    // each XCD (work-groups are dealt to the 8 XCDs round-robin by linear id) walks a contiguous eighth of the pixel tiles, the
    // cout tiles of a pixel tile back to back: neighbouring patches and the re-read patch are hits in that XCD's L2
    const int per_img = tiles_x * tiles_y;
    const int total = per_img * p.N, per_xcd = (total + 7) >> 3;
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int j = slot / ncot;
    const int pt = xcd * per_xcd + j;
    if (j >= per_xcd || pt >= total) return;
    cot = slot - j * ncot;
    n = pt / per_img; tile = pt - n * per_img;
  }
  n = __builtin_amdgcn_readfirstlane(n); cot = __builtin_amdgcn_readfirstlane(cot); tile = __builtin_amdgcn_readfirstlane(tile);
  const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int C0 = p.s0.C, C1 = p.s1.C;
  const int nku = (C0 + 15) >> 4, nkt = nku + 4 * ((C1 + 15) >> 4);
  const uintx4* wbase = p.wpk + (size_t)cot * nkt * WST;

  const float hsx = BF ? 1.f : sp_act_scale(p.xb0, p.xb1);
  const float hinv = BF ? 1.f : 1.f / (hsx * SP_SW);

  floatx16 acc[NPP][2];    // [group pp][px]
#pragma unroll
  for (int i = 0; i < 2 * NPP; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;

  // ---- staging state: thread = (channel quarter q4: half hw, 4-channel sub-block), patch pixels sidx + 128 r ----
  const int q4 = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int hw = q4 >> 1, sub = q4 & 1;
  const int sidx = tid & 127;
  unsigned pyx[NR];        // patch pixel of round r: (y << 16) | x of the low-resolution plane, 0xFFFFFFFF outside the image / the patch
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = sidx + 128 * r;
    const int pr = e / PC, pc = e - pr * PC;
    const int y = y0 - 1 + pr, x = x0 - 1 + pc;
    const bool ok = (e < NPXU) && (y >= 0) && (y < Hl) && (x >= 0) && (x < Wl);
    pyx[r] = ok ? ((unsigned)y << 16) | (unsigned)x : 0xFFFFFFFFu;
  }
  float xv[NR][4];
  uintx4 wv[NWV];
  float cs0[4], cs1[4];
  float slo = 0.f, shi = 0.f;
  int nch = 0;

  // requests chunk kc: its filters first (vmcnt retires in order), the patch values, the per-channel constants
  auto request = [&](int kc) __attribute__((always_inline)) {
    const uintx4* wsrc = wbase + (size_t)kc * WST;
#pragma unroll
    for (int j = 0; j < NWV; ++j) wv[j] = wsrc[tid + 512 * j];
    const bool second = kc >= nku;
    const int c4 = kc - nku;
    const SrcD& s = second ? p.s1 : p.s0;
    const int cbase = (second ? (c4 >> 2) : kc) * 16 + hw * 8 + sub * 4;
    const size_t plane = second ? (size_t)4 * Hl * Wl : (size_t)Hl * Wl;
    const unsigned qoff = second ? (unsigned)(((c4 >> 1) & 1) * 2 * Wl + (c4 & 1)) : 0u;
    nch = s.C - cbase;
    const int cb0 = nch > 0 ? cbase : 0;
    const float* xb = s.x + ((size_t)n * s.C + cb0) * plane;
    const int jmax = nch > 0 ? nch - 1 : 0;      // (channels beyond the last re-read it; masked at conversion)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const bool ok = pyx[r] != 0xFFFFFFFFu;
      const unsigned y = pyx[r] >> 16, x = pyx[r] & 0xFFFFu;
      const unsigned o = !ok ? 0u : (second ? (4u * y * (unsigned)Wl + 2u * x + qoff) : (y * (unsigned)Wl + x));
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[r][j] = xb[(size_t)(j < jmax ? j : jmax) * plane + o];
    }
    if constexpr (BF) { slo = sc_act_lo(s.act); shi = sc_act_hi(s.act); }
    else { slo = fmaxf(sc_act_lo(s.act) * hsx, -SP_HMAX); shi = fminf(sc_act_hi(s.act) * hsx, SP_HMAX); }
    const float* cb = s.cst + (size_t)cb0 * SC_CST;      // (RAW sources read the host's identity table: no load under a branch)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float2 c = *reinterpret_cast<const float2*>(cb + (size_t)(j < jmax ? j : jmax) * SC_CST);
      cs0[j] = c.x * hsx; cs1[j] = c.y * hsx;
    }
  };
  // prologue + split of round r of the requested chunk -> patch buffer `buf`; its filters -> filter buffer `buf`
  auto stage_round = [&](int buf, int r) __attribute__((always_inline)) {
    uint2* const sp2 = reinterpret_cast<uint2*>(s_p + (size_t)buf * NT * 2 * NPX);
    uint2 t0, t1;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * jp + h;
        const float t = __builtin_amdgcn_fmed3f(fmaf(xv[r][j], cs0[j], cs1[j]), slo, shi);
        v[h] = (pyx[r] != 0xFFFFFFFFu && j < nch) ? t : 0.f;
      }
      unsigned a0, a1 = 0u;
      if constexpr (BF) a0 = __builtin_bit_cast(unsigned, __builtin_convertvector((floatx2){v[0], v[1]}, bf16x2));
      else sp_split2h(v[0], v[1], a0, a1);
      if (jp == 0) { t0.x = a0; t1.x = a1; } else { t0.y = a0; t1.y = a1; }
    }
    const int e = sidx + 128 * r;
    if (e < NPXU) {
      sp2[((0 * 2 + hw) * NPX + e) * 2 + sub] = t0;
      if constexpr (!BF) sp2[((1 * 2 + hw) * NPX + e) * 2 + sub] = t1;
    }
  };
  auto stage_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NWV; ++j) s_w[buf * WST + tid + 512 * j] = wv[j];
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < NR; ++r) stage_round(buf, r);
    stage_w(buf);
  };

  // lane -> patch entry of its pixel in group pp (centre tap)
  int eb[NPP];
#pragma unroll
  for (int pp = 0; pp < NPP; ++pp) {
    const int g = NPP * wq + pp;
    const int rowt = TW == 32 ? g : 2 * g + (l31 >> 4), colt = TW == 32 ? l31 : (l31 & 15);
    eb[pp] = (rowt + 1 + py - 1) * PC + colt;           // + a * PC + o   (source row offset a - 1 + py, o = column offset + 1 in 0..2)
  }
  const int wlane = py * (NT * 2 * 2 * 2 * 2 * 32) + lhi * 32 + l31;       // + ((((c*2 + px)*2 + a)*2 + b)*2)*32

  // the 48 MFMAs of one chunk: steps (a, pp) of 12 MFMAs, every operand requested one step before its use.  TW = 32: groups are
  // consecutive patch rows, so step (a = 1, group 0) and step (a = 0, group 1) read the SAME patch row: three row operand sets
  // instead of four (34 instead of 40 LDS reads per chunk and wave).
  // (Parity chunks run all 16 slots although 7 of them have zero filters: compile-time slot masks per parity -- 7 variants of this
  // body -- pushed the kernel from 239 registers to 256 + scratch reloads inside the loop for < 0.5 % of the step: not kept.)
  auto compute = [&](int buf, int knext) __attribute__((always_inline)) {
    const uintx4* const sw = s_w + buf * WST + wlane;
    const uintx4* const sp = s_p + (size_t)buf * NT * 2 * NPX + lhi * NPX;
    halfx8 A0[2][2][NT], A1[2][2][NT], B0[3][NT], B1[3][NT];
    auto load_A = [&](halfx8 (&A)[2][2][NT], int a) {
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < NT; ++c) A[px][b][c] = __builtin_bit_cast(halfx8, sw[((((c * 2 + px) * 2 + a) * 2 + b) * 2) * 32]);
    };
    auto load_B = [&](halfx8 (&B)[3][NT], int a, int pp) {
#pragma unroll
      for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int c = 0; c < NT; ++c) B[o][c] = __builtin_bit_cast(halfx8, sp[c * 2 * NPX + eb[pp] + a * PC + o]);
    };
    auto mfmas = [&](const halfx8 (&A)[2][2][NT], const halfx8 (&B)[3][NT], auto ppc) {
      constexpr int pp = decltype(ppc)::value;
#pragma unroll
      for (int t = 0; t < (BF ? 1 : 3); ++t)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int px = 0; px < 2; ++px) {      // the two accumulators alternate: no back-to-back dependent MFMAs
            const int o = b + px;
            if constexpr (BF)
              acc[pp][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[px][b][0]), __builtin_bit_cast(bf16x8, B[o][0]), acc[pp][px], 0, 0, 0);
            else
              acc[pp][px] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[px][b][t == 1 ? NT - 1 : 0], B[o][t == 0 ? NT - 1 : 0], acc[pp][px], 0, 0, 0);
          }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // one step: 12 MFMAs with a slice of the NEXT chunk's staging (VALU / LDS stores / the refill's loads) issued in their shadows:
    // the matrix pipe runs an MFMA for 32 cycles while the wave may issue ~5 independent instructions (MI355X_MICROARCH.md); as
    // separate phases the two cost their sum (elimination builds, decoder.blocks.2.conv1 at batch 16: 116 us, without the MFMAs 72,
    // without prologue / split / patch stores 86 -- tools/build_exp_sp.sh)
    auto step = [&](const halfx8 (&A)[2][2][NT], const halfx8 (&B)[3][NT], auto ppc, auto hook) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      hook();
      mfmas(A, B, ppc);
#if SP_INTERLEAVE
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, SP_INTERLEAVE * (12 / NM), 0);      // then a few VALU
        __builtin_amdgcn_sched_group_barrier(0x200, 12 / NM, 0);      // and at most one LDS store (BF: three)
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    };
    const int nb = buf ^ 1;
    if constexpr (NPP == 1) {       // one group: two steps (a = 0, 1) of 12 MFMAs; two staging rounds
      static_assert(NPP == 2 || NR == 2, "two staging rounds");
      load_A(A0, 0); load_B(B0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_A(A1, 1); load_B(B1, 1, 0);
      step(A0, B0, P0{}, [&]() __attribute__((always_inline)) { stage_round(nb, 0); });
      step(A1, B1, P0{}, [&]() __attribute__((always_inline)) { stage_round(nb, 1); stage_w(nb); request(knext); });
    } else {
      load_A(A0, 0); load_B(B0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_B(B1, 0, 1);
      step(A0, B0, P0{}, [&]() __attribute__((always_inline)) { stage_round(nb, 0); });
      if constexpr (TW == 32) {
        load_A(A1, 1); load_B(B0, 1, 1);
        step(A0, B1, P1{}, [&]() __attribute__((always_inline)) { stage_round(nb, 1); });
        step(A1, B1, P0{}, [&]() __attribute__((always_inline)) { stage_round(nb, 2); stage_w(nb); });      // (a = 1, group 0) reads the patch row of (a = 0, group 1)
        step(A1, B0, P1{}, [&]() __attribute__((always_inline)) { request(knext); });
      } else {
        load_A(A1, 1); load_B(B0, 1, 0);
        step(A0, B1, P1{}, [&]() __attribute__((always_inline)) { stage_round(nb, 1); });
        load_B(B1, 1, 1);
        step(A1, B0, P0{}, [&]() __attribute__((always_inline)) { stage_round(nb, 2); stage_w(nb); });
        step(A1, B1, P1{}, [&]() __attribute__((always_inline)) { request(knext); });
      }
    }
  };

  // ---- pipeline ----
  // chunk kc: its MFMAs from buffers kc & 1, with chunk kc+1 (requested one chunk ago) converted / stored into the other buffers and
  // chunk kc+2 requested into the same registers, all inside the MFMA steps (the last chunk restages / re-requests itself: unused)
  request(0);
  stage(0);
  request(nkt > 1 ? 1 : 0);
  __syncthreads();
  for (int kc = 0; kc < nkt; ++kc) {
    compute(kc & 1, kc + 2 < nkt ? kc + 2 : nkt - 1);
    __syncthreads();
  }

  // ---- epilogue: two adjacent output pixels (px = 0, 1) per lane and row -> 8-byte stores, 256-byte runs per (cout, row) ----
  const int W = 2 * Wl;
  const unsigned hw32 = (unsigned)((size_t)4 * Hl * Wl);
  float* const ob = p.out + ((size_t)n * p.Cout + cot * 32) * (size_t)hw32;
  const bool want_stats = p.stats != nullptr;
  unsigned loff[NPP]; bool okp[NPP];
#pragma unroll
  for (int pp = 0; pp < NPP; ++pp) {
    const int g = NPP * wq + pp;
    const int i = y0 + (TW == 32 ? g : 2 * g + (l31 >> 4)), j = x0 + (TW == 32 ? l31 : (l31 & 15));
    okp[pp] = i < Hl && j < Wl;
    loff[pp] = (unsigned)(4 * lhi) * hw32 + (unsigned)(okp[pp] ? (2 * i + py) * W + 2 * j : 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int cu = (r & 3) + 8 * (r >> 2);
    const int col = cu + 4 * lhi;
    const bool okc = cot * 32 + col < p.Cout;
    float sv = 0.f, sq = 0.f;
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp) {
      floatx2 v = {acc[pp][0][r] * hinv, acc[pp][1][r] * hinv};
      if (okp[pp] && okc) *reinterpret_cast<floatx2*>(ob + loff[pp] + (unsigned)cu * hw32) = v;
      else v = floatx2{0.f, 0.f};
      sv += v[0] + v[1];
      sq = fmaf(v[0], v[0], fmaf(v[1], v[1], sq));
    }
    if (want_stats) {
      const float s = half_sum32(sv);
      const float ss = half_sum32(sq);
      if (l31 == SC_HALF_SUM_LANE) { s_red[(wave * 32 + col) * 2 + 0] = s; s_red[(wave * 32 + col) * 2 + 1] = ss; }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < 64) {
      const int col = tid >> 1, k = tid & 1;
      const int co = cot * 32 + col;
      if (co < p.Cout) {
        float t = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) t += s_red[(w8 * 32 + col) * 2 + k];
        const size_t row = (size_t)n * (tiles_x * tiles_y) + tile;
        p.stats[(row * p.Cout + co) * 2 + k] = t;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Data gradient of the same layer w.r.t. the LOW-resolution source (the up-sampled channels): the four parity planes of dy are
// the K chunks, every one with 2 x 2 taps (conv_sp_pack.h); the work-group skeleton, LDS layout and pipeline are the forward
// kernel's with   phase row py -> wave half h = cin block pair,  px -> cin block mx,  and the source offset of tap (a, b) taken from
// the chunk's parity: (1 - a - qy, 1 - b - qx).  Tile = 8 groups of 32 low-resolution pixels x 128 input channels; the source is the
// BatchNorm / activation-backward operand (g, y -> A g' + B y + D) with the gradient range scale of conv_bx3.hip (absmax).
struct ConvSPD {
  SrcD dy;                 // SC_SRC_BNBWD source [N][Co][2 Hl][2 Wl]
  const uintx4* wpk;       // spd pack
  int N, Hl, Wl, Cup;      // output [N][Cup][Hl][Wl]
  float* out;
  int accum;
  const float* absmax;
  float* out_skip;         // [N][Cskip][2 Hl][2 Wl] gradient of the skip channels, or NULL.  skip_mode 1 (vskip: Cup <= 64, Cskip <= 16): from
  int Cskip, accum_skip;   // wave half 1 of the one channel tile; skip_mode 2 (skip tiles): from the channel tiles cot >= ntu (conv_sp_pack.h)
  int skip_mode, ntu, nts; // ntu: tiles of 128 up-sampled channels; nts: skip tiles (mode 2) of 32 skip channels x 4 output parities
};

__device__ __forceinline__ float spd_grad_scale(const float* absmax) {      // (h_grad_scale of conv_bx3.hip)
  const float M = absmax ? *absmax : 0.f;
  if (!(M > 0.f) || !(M < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(M, &e);
  e = 5 - e;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.f, e);
}

template <int TW, bool BF, int NPP = 2>
__global__ __launch_bounds__(512, 2) void k_conv3_spd(const ConvSPD p) {
  static_assert(NPP == 1 || NPP == 2, "one or two pixel groups per wave (see k_conv3_sp)");
  constexpr int NT = BF ? 1 : 2;
  constexpr int NM = BF ? 4 : 12;
  constexpr int TH = 128 * NPP / TW;
  constexpr int PC = sp_pc<TW>(), NPX = sp_npx<TW>();
  constexpr int NPXU = (TH + 2) * PC;
  constexpr int NR = (NPXU + 127) / 128;
  static_assert(NPXU <= NPX && (NPP == 2 || NR == 2), "staging rounds");
  constexpr int WST = SP_WST / 2 * NT;
  constexpr int NWV = WST / 512;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uintx4* const s_w = reinterpret_cast<uintx4*>(smem);                                   // [2][WST]
  uintx4* const s_p = reinterpret_cast<uintx4*>(smem + SP_SMEM_W);                        // [2 buf][NT][2 halves][NPX]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int hh = wave >> 2, wq = wave & 3;
  const int Hl = p.Hl, Wl = p.Wl;
  const int tiles_x = (Wl + TW - 1) / TW, tiles_y = (Hl + TH - 1) / TH;
  const int ncot = p.ntu + p.nts;
  int n, cot, tile;
  {
    const int per_img = tiles_x * tiles_y;
    const int total = per_img * p.N, per_xcd = (total + 7) >> 3;
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int j = slot / ncot;
    const int pt = xcd * per_xcd + j;
    if (j >= per_xcd || pt >= total) return;
    cot = slot - j * ncot;
    n = pt / per_img; tile = pt - n * per_img;
  }
  n = __builtin_amdgcn_readfirstlane(n); cot = __builtin_amdgcn_readfirstlane(cot); tile = __builtin_amdgcn_readfirstlane(tile);
  const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int Co = p.dy.C;
  const int nkt = 4 * ((Co + 15) >> 4);
  const uintx4* wbase = p.wpk + (size_t)cot * nkt * WST;

  const float hsx = BF ? 1.f : spd_grad_scale(p.absmax);
  const float hinv = BF ? 1.f : 1.f / (hsx * SP_SW);

  floatx16 acc[NPP][2];    // [group pp][cin block mx]
#pragma unroll
  for (int i = 0; i < 2 * NPP; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;

  const int q4 = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int hw = q4 >> 1, sub = q4 & 1;
  const int sidx = tid & 127;
  unsigned pyx[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = sidx + 128 * r;
    const int pr = e / PC, pc = e - pr * PC;
    const int y = y0 - 1 + pr, x = x0 - 1 + pc;
    const bool ok = (e < NPXU) && (y >= 0) && (y < Hl) && (x >= 0) && (x < Wl);
    pyx[r] = ok ? ((unsigned)y << 16) | (unsigned)x : 0xFFFFFFFFu;
  }
  float xv[NR][4], av[NR][4];
  uintx4 wv[NWV];
  float cs0[4], cs1[4], cs2[4], cs3[4], cs4[4];
  const float slo = sc_act_lo(p.dy.act), shi = sc_act_hi(p.dy.act);
  int nch = 0;
  const size_t plane = (size_t)4 * Hl * Wl;

  auto request = [&](int kc) __attribute__((always_inline)) {
    const uintx4* wsrc = wbase + (size_t)kc * WST;
#pragma unroll
    for (int j = 0; j < NWV; ++j) wv[j] = wsrc[tid + 512 * j];
    const int cbase = (kc >> 2) * 16 + hw * 8 + sub * 4;
    const unsigned qoff = (unsigned)(((kc >> 1) & 1) * 2 * Wl + (kc & 1));
    nch = Co - cbase;
    const int cb0 = nch > 0 ? cbase : 0;
    const size_t o0 = ((size_t)n * Co + cb0) * plane;
    const float* xb = p.dy.x + o0;
    const float* ab = p.dy.aux + o0;
    const int jmax = nch > 0 ? nch - 1 : 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const bool ok = pyx[r] != 0xFFFFFFFFu;
      const unsigned y = pyx[r] >> 16, x = pyx[r] & 0xFFFFu;
      const unsigned o = !ok ? 0u : (4u * y * (unsigned)Wl + 2u * x + qoff);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t cj = (size_t)(j < jmax ? j : jmax) * plane;
        xv[r][j] = xb[cj + o];
        av[r][j] = ab[cj + o];
      }
    }
    const float* cb = p.dy.cst + (size_t)cb0 * SC_CST;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* cj = cb + (size_t)(j < jmax ? j : jmax) * SC_CST;
      const float4 c = *reinterpret_cast<const float4*>(cj);
      cs0[j] = c.x; cs1[j] = c.y; cs2[j] = c.z; cs3[j] = c.w; cs4[j] = cj[4];
    }
  };
  auto stage_round = [&](int buf, int r) __attribute__((always_inline)) {
    uint2* const sp2 = reinterpret_cast<uint2*>(s_p + (size_t)buf * NT * 2 * NPX);
    uint2 t0, t1;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * jp + h;
        const float t = sc_pro_bnbwd(xv[r][j], av[r][j], cs0[j], cs1[j], cs2[j], cs3[j], cs4[j], slo, shi) * hsx;
        v[h] = (pyx[r] != 0xFFFFFFFFu && j < nch) ? (BF ? t : __builtin_amdgcn_fmed3f(t, -SP_HMAX, SP_HMAX)) : 0.f;
      }
      unsigned a0, a1 = 0u;
      if constexpr (BF) a0 = __builtin_bit_cast(unsigned, __builtin_convertvector((floatx2){v[0], v[1]}, bf16x2));
      else sp_split2h(v[0], v[1], a0, a1);
      if (jp == 0) { t0.x = a0; t1.x = a1; } else { t0.y = a0; t1.y = a1; }
    }
    const int e = sidx + 128 * r;
    if (e < NPXU) {
      sp2[((0 * 2 + hw) * NPX + e) * 2 + sub] = t0;
      if constexpr (!BF) sp2[((1 * 2 + hw) * NPX + e) * 2 + sub] = t1;
    }
  };
  auto stage_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NWV; ++j) s_w[buf * WST + tid + 512 * j] = wv[j];
  };

  int eb[NPP];
#pragma unroll
  for (int pp = 0; pp < NPP; ++pp) {
    const int g = NPP * wq + pp;
    const int rowt = TW == 32 ? g : 2 * g + (l31 >> 4), colt = TW == 32 ? l31 : (l31 & 15);
    eb[pp] = (rowt + 1) * PC + colt;           // + (1 - a - qy) * PC + o,  o = 2 - b - qx
  }
  const int wlane = hh * (NT * 2 * 2 * 2 * 2 * 32) + lhi * 32 + l31;

  // chunk kc (parity qy = (kc >> 1) & 1, qx = QX at compile time: it selects the operand registers), 48 MFMAs per wave
  auto compute = [&](int buf, int kc, int knext, auto qxc) __attribute__((always_inline)) {
    constexpr int QX = decltype(qxc)::value;
    const uintx4* const sw = s_w + buf * WST + wlane;
    const uintx4* const sp = s_p + (size_t)buf * NT * 2 * NPX + lhi * NPX + (1 - ((kc >> 1) & 1)) * PC;
    halfx8 A0[2][2][NT], A1[2][2][NT], B0[3][NT], B1[3][NT];
    auto load_A = [&](halfx8 (&A)[2][2][NT], int a) {
#pragma unroll
      for (int mx = 0; mx < 2; ++mx)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < NT; ++c) A[mx][b][c] = __builtin_bit_cast(halfx8, sw[((((c * 2 + mx) * 2 + a) * 2 + b) * 2) * 32]);
    };
    auto load_B = [&](halfx8 (&B)[3][NT], int a, int pp) {
#pragma unroll
      for (int o = 1 - QX; o < 3 - QX; ++o)
#pragma unroll
        for (int c = 0; c < NT; ++c) B[o][c] = __builtin_bit_cast(halfx8, sp[c * 2 * NPX + eb[pp] - a * PC + o]);
    };
    auto mfmas = [&](const halfx8 (&A)[2][2][NT], const halfx8 (&B)[3][NT], auto ppc) {
      constexpr int pp = decltype(ppc)::value;
#pragma unroll
      for (int t = 0; t < (BF ? 1 : 3); ++t)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int mx = 0; mx < 2; ++mx) {
            const int o = 2 - b - QX;
            if constexpr (BF)
              acc[pp][mx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[mx][b][0]), __builtin_bit_cast(bf16x8, B[o][0]), acc[pp][mx], 0, 0, 0);
            else
              acc[pp][mx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[mx][b][t == 1 ? NT - 1 : 0], B[o][t == 0 ? NT - 1 : 0], acc[pp][mx], 0, 0, 0);
          }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto step = [&](const halfx8 (&A)[2][2][NT], const halfx8 (&B)[3][NT], auto ppc, auto hook) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      hook();
      mfmas(A, B, ppc);
#if SP_INTERLEAVE
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (SP_INTERLEAVE + 2) * (12 / NM), 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 12 / NM, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    };
    const int nb = buf ^ 1;
    // rows: (a, pp) reads patch row g_pp + 2 - a - qy; with TW = 32 (g_1 = g_0 + 1) steps (a = 0, group 0) and (a = 1, group 1) share theirs
    load_A(A1, 1); load_B(B0, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_A(A0, 0); load_B(B1, 0, 0);
    step(A1, B0, P0{}, [&]() __attribute__((always_inline)) { stage_round(nb, 0); });
    if constexpr (NPP == 1) {       // one group: the second step carries the rest of the staging
      step(A0, B1, P0{}, [&]() __attribute__((always_inline)) { stage_round(nb, 1); stage_w(nb); request(knext); });
    } else if constexpr (TW == 32) {
      load_B(B0, 0, 1);
      step(A0, B1, P0{}, [&]() __attribute__((always_inline)) { stage_round(nb, 1); });
      step(A1, B1, P1{}, [&]() __attribute__((always_inline)) { stage_round(nb, 2); stage_w(nb); });
      step(A0, B0, P1{}, [&]() __attribute__((always_inline)) { request(knext); });
    } else {
      load_B(B0, 1, 1);
      step(A0, B1, P0{}, [&]() __attribute__((always_inline)) { stage_round(nb, 1); });
      load_B(B1, 0, 1);
      step(A1, B0, P1{}, [&]() __attribute__((always_inline)) { stage_round(nb, 2); stage_w(nb); });
      step(A0, B1, P1{}, [&]() __attribute__((always_inline)) { request(knext); });
    }
  };

  request(0);
#pragma unroll
  for (int r = 0; r < NR; ++r) stage_round(0, r);
  stage_w(0);
  request(1);
  __syncthreads();
  for (int kc = 0; kc < nkt; kc += 2) {          // (nkt is a multiple of 4; the last chunk restages / re-requests itself: unused)
    compute(0, kc, kc + 2 < nkt ? kc + 2 : nkt - 1, std::integral_constant<int, 0>{});
    __syncthreads();
    compute(1, kc + 1, kc + 3 < nkt ? kc + 3 : nkt - 1, std::integral_constant<int, 1>{});
    __syncthreads();
  }

  // ---- epilogue of a skip tile: block (hh, mx) = output parity (oy, ox) of the tile's 32 skip channels -> full resolution
  if (p.skip_mode == 2 && cot >= p.ntu) {
    const unsigned W2 = 2u * (unsigned)Wl, ph32 = (unsigned)((size_t)4 * Hl * Wl);
    float* const os = p.out_skip + (size_t)n * p.Cskip * ph32;
    const int cb = 32 * (cot - p.ntu);
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp) {
      const int g = NPP * wq + pp;
      const int i = y0 + (TW == 32 ? g : 2 * g + (l31 >> 4)), j = x0 + (TW == 32 ? l31 : (l31 & 15));
      const bool okp = i < Hl && j < Wl;
#pragma unroll
      for (int mx = 0; mx < 2; ++mx)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = cb + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (okp && c < p.Cskip) {
            const unsigned off = (unsigned)c * ph32 + (unsigned)(2 * i + hh) * W2 + (unsigned)(2 * j + mx);
            float v = acc[pp][mx][r] * hinv;
            if (p.accum_skip) v += os[off];
            os[off] = v;
          }
        }
    }
    return;
  }
  // ---- epilogue, wave half 1 with virtual skip channels: v = 16 * parity + c -> dskip[c][2i + oy][2j + ox] at full resolution
  if (p.skip_mode == 1 && hh == 1) {
    const unsigned W2 = 2u * (unsigned)Wl, ph32 = (unsigned)((size_t)4 * Hl * Wl);
    float* const os = p.out_skip + (size_t)n * p.Cskip * ph32;
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp) {
      const int g = NPP * wq + pp;
      const int i = y0 + (TW == 32 ? g : 2 * g + (l31 >> 4)), j = x0 + (TW == 32 ? l31 : (l31 & 15));
      const bool okp = i < Hl && j < Wl;
#pragma unroll
      for (int mx = 0; mx < 2; ++mx)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int vch = mx * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, par = vch >> 4, c = vch & 15;
          if (okp && c < p.Cskip) {
            const unsigned off = (unsigned)c * ph32 + (unsigned)(2 * i + (par >> 1)) * W2 + (unsigned)(2 * j + (par & 1));
            float v = acc[pp][mx][r] * hinv;
            if (p.accum_skip) v += os[off];
            os[off] = v;
          }
        }
    }
    return;
  }
  // ---- epilogue: low-resolution stores (the 2x2 sum over the up-sampled copies is in the phase filters)
  const unsigned pl32 = (unsigned)((size_t)Hl * Wl);
  const int cb = cot * 128 + hh * 64;
  float* const ob = p.out + ((size_t)n * p.Cup + cb) * (size_t)pl32;
#pragma unroll
  for (int pp = 0; pp < NPP; ++pp) {
    const int g = NPP * wq + pp;
    const int i = y0 + (TW == 32 ? g : 2 * g + (l31 >> 4)), j = x0 + (TW == 32 ? l31 : (l31 & 15));
    const bool okp = i < Hl && j < Wl;
    const unsigned loff = (unsigned)(4 * lhi) * pl32 + (unsigned)(okp ? i * Wl + j : 0);
#pragma unroll
    for (int mx = 0; mx < 2; ++mx)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = mx * 32 + (r & 3) + 8 * (r >> 2);
        if (okp && cb + cu + 4 * lhi < p.Cup) {
          const unsigned off = loff + (unsigned)cu * pl32;
          float v = acc[pp][mx][r] * hinv;
          if (p.accum) v += ob[off];
          ob[off] = v;
        }
      }
  }
}

__global__ void k_pack_weights_spd(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout, int CinTot, int Cup, int vskip, int bf, size_t total) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < total) spd_pack_item(w, wpk, i, Cout, CinTot, Cup, vskip == 1, bf != 0, vskip == 2 ? CinTot - Cup : 0);
}

__global__ void k_pack_weights_sp(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout, int Cup, int Csk, int bf, size_t total) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < total) sp_pack_item(w, wpk, i, Cout, Cup, Csk, bf != 0);
}

}  // namespace

extern "C" size_t sc_packed_weight_floats_sp(int Cout, int Cup, int Cskip) {
  return (size_t)((Cout + 31) / 32) * sp_chunks(Cup, Cskip) * SP_WST * 4;      // 16-byte entries -> floats
}

extern "C" int sc_pack_weights_sp(const float* w, float* wpk, int Cout, int Cup, int Cskip, int terms, sc_stream stream) {
  SC_REQUIRE(w && wpk && Cout > 0 && Cup > 0 && Cskip >= 0, "sc_pack_weights_sp: bad argument");
  SC_REQUIRE(terms == SC_TERMS_F16X2 || terms == 1, "sc_pack_weights_sp: terms must be SC_TERMS_F16X2 or 1 (one bf16 term)");
  SC_REQUIRE(((uintptr_t)wpk & 15) == 0, "sc_pack_weights_sp: destination must be 16-byte aligned");
  const size_t total = sp_pack_items(Cout, Cup, Cskip);
  hipLaunchKernelGGL(k_pack_weights_sp, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<unsigned short*>(wpk), Cout, Cup, Cskip, terms == 1 ? 1 : 0, total);
  SC_LAUNCH_OK("sc_pack_weights_sp");
  return SC_OK;
}

// pixel groups per wave: 2 (256-pixel tiles) unless the launch's 16-wide tiles x channel tiles give at most 128 work-groups -- half the
// chip or less -- then 1 (128-pixel tiles, twice the work-groups, still one round).  Measured on decoder.blocks.0 at batch 16 (16 planes
// of 16 x 16; tools/bench_sp.py, us): forward, 8 cout tiles = 128 work-groups: 290 -> 224; data gradient, 10 channel tiles = 160: 232 ->
// 287 as 320 work-groups (a second, quarter-full round) -- so it keeps the 256-pixel tiles.
static inline int sp_groups_per_wave(int N, int Hl, int Wl, int ctiles) {
  const long wgs2 = Wl >= 32 ? (long)N * ((Wl + 31) / 32) * ((Hl + 7) / 8) * ctiles : (long)N * ((Wl + 15) / 16) * ((Hl + 15) / 16) * ctiles;
  static const int force = [] { const char* e = getenv("STARCOP_SP_NPP"); return e ? atoi(e) : 0; }();      // (tests, A/B: 1 or 2)
  if (force == 1 || force == 2) return force;
  return wgs2 <= 128 ? 1 : 2;
}

extern "C" int sc_sp_stat_rows(int N, int H, int W, int Cout) {
  const int Hl = H / 2, Wl = W / 2;
  const int TW = Wl >= 32 ? 32 : 16, TH = 128 * sp_groups_per_wave(N, Hl, Wl, (Cout + 31) / 32) / TW;
  return N * ((Wl + TW - 1) / TW) * ((Hl + TH - 1) / TH);
}

extern "C" int sc_conv3x3_sp(const sc_conv_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_conv3x3_sp: null args");
  SC_REQUIRE(a->ks == 3 && (a->nsrc == 1 || a->nsrc == 2), "sc_conv3x3_sp: ks = 3, one (up-sampled) or two (up-sampled, skip) sources");
  SC_REQUIRE(a->src[0].up == 1, "sc_conv3x3_sp: src[0] must be the half-resolution tensor (up = 1)");
  SC_REQUIRE(a->nsrc == 1 || a->src[1].up == 0, "sc_conv3x3_sp: src[1] is the full-resolution skip tensor (up = 0)");
  for (int s = 0; s < a->nsrc; ++s) {
    SC_REQUIRE(a->src[s].mode == SC_SRC_RAW || a->src[s].mode == SC_SRC_AFFINE, "sc_conv3x3_sp: RAW or AFFINE sources");
    SC_REQUIRE(a->src[s].mode == SC_SRC_RAW || a->src[s].cst != nullptr, "sc_conv3x3_sp: source %d needs constants", s);
    SC_REQUIRE(a->src[s].C > 0, "sc_conv3x3_sp: source %d has no channels", s);
  }
  SC_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cout > 0, "sc_conv3x3_sp: bad shape");
  SC_REQUIRE(a->H % 2 == 0 && a->W % 2 == 0, "sc_conv3x3_sp: even output size");
  SC_REQUIRE(a->terms == SC_TERMS_F16X2 || a->terms == 1, "sc_conv3x3_sp: terms = SC_TERMS_F16X2 (two fp16 terms) or 1 (one bf16 term)");
  SC_REQUIRE(a->csplit == a->Cout && a->out1 == nullptr && a->add0 == nullptr && a->add1 == nullptr && !a->down0 && !a->accum0,
             "sc_conv3x3_sp: a single plain output");
  SC_REQUIRE(((uintptr_t)a->wpk & 15) == 0 && ((uintptr_t)a->out0 & 7) == 0, "sc_conv3x3_sp: alignment");
  SC_REQUIRE((size_t)32 * a->H * a->W < (1ull << 32), "sc_conv3x3_sp: plane too large");
  ConvSPP p;
  p.s0 = to_srcd(a->src[0]);
  p.s1 = a->nsrc == 2 ? to_srcd(a->src[1]) : p.s0;          // (C = 0 below: never selected, but its pointers stay valid)
  if (a->nsrc != 2) p.s1.C = 0;
  if (p.s0.mode == SC_SRC_RAW) p.s0.cst = sc_identity_cst_table(p.s0.C);
  if (p.s1.mode == SC_SRC_RAW) p.s1.cst = sc_identity_cst_table(p.s1.C);
  SC_REQUIRE(p.s0.cst != nullptr && p.s1.cst != nullptr, "sc_conv3x3_sp: identity constants unavailable");
  p.wpk = reinterpret_cast<const uintx4*>(a->wpk);
  p.N = a->N; p.Hl = a->H / 2; p.Wl = a->W / 2; p.Cout = a->Cout;
  p.out = a->out0; p.stats = a->stats;
  p.xb0 = a->xbound[0]; p.xb1 = a->nsrc == 2 ? a->xbound[1] : nullptr;
  const int npp = sp_groups_per_wave(a->N, p.Hl, p.Wl, (a->Cout + 31) / 32);
  const int TW = p.Wl >= 32 ? 32 : 16, TH = 128 * npp / TW;
  const long tiles = (long)((p.Wl + TW - 1) / TW) * ((p.Hl + TH - 1) / TH) * a->N;
  const long ncot = (a->Cout + 31) / 32;
  const long grid = (tiles + 7) / 8 * 8 * ncot;
  SC_REQUIRE(grid < (1L << 31), "sc_conv3x3_sp: grid too large");
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_sp<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<32>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_sp<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<16>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_sp<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<32>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_sp<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<16>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_sp<16, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<16>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_sp<16, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<16>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_sp<32, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<32>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_sp<32, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<32>()) == hipSuccess;
  }();
  SC_REQUIRE(attr_ok, "sc_conv3x3_sp: cannot reserve %d bytes of LDS", sp_smem_bytes<32>());
  const bool bf = a->terms == 1;
  if (TW == 32 && npp == 1) {
    if (bf) hipLaunchKernelGGL((k_conv3_sp<32, true, 1>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<32>(), (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_conv3_sp<32, false, 1>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<32>(), (hipStream_t)stream, p);
  } else if (TW == 32) {
    if (bf) hipLaunchKernelGGL((k_conv3_sp<32, true>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<32>(), (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_conv3_sp<32, false>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<32>(), (hipStream_t)stream, p);
  } else if (npp == 1) {
    if (bf) hipLaunchKernelGGL((k_conv3_sp<16, true, 1>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<16>(), (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_conv3_sp<16, false, 1>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<16>(), (hipStream_t)stream, p);
  } else {
    if (bf) hipLaunchKernelGGL((k_conv3_sp<16, true>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<16>(), (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_conv3_sp<16, false>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<16>(), (hipStream_t)stream, p);
  }
  SC_LAUNCH_OK("sc_conv3x3_sp");
  return SC_OK;
}

extern "C" size_t sc_packed_weight_floats_spd(int Cout, int Cup, int Cskip_tiles) {
  return (size_t)((Cup + 127) / 128 + (Cskip_tiles + 31) / 32) * 4 * ((Cout + 15) / 16) * SP_WST * 4;
}

extern "C" int sc_spd_vskip_ok(int Cup, int Cskip) { return Cup > 0 && Cup <= 64 && Cskip > 0 && Cskip <= 16; }

extern "C" int sc_pack_weights_spd(const float* w, float* wpk, int Cout, int CinTotal, int Cup, int vskip, int terms, sc_stream stream) {
  SC_REQUIRE(w && wpk && Cout > 0 && Cup > 0 && Cup <= CinTotal, "sc_pack_weights_spd: bad argument");
  SC_REQUIRE(terms == SC_TERMS_F16X2 || terms == 1, "sc_pack_weights_spd: terms must be SC_TERMS_F16X2 or 1 (one bf16 term)");
  SC_REQUIRE(vskip >= 0 && vskip <= 2, "sc_pack_weights_spd: vskip must be 0 (up-sampled channels only), 1 (virtual skip channels) or 2 (skip tiles)");
  SC_REQUIRE(vskip != 1 || sc_spd_vskip_ok(Cup, CinTotal - Cup), "sc_pack_weights_spd: virtual skip channels need Cup <= 64 and 1..16 skip channels");
  SC_REQUIRE(vskip != 2 || CinTotal > Cup, "sc_pack_weights_spd: skip tiles need skip channels");
  SC_REQUIRE(((uintptr_t)wpk & 15) == 0, "sc_pack_weights_spd: destination must be 16-byte aligned");
  const size_t total = spd_pack_items(Cout, Cup, vskip == 2 ? CinTotal - Cup : 0);
  hipLaunchKernelGGL(k_pack_weights_spd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<unsigned short*>(wpk), Cout, CinTotal, Cup, vskip, terms == 1 ? 1 : 0, total);
  SC_LAUNCH_OK("sc_pack_weights_spd");
  return SC_OK;
}

extern "C" int sc_conv3x3_sp_dgrad(const sc_conv_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_conv3x3_sp_dgrad: null args");
  SC_REQUIRE(a->ks == 3 && a->nsrc == 1, "sc_conv3x3_sp_dgrad: ks = 3, one source (the layer's output gradient)");
  SC_REQUIRE(a->src[0].mode == SC_SRC_BNBWD && a->src[0].aux != nullptr && a->src[0].cst != nullptr && a->src[0].up == 0,
             "sc_conv3x3_sp_dgrad: the source is the full-resolution SC_SRC_BNBWD operand (g, y, constants)");
  SC_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cout > 0 && a->src[0].C > 0, "sc_conv3x3_sp_dgrad: bad shape");
  SC_REQUIRE(a->H % 2 == 0 && a->W % 2 == 0, "sc_conv3x3_sp_dgrad: even gradient size");
  SC_REQUIRE(a->terms == SC_TERMS_F16X2 || a->terms == 1, "sc_conv3x3_sp_dgrad: terms = SC_TERMS_F16X2 (two fp16 terms) or 1 (one bf16 term)");
  SC_REQUIRE(a->add0 == nullptr && a->add1 == nullptr && a->stats == nullptr && a->csplit > 0 && a->csplit <= a->Cout,
             "sc_conv3x3_sp_dgrad: outputs are out0 [N, csplit, H/2, W/2] (+ out1 [N, Cout - csplit, H, W]); no add / stats");
  SC_REQUIRE((a->csplit == a->Cout) == (a->out1 == nullptr), "sc_conv3x3_sp_dgrad: out1 exactly when csplit < Cout");
  SC_REQUIRE(((uintptr_t)a->wpk & 15) == 0, "sc_conv3x3_sp_dgrad: alignment");
  SC_REQUIRE((size_t)128 * (a->H / 2) * (a->W / 2) < (1ull << 32) && (size_t)a->H * a->W < (1ull << 31), "sc_conv3x3_sp_dgrad: plane too large");
  ConvSPD p;
  p.dy = to_srcd(a->src[0]);
  p.wpk = reinterpret_cast<const uintx4*>(a->wpk);
  p.N = a->N; p.Hl = a->H / 2; p.Wl = a->W / 2; p.Cup = a->csplit;
  p.out = a->out0; p.accum = a->accum0; p.absmax = a->absmax;
  p.out_skip = a->out1; p.Cskip = a->Cout - a->csplit; p.accum_skip = a->accum1;
  // the skip channels' gradient in the same launch: in the idle half of the one channel tile (vskip) where that fits, else as skip tiles
  p.skip_mode = a->out1 == nullptr ? 0 : (sc_spd_vskip_ok(a->csplit, a->Cout - a->csplit) ? 1 : 2);
  p.ntu = (a->csplit + 127) / 128;
  p.nts = p.skip_mode == 2 ? (p.Cskip + 31) / 32 : 0;
  const int npp = sp_groups_per_wave(a->N, p.Hl, p.Wl, p.ntu + p.nts);
  const int TW = p.Wl >= 32 ? 32 : 16, TH = 128 * npp / TW;
  const long tiles = (long)((p.Wl + TW - 1) / TW) * ((p.Hl + TH - 1) / TH) * a->N;
  const long ncot = p.ntu + p.nts;
  const long grid = (tiles + 7) / 8 * 8 * ncot;
  SC_REQUIRE(grid < (1L << 31), "sc_conv3x3_sp_dgrad: grid too large");
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_spd<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<32>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_spd<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<16>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_spd<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<32>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_spd<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<16>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_spd<16, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<16>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_spd<16, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<16>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_spd<32, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<32>()) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_spd<32, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes<32>()) == hipSuccess;
  }();
  SC_REQUIRE(attr_ok, "sc_conv3x3_sp_dgrad: cannot reserve %d bytes of LDS", sp_smem_bytes<32>());
  const bool bf = a->terms == 1;
  if (TW == 32 && npp == 1) {
    if (bf) hipLaunchKernelGGL((k_conv3_spd<32, true, 1>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<32>(), (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_conv3_spd<32, false, 1>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<32>(), (hipStream_t)stream, p);
  } else if (TW == 32) {
    if (bf) hipLaunchKernelGGL((k_conv3_spd<32, true>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<32>(), (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_conv3_spd<32, false>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<32>(), (hipStream_t)stream, p);
  } else if (npp == 1) {
    if (bf) hipLaunchKernelGGL((k_conv3_spd<16, true, 1>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<16>(), (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_conv3_spd<16, false, 1>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<16>(), (hipStream_t)stream, p);
  } else {
    if (bf) hipLaunchKernelGGL((k_conv3_spd<16, true>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<16>(), (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_conv3_spd<16, false>), dim3((unsigned)grid), dim3(512), sp_smem_bytes<16>(), (hipStream_t)stream, p);
  }
  SC_LAUNCH_OK("sc_conv3x3_sp_dgrad");
  return SC_OK;
}
