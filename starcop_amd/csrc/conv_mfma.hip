// Dense NCHW conv2d (k in {1,3}, stride 1, pad k/2) as implicit GEMM on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32) of gfx950: forward, backward-data (same kernel, transposed+flipped
// packed filter) and backward-weight.
//
// Replaces, for the smp.Unet built at starcop/models/model_module.py:244-251, the torch ops
//   F.conv2d  +  (producer's) F.batch_norm / relu / relu6  +  F.interpolate(nearest, x2) + torch.cat
// and their autograd backward.  Producers store RAW conv outputs; BatchNorm + activation (or the
// BatchNorm/activation backward, or DataNormalizer) are applied while the tile is staged to LDS.
//
// GEMM view (forward):  D[co][pixel] = sum_{ci,tap} Wp[ci][tap][co] * patch[ci][pixel + d(tap)]
//   A operand (32 x 2): lane l -> A[i = l&31 (cout)][k = l>>5 (ci of the pair)]
//   B operand ( 2 x 32): lane l -> B[k = l>>5][j = l&31 (pixel in a 32-wide row segment)]
//   D (32 x 32, 16 regs): col = l&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(l>>5) (cout)
// Work-group = 4 waves; tile = (32*RM couts) x (4 rows x 32 cols | 128 flat pixels); wave w owns
// pixel row w and all RM cout blocks.  K loop = chunks of 8 input channels, double-buffered in LDS,
// global loads for chunk k+1 issued before the MFMAs of chunk k (register staged because of the
// prologue), one barrier per chunk.
#include "sc_common.h"
#include <vector>
#include <cstdlib>

namespace {

struct ConvP {
  SrcD s0, s1;
  const float* wpk;
  int N, H, W, Cout;
  float* out0; float* out1;
  int csplit, accum0, accum1;
  const float* add0; const float* add1;
  float* stats;
  int xcdmap;     // 1: 1-D grid; each XCD walks a contiguous eighth of the pixel tiles, the cout tiles of a pixel tile back to back
};

// BNB: the (single) source is a BatchNorm-backward source (dgrad launches); mixing it with other modes in a concat is not used
template <int KS, int RM, bool BNB, int PF_ = 1>
__global__ __launch_bounds__(256, 2) void k_conv_mfma(const ConvP p) {
  constexpr int TAPS = KS * KS;
  constexpr int CO_T = 32 * RM;
  constexpr int PR = (KS == 3) ? 6 : 4;
  constexpr int PC = (KS == 3) ? 34 : 32;
  constexpr int PCH = PR * PC;                 // patch floats per channel
  constexpr int KC = (KS == 3) ? 8 : 16;       // input channels per K chunk (one barrier per chunk)
  // staging decomposition.  3x3: a half-wave per channel (8 channels per round), 32 lanes over the 6x34 patch.  1x1: a WAVE per
  // channel (4 per round, 4 rounds), 64 lanes over the 128 pixels in two slots: the channel is wave-uniform, so its plane pointer
  // and BatchNorm constants live in SGPRs and a staged element costs one global load + the prologue instead of per-lane 64-bit
  // address arithmetic and constant broadcasts (these kernels were VALU-issue bound: 940 VALU per 141 MFMA per wave, SQ counters
  // in profiles/r02_pmc_sq_pointwise.txt)
  constexpr int NE = (KS == 3) ? (PCH + 31) / 32 : 2;
  constexpr int NC = (KS == 3) ? KC / 8 : KC / 4;
  constexpr int WCH = KC * TAPS * CO_T;        // weight floats per chunk
  constexpr int NW = (WCH / 4 + 255) / 256;    // float4 per thread

  __shared__ __attribute__((aligned(16))) float s_w[2][WCH];
  __shared__ float s_p[2][KC * PCH];
  __shared__ float s_red[8][CO_T][2];       // [wave][16-lane row of the half-wave]: two partials per 32-pixel sum, see row_sum16

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, W = p.W;
  const int per_img = (KS == 3) ? ((W + 31) >> 5) * ((H + 3) >> 2) : (H * W + 127) >> 7;
  int n, cot, tile;
  if (p.xcdmap) {
    // work-groups go to the 8 XCDs round-robin by linear id; each XCD has its own L2.  XCD k walks the k-th contiguous eighth of
    // the pixel tiles, all cout tiles of a pixel tile back to back: the input tile they share (and, for 3x3, the halo lines of
    // neighbouring tiles) is fetched from memory once and then hits in that L2 -- lower latency for these latency-bound kernels
    const int ncot = (p.Cout + CO_T - 1) / CO_T;
    const int total = per_img * p.N, per_xcd = (total + 7) >> 3;
    const int slot = blockIdx.x >> 3, j = slot / ncot;
    const int pt = (blockIdx.x & 7) * per_xcd + j;
    if (j >= per_xcd || pt >= total) return;
    cot = slot - j * ncot;
    n = pt / per_img; tile = pt - n * per_img;
  } else {
    n = blockIdx.z; cot = blockIdx.y; tile = blockIdx.x;
  }
  // integer division has no scalar form: its (uniform) results come back in VGPRs and drag every address derived from them into
  // 64-bit VALU arithmetic (11 instructions per channel plane and chunk in the 1x1 kernel).  Say that they are uniform.
  n = __builtin_amdgcn_readfirstlane(n); cot = __builtin_amdgcn_readfirstlane(cot); tile = __builtin_amdgcn_readfirstlane(tile);
  int y0 = 0, x0 = 0, p0 = 0;
  if (KS == 3) {
    const int tiles_x = (W + 31) >> 5;
    const int ty = tile / tiles_x;
    y0 = ty * 4;
    x0 = (tile - ty * tiles_x) * 32;
  } else {
    p0 = tile * 128;
  }
  const int C0 = p.s0.C;
  const int Cin = C0 + p.s1.C;
  const int nk = (Cin + KC - 1) / KC;          // packed filters are zero-padded to nk*KC input channels
  const float* wbase = p.wpk + (size_t)cot * nk * WCH;

  floatx16 acc[RM];
#pragma unroll
  for (int m = 0; m < RM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // ---- staging state (registers) ----
  const int sci = (KS == 3) ? tid >> 5 : __builtin_amdgcn_readfirstlane(tid >> 6);     // first channel of the chunk this thread stages
  const int sq = (KS == 3) ? tid & 31 : tid & 63;
  constexpr int CSTEP = (KS == 3) ? 8 : 4, ESTEP = (KS == 3) ? 32 : 64;
  // pixel offsets of this thread's patch positions: the same for every chunk (per source: `up` may differ)
  unsigned off0[NE], off1[NE];
  unsigned inb = 0;
  {
    const int up0 = p.s0.up, up1 = p.s1.up;
    const int Ws0 = W >> up0, Ws1 = W >> up1;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      bool ok;
      if (KS == 3) {
        const int e = sq + 32 * i;
        const int pr = e / PC, pc = e - pr * PC;
        const int y = y0 - 1 + pr, x = x0 - 1 + pc;
        ok = (e < PCH) && (y >= 0) && (y < H) && (x >= 0) && (x < W);
        off0[i] = ok ? (unsigned)((y >> up0) * Ws0 + (x >> up0)) : 0u;      // clamped: unconditional loads, no exec-mask branches
        off1[i] = ok ? (unsigned)((y >> up1) * Ws1 + (x >> up1)) : 0u;
      } else {
        const int pix = p0 + sq + ESTEP * i;
        ok = pix < H * W;
        off0[i] = ok ? (unsigned)pix : (unsigned)(H * W - 1);      // clamped to a valid pixel: a 1x1 output depends on its own pixel only
        off1[i] = off0[i];
      }
      inb |= ok ? (1u << i) : 0u;
    }
  }
  // Staging registers of one K chunk.  1x1 layers keep TWO chunks in flight (PF = 2): their work per chunk (16 channels x 128 pixels,
  // 8-16 MFMAs per wave) is far shorter than a memory round trip, so with one chunk of look-ahead every iteration waits for its loads
  // (measured: 2.9k cycles per chunk against ~1k of MFMA); 3x3 chunks carry 9x the MFMA work and stay at one.
  constexpr int PF = (KS == 1) ? PF_ : 1;
  struct Stg {
    float xv[NC][NE], av[BNB ? NC : 1][NE];
    bool chok[NC];
    float4 c0[NC];
    float c4[NC];
    floatx4 wv[NW];
    float slo, shi;
  };
  Stg stg[PF];

  auto load_chunk = [&](int kc, Stg& g) {
    // the source (of a concat) is uniform per chunk: KC divides the first source's channel count when nsrc == 2
    const bool second = kc * KC >= C0;
    const SrcD& s = second ? p.s1 : p.s0;
    g.slo = sc_act_lo(s.act); g.shi = sc_act_hi(s.act);
    const size_t plane = (size_t)(H >> s.up) * (W >> s.up);
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      const int cg = kc * KC + sci + CSTEP * cc;      // channel in concat space
      g.chok[cc] = cg < Cin;
      // 1x1: the channel is wave-uniform (sci is); saying so keeps the plane pointers in SGPRs (scalar multiplies, saddr loads
      // with the 32-bit pixel offset) instead of a 64-bit VALU multiply chain per channel and chunk
      const int cs_v = g.chok[cc] ? (second ? cg - C0 : cg) : 0;
      const int cs = (KS == 1) ? __builtin_amdgcn_readfirstlane(cs_v) : cs_v;
      // 1x1: never a load under a branch.  For the wave-uniform channels of the 1x1 kernels these are SCALAR loads, and a
      // conditional one is waited for (lgkmcnt(0)) inside its branch: four exposed scalar-memory round trips per chunk.  The host
      // therefore hands RAW sources a table of identity constants (sc_identity_cst) and the kernel always loads.
      if (KS == 1 || s.mode != SC_SRC_RAW) {
        g.c0[cc] = *reinterpret_cast<const float4*>(s.cst + (size_t)cs * SC_CST);
        g.c4[cc] = BNB ? s.cst[(size_t)cs * SC_CST + 4] : 0.f;
      } else {
        g.c0[cc] = make_float4(1.f, 0.f, 0.f, 0.f); g.c4[cc] = 0.f;
      }
      const float* xb = s.x + ((size_t)n * s.C + cs) * plane;
      const float* ab = BNB ? s.aux + ((size_t)n * s.C + cs) * plane : nullptr;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const unsigned o = second ? off1[i] : off0[i];
        if (KS == 1) {
          // uniform plane pointer + 32-bit BYTE offset (host-checked: a plane is < 4 GB): one saddr load, no 64-bit lane arithmetic
          const unsigned ob = o * 4u;
          g.xv[cc][i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xb) + ob);
          if (BNB) g.av[cc][i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ab) + ob);
        } else {
          g.xv[cc][i] = xb[o];
          if (BNB) g.av[cc][i] = ab[o];
        }
      }
    }
    const floatx4* wsrc = reinterpret_cast<const floatx4*>(wbase + (size_t)kc * WCH);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i4 = tid + 256 * j;
      g.wv[j] = wsrc[i4 < WCH / 4 ? i4 : WCH / 4 - 1];   // clamped: unconditional load keeps wv in registers
    }
  };

  auto store_chunk = [&](int buf, const Stg& g) {
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      const int chl = sci + CSTEP * cc;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int e = sq + ESTEP * i;
        if (e < PCH) {
          const float v = BNB ? sc_pro_bnbwd(g.xv[cc][i], g.av[BNB ? cc : 0][i], g.c0[cc].x, g.c0[cc].y, g.c0[cc].z, g.c0[cc].w, g.c4[cc], g.slo, g.shi)
                              : sc_pro_affine(g.xv[cc][i], g.c0[cc].x, g.c0[cc].y, g.slo, g.shi);
          // 1x1: out-of-range pixel slots hold a clamped duplicate (their outputs are masked in the epilogue); padded channels are zero
          if (KS == 3) s_p[buf][chl * PCH + e] = (((inb >> i) & 1u) && g.chok[cc]) ? v : 0.f;
          else s_p[buf][chl * PCH + e] = g.chok[cc] ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i4 = tid + 256 * j;
      if (i4 < WCH / 4) *reinterpret_cast<floatx4*>(&s_w[buf][i4 * 4]) = g.wv[j];
    }
  };

  auto compute_chunk = [&](int buf) {
    if constexpr (KS == 1) {
      // all operand reads of the chunk first, then the MFMAs: the compiler otherwise emits read, wait lgkmcnt(0), MFMA for every
      // K step, i.e. one exposed LDS round trip per MFMA
      float b[KC / 2], a[KC / 2][RM];
#pragma unroll
      for (int cp = 0; cp < KC / 2; ++cp) {
        const int cil = 2 * cp + lhi;
        b[cp] = s_p[buf][cil * PCH + wave * 32 + l31];
#pragma unroll
        for (int m = 0; m < RM; ++m) a[cp][m] = s_w[buf][cil * CO_T + m * 32 + l31];
      }
#pragma unroll
      for (int cp = 0; cp < KC / 2; ++cp)
#pragma unroll
        for (int m = 0; m < RM; ++m) {
#ifdef SC_EXPERIMENT_SKIP_MFMA
          acc[m][0] = fmaf(a[cp][m], b[cp], acc[m][0]);
#else
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp][m], b[cp], acc[m], 0, 0, 0);
#endif
        }
      return;
    }
#pragma unroll
    for (int cp = 0; cp < KC / 2; ++cp) {
      const int cil = 2 * cp + lhi;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        float b;
        if (KS == 3) {
          const int kh = tap / 3, kw = tap - 3 * kh;
          b = s_p[buf][cil * PCH + (wave + kh) * PC + l31 + kw];
        } else {
          b = s_p[buf][cil * PCH + wave * 32 + l31];
        }
#pragma unroll
        for (int m = 0; m < RM; ++m) {
          const float a = s_w[buf][(cil * TAPS + tap) * CO_T + m * 32 + l31];
#ifdef SC_EXPERIMENT_SKIP_MFMA
          acc[m][0] = fmaf(a, b, acc[m][0]);      // timing experiment only: operand traffic without the matrix work
#else
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#endif
        }
      }
    }
  };

  if (PF == 1) {
    load_chunk(0, stg[0]);
    store_chunk(0, stg[0]);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
      const bool more = (kc + 1) < nk;
      if (more) load_chunk(kc + 1, stg[0]);
      compute_chunk(kc & 1);
      if (more) store_chunk((kc + 1) & 1, stg[0]);
      __syncthreads();
    }
  } else {
    // Two chunks of look-ahead, STRAIGHT-LINE: chunk kc + 2 is requested into the register set that the store of chunk kc just
    // freed, every load is unconditional (past the end: the last chunk again, stored but never multiplied) and only the
    // LDS-read + MFMA block of an odd tail sits under a (uniform) branch.  The first version of this path had `if (kc + 2 < nk)`
    // around its loads; hipcc merges the two request histories at such a join into s_waitcnt vmcnt(0), which turned the
    // look-ahead into "wait for everything" (and measured slower than PF = 1).
    auto ld = [&](int kc, Stg& g) { load_chunk(kc < nk ? kc : nk - 1, g); };
    ld(0, stg[0]);
    ld(1, stg[PF - 1]);
    store_chunk(0, stg[0]);
    __syncthreads();
    for (int kc = 0; kc < nk; kc += 2) {
      ld(kc + 2, stg[0]);
      compute_chunk(0);
      store_chunk(1, stg[PF - 1]);
      __syncthreads();
      ld(kc + 3, stg[PF - 1]);
      if (kc + 1 < nk) compute_chunk(1);
      store_chunk(0, stg[0]);
      __syncthreads();
    }
  }

  // ---- epilogue ----
  int oy = 0, ox = 0, opix = 0;
  bool pix_ok;
  if (KS == 3) {
    oy = y0 + wave; ox = x0 + l31;
    pix_ok = (oy < H) && (ox < W);
    opix = oy * W + ox;
  } else {
    opix = p0 + wave * 32 + l31;
    pix_ok = opix < H * W;
  }
  const size_t HWs = (size_t)H * W;
  const bool want_stats = p.stats != nullptr;
  if (p.csplit == p.Cout || p.csplit % CO_T == 0) {
    // the cout tile lies entirely in one output: uniform base pointer + 32-bit lane offsets (see k_conv3_bx3's epilogue)
    const bool first = cot * CO_T < p.csplit;
    const int Cs = first ? p.csplit : p.Cout - p.csplit;
    const int c0 = first ? cot * CO_T : cot * CO_T - p.csplit;
    const bool accum = first ? p.accum0 != 0 : p.accum1 != 0;
    const int Climit = first ? p.csplit : p.Cout;
    const size_t cbase = ((size_t)n * Cs + c0) * HWs;
    float* const ob = (first ? p.out0 : p.out1) + cbase;
    const float* const a0 = p.add0 ? p.add0 + cbase : nullptr;
    const float* const a1 = p.add1 ? p.add1 + cbase : nullptr;
    const unsigned hw32 = (unsigned)HWs;
    const unsigned loff = (unsigned)(4 * lhi) * hw32 + (unsigned)(pix_ok ? opix : 0);
#pragma unroll
    for (int m = 0; m < RM; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = m * 32 + (r & 3) + 8 * (r >> 2);
        const int col = cu + 4 * lhi;
        float v = acc[m][r];
        const bool ok = pix_ok && (cot * CO_T + col < Climit);
        if (!ok) v = 0.f;
        if (want_stats) {
          const float s = row_sum16(v);
          const float ss = row_sum16(v * v);
          if ((l31 & 15) == 0) { s_red[2 * wave + (l31 >> 4)][col][0] = s; s_red[2 * wave + (l31 >> 4)][col][1] = ss; }
        }
        if (ok) {
          const unsigned off = loff + (unsigned)cu * hw32;
          if (a0) v += a0[off];
          if (a1) v += a1[off];
          if (accum) v += ob[off];
          ob[off] = v;
        }
      }
    }
  } else
#pragma unroll
  for (int m = 0; m < RM; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int co = cot * CO_T + col;
      float v = acc[m][r];
      const bool ok = pix_ok && (co < p.Cout);
      if (!ok) v = 0.f;
      if (want_stats) {
        const float s = row_sum16(v);
        const float ss = row_sum16(v * v);
        if ((l31 & 15) == 0) { s_red[2 * wave + (l31 >> 4)][col][0] = s; s_red[2 * wave + (l31 >> 4)][col][1] = ss; }
      }
      if (ok) {
        float* o; size_t idx; int accum;
        if (co < p.csplit) {
          idx = ((size_t)n * p.csplit + co) * HWs + opix; o = p.out0; accum = p.accum0;
        } else {
          idx = ((size_t)n * (p.Cout - p.csplit) + (co - p.csplit)) * HWs + opix; o = p.out1; accum = p.accum1;
        }
        if (p.add0) v += p.add0[idx];
        if (p.add1) v += p.add1[idx];
        if (accum) v += o[idx];
        o[idx] = v;
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < CO_T * 2) {
      const int col = tid >> 1, k = tid & 1;
      const int co = cot * CO_T + col;
      if (co < p.Cout) {
        // (the association of the former half-wave sums: bit-identical statistics)
        const float t = (((s_red[0][col][k] + s_red[1][col][k]) + (s_red[2][col][k] + s_red[3][col][k])) + (s_red[4][col][k] + s_red[5][col][k])) +
                        (s_red[6][col][k] + s_red[7][col][k]);
        p.stats[(((size_t)n * per_img + tile) * p.Cout + co) * 2 + k] = t;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Thin layers (Cout <= 16: decoder.blocks.4.*): the same implicit GEMM on v_mfma_f32_16x16x4_f32 so that no MFMA row is
// wasted.  A: lane -> W[co = l&15][ci = 4*kq + (l>>4)],  B: lane -> patch[ci = 4*kq + (l>>4)][pixel = 16*pb + (l&15)],
// D (4 regs): col = l&15 (pixel), row = 4*(l>>4) + r (cout).  Tile = 16 couts x (4 rows x 32 px), wave w owns row w.
// RW: image rows per wave (tile = 4*RW rows x 32 px): RW = 2 halves the filter staging per pixel and the halo overhead
template <bool BNB, int RW>
__global__ __launch_bounds__(256, 2) void k_conv_mfma16(const ConvP p) {
  constexpr int TAPS = 9, CO_T = 16, KC = 8;
  constexpr int PR = 4 * RW + 2, PC = 34, PCH = PR * PC, NE = (PCH + 31) / 32;
  constexpr int PCHP = NE * 32;                // padded channel pitch: the NE staging rounds store unconditionally
  constexpr int WCH = KC * TAPS * CO_T;        // 1152 floats per chunk
  constexpr int NW = (WCH / 4 + 255) / 256;    // 2
  constexpr int WCHP = NW * 256 * 4;           // padded likewise

  __shared__ __attribute__((aligned(16))) float s_w[2][WCHP];
  __shared__ float s_p[2][KC * PCHP];
  __shared__ float s_red[4][CO_T][2];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int n = blockIdx.z;
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  const int ty = blockIdx.x / tiles_x;
  const int y0 = ty * 4 * RW, x0 = (blockIdx.x - ty * tiles_x) * 32;
  const int C0 = p.s0.C;
  const int Cin = C0 + p.s1.C;
  const int nk = (Cin + KC - 1) / KC;
  const float* wbase = p.wpk;

  floatx4 acc[RW][2];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) { acc[rr][0] = (floatx4){0.f, 0.f, 0.f, 0.f}; acc[rr][1] = acc[rr][0]; }

  const int sci = tid >> 5, sq = tid & 31;
  // pixel offsets of this thread's patch positions: the same for every chunk (per source: `up` may differ)
  unsigned off0[NE], off1[NE];
  unsigned inb = 0;
  {
    const int up0 = p.s0.up, up1 = p.s1.up;
    const int Ws0 = W >> up0, Ws1 = W >> up1;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = sq + 32 * i;
      const int pr = e / PC, pc = e - pr * PC;
      const int y = y0 - 1 + pr, x = x0 - 1 + pc;
      const bool ok = (e < PCH) && (y >= 0) && (y < H) && (x >= 0) && (x < W);
      off0[i] = ok ? (unsigned)((y >> up0) * Ws0 + (x >> up0)) : 0u;
      off1[i] = ok ? (unsigned)((y >> up1) * Ws1 + (x >> up1)) : 0u;
      inb |= ok ? (1u << i) : 0u;
    }
  }
  float xv[NE], av[BNB ? NE : 1];
  float4 c0 = make_float4(1.f, 0.f, 0.f, 0.f);
  float c4 = 0.f, slo = 0.f, shi = 0.f;
  bool chok = true;
  floatx4 wv[NW];

  auto load_chunk = [&](int kc) {
    const int cg = kc * KC + sci;
    const bool second = cg >= C0;
    const SrcD& s = second ? p.s1 : p.s0;
    chok = cg < Cin;
    const int cs = chok ? (second ? cg - C0 : cg) : 0;
    slo = sc_act_lo(s.act); shi = sc_act_hi(s.act);
    if (s.mode != SC_SRC_RAW) { c0 = *reinterpret_cast<const float4*>(s.cst + (size_t)cs * SC_CST); c4 = BNB ? s.cst[(size_t)cs * SC_CST + 4] : 0.f; }
    else { c0 = make_float4(1.f, 0.f, 0.f, 0.f); c4 = 0.f; }
    const size_t plane = (size_t)(H >> s.up) * (W >> s.up);
    const float* xb = s.x + ((size_t)n * s.C + cs) * plane;
    const float* ab = BNB ? s.aux + ((size_t)n * s.C + cs) * plane : nullptr;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const unsigned o = second ? off1[i] : off0[i];
      xv[i] = xb[o];
      if (BNB) av[i] = ab[o];
    }
    const floatx4* wsrc = reinterpret_cast<const floatx4*>(wbase + (size_t)kc * WCH);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i4 = tid + 256 * j;
      wv[j] = wsrc[i4 < WCH / 4 ? i4 : WCH / 4 - 1];
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const float t = BNB ? sc_pro_bnbwd(xv[i], av[BNB ? i : 0], c0.x, c0.y, c0.z, c0.w, c4, slo, shi) : sc_pro_affine(xv[i], c0.x, c0.y, slo, shi);
      s_p[buf][sci * PCHP + sq + 32 * i] = (((inb >> i) & 1u) && chok) ? t : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) *reinterpret_cast<floatx4*>(&s_w[buf][(tid + 256 * j) * 4]) = wv[j];
  };
  auto compute_chunk = [&](int buf) {
#pragma unroll
    for (int kq = 0; kq < KC / 4; ++kq) {
      const int cil = 4 * kq + lq;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        const float a = s_w[buf][(cil * TAPS + tap) * CO_T + l15];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            const float b = s_p[buf][cil * PCHP + (wave * RW + rr + kh) * PC + pb * 16 + l15 + kw];
            acc[rr][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[rr][pb], 0, 0, 0);
          }
      }
    }
  };

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const bool more = (kc + 1) < nk;
    if (more) load_chunk(kc + 1);
    compute_chunk(kc & 1);
    if (more) store_chunk((kc + 1) & 1);
    __syncthreads();
  }

  const size_t HWs = (size_t)H * W;
  const bool want_stats = p.stats != nullptr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = 4 * lq + r;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int oy = y0 + wave * RW + rr;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const int ox = x0 + pb * 16 + l15;
        float v = acc[rr][pb][r];
        const bool ok = (oy < H) && (ox < W) && (co < p.Cout);
        if (!ok) v = 0.f;
        s += v; ss = fmaf(v, v, ss);
        if (ok) {
          const size_t idx = ((size_t)n * p.Cout + co) * HWs + (size_t)oy * W + ox;
          if (p.add0) v += p.add0[idx];
          if (p.add1) v += p.add1[idx];
          if (p.accum0) v += p.out0[idx];
          p.out0[idx] = v;
        }
      }
    }
    if (want_stats) {      // 16-lane row sums (each row of 16 lanes = one cout)
      SC_DPP_ADD(s, 0xB1, 0xF); SC_DPP_ADD(s, 0x4E, 0xF); SC_DPP_ADD(s, 0x141, 0xF); SC_DPP_ADD(s, 0x140, 0xF);
      SC_DPP_ADD(ss, 0xB1, 0xF); SC_DPP_ADD(ss, 0x4E, 0xF); SC_DPP_ADD(ss, 0x141, 0xF); SC_DPP_ADD(ss, 0x140, 0xF);
      if (l15 == 0) { s_red[wave][co][0] = s; s_red[wave][co][1] = ss; }
    }
  }
  if (want_stats) {
    // statistics rows keep the 4-image-row granularity of SC_STAT_CONV3: with RW = 2 a work-group writes two of them
    __syncthreads();
    const int rows4 = (H + 3) >> 2;
    const int tx = blockIdx.x - ty * tiles_x;
    for (int i = tid; i < RW * CO_T * 2; i += 256) {
      const int hh = i / (CO_T * 2), rem = i - hh * (CO_T * 2);
      const int co = rem >> 1, k2 = rem & 1;
      const int t4 = RW * ty + hh;
      if (co < p.Cout && t4 < rows4) {
        float t = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4 / RW; ++w2) t += s_red[hh * (4 / RW) + w2][co][k2];
        const size_t row = ((size_t)n * rows4 + t4) * tiles_x + tx;
        p.stats[(row * p.Cout + co) * 2 + k2] = t;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Pointwise (1x1) convolution for the low-resolution layers: few pixels, long K (Cin up to 1280).  The 128-pixel tiles of
// k_conv_mfma<1> leave most CUs idle there (N*H*W/128 x Cout/64 work-groups, each walking all of K serially).  Here a
// work-group owns 32 pixels x (32*RM) couts and its four waves split K (chunk kc -> wave kc & 3).  Nothing is shared
// between the waves, so there is no LDS staging and no barrier in the K loop: the MFMA operand layouts ARE coalesced global
// reads (B: 32 consecutive pixels of channel 2cp+(l>>5); A: 32 consecutive couts of the packed filter row), prologue in
// registers, next chunk prefetched.  The four partial accumulators are summed through LDS, then the usual epilogue;
// statistics rows are per 32-pixel tile (SC_STAT_CONV1K).
template <int RM, bool BNB>
__global__ __launch_bounds__(256, 2) void k_conv1_ksplit(const ConvP p) {
  constexpr int CO_T = 32 * RM, KC = 16, PS = 33;
  __shared__ float s_acc[4][CO_T * PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n = blockIdx.z, cot = blockIdx.y;
  const int HW = p.H * p.W;
  const int p0 = blockIdx.x * 32;
  const int Cin = p.s0.C;
  const int nk = (Cin + KC - 1) / KC;
  const float* wbase = p.wpk + (size_t)cot * nk * KC * CO_T;
  const int pix = p0 + l31;
  const bool pok = pix < HW;
  const float* xb = p.s0.x + (size_t)n * Cin * HW + (pok ? pix : 0);
  const float* ab = BNB ? p.s0.aux + (size_t)n * Cin * HW + (pok ? pix : 0) : nullptr;
  const float* cst = p.s0.cst;          // never NULL: the host hands RAW sources the identity table (no load under a branch)
  const float lo = sc_act_lo(p.s0.act), hi = sc_act_hi(p.s0.act);

  floatx16 acc[RM];
#pragma unroll
  for (int m = 0; m < RM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  float bx[8], by[BNB ? 8 : 1], wa[8][RM];
  float4 c0[8]; float c4[BNB ? 8 : 1];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int cp = 0; cp < 8; ++cp) {
      const int ci = kc * KC + 2 * cp + lhi;
      const int cc = ci < Cin ? ci : 0;
      bx[cp] = xb[(size_t)cc * HW];
      if (BNB) by[cp] = ab[(size_t)cc * HW];
      if (BNB) { c0[cp] = *reinterpret_cast<const float4*>(cst + (size_t)cc * SC_CST); c4[cp] = cst[(size_t)cc * SC_CST + 4]; }
      else { const float2 t = *reinterpret_cast<const float2*>(cst + (size_t)cc * SC_CST); c0[cp] = make_float4(t.x, t.y, 0.f, 0.f); }
#pragma unroll
      for (int m = 0; m < RM; ++m) wa[cp][m] = wbase[(size_t)ci * CO_T + m * 32 + l31];     // filters are zero-padded to nk*16
    }
  };
  int kc = wave;
  if (kc < nk) load_chunk(kc);
  while (kc < nk) {
    float b[8], a[8][RM];
#pragma unroll
    for (int cp = 0; cp < 8; ++cp) {
      const int ci = kc * KC + 2 * cp + lhi;
      const float t = BNB ? sc_pro_bnbwd(bx[cp], by[BNB ? cp : 0], c0[cp].x, c0[cp].y, c0[cp].z, c0[cp].w, c4[BNB ? cp : 0], lo, hi)
                          : sc_pro_affine(bx[cp], c0[cp].x, c0[cp].y, lo, hi);
      b[cp] = (pok && ci < Cin) ? t : 0.f;
#pragma unroll
      for (int m = 0; m < RM; ++m) a[cp][m] = wa[cp][m];
    }
    const int kn = kc + 4;
    load_chunk(kn < nk ? kn : kc);         // unconditional (past the end: this chunk again, never used): exact vmcnt counting
#pragma unroll
    for (int cp = 0; cp < 8; ++cp)
#pragma unroll
      for (int m = 0; m < RM; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp][m], b[cp], acc[m], 0, 0, 0);
    kc = kn;
  }
  // ---- sum the four K parts ----
#pragma unroll
  for (int m = 0; m < RM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_acc[wave][(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * PS + l31] = acc[m][r];
  __syncthreads();
  const int px = tid & 31, cg = tid >> 5;
  const int opix = p0 + px;
  const bool ok_px = opix < HW;
  const size_t HWs = (size_t)HW;
  const size_t srow = (size_t)n * gridDim.x + blockIdx.x;
  const bool single = p.csplit == p.Cout;
  const size_t cbase = ((size_t)n * p.Cout + (size_t)cot * CO_T) * HWs;
  float* const ob = p.out0 + cbase;
  const float* const a0 = p.add0 ? p.add0 + cbase : nullptr;
  const float* const a1 = p.add1 ? p.add1 + cbase : nullptr;
#pragma unroll
  for (int i = 0; i < CO_T / 8; ++i) {
    const int col = cg + 8 * i;
    const int co = cot * CO_T + col;
    float v = s_acc[0][col * PS + px] + s_acc[1][col * PS + px] + s_acc[2][col * PS + px] + s_acc[3][col * PS + px];
    const bool ok = ok_px && (co < p.Cout);
    if (!ok) v = 0.f;
    if (p.stats) {
      const float s = half_sum32(v), ss = half_sum32(v * v);
      if ((lane & 31) == SC_HALF_SUM_LANE && co < p.Cout) {
        p.stats[(srow * p.Cout + co) * 2] = s;
        p.stats[(srow * p.Cout + co) * 2 + 1] = ss;
      }
    }
    if (ok) {
      if (single) {         // one output (every launch of the network): uniform base + 32-bit offset
        const unsigned off = (unsigned)col * (unsigned)HWs + (unsigned)opix;
        if (a0) v += a0[off];
        if (a1) v += a1[off];
        if (p.accum0) v += ob[off];
        ob[off] = v;
        continue;
      }
      float* o; size_t idx; int accum;
      if (co < p.csplit) {
        idx = ((size_t)n * p.csplit + co) * HWs + opix; o = p.out0; accum = p.accum0;
      } else {
        idx = ((size_t)n * (p.Cout - p.csplit) + (co - p.csplit)) * HWs + opix; o = p.out1; accum = p.accum1;
      }
      if (p.add0) v += p.add0[idx];
      if (p.add1) v += p.add1[idx];
      if (accum) v += o[idx];
      o[idx] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Streaming form of the pointwise FORWARD for the large planes (H*W >= 8192) with a SHORT contraction: features.1's projection, the features.3 / .4
// expansions and the projections' data gradients at 256^2 / 128^2, where a launch moves 150-270 MB and the LDS-staged kernel above runs at 2.8-3.7 TB/s.  GEMM on v_mfma_f32_16x16x4_f32 with the PIXELS PERMUTED so that every
// global access is 16 bytes per lane:
//   a wave owns 64 consecutive pixels; lane (n = l&15, kq = l>>4) reads ONE float4 per K step -- pixels 4n..4n+3 of channel 4*ks + kq --
//   and MFMA j (0..3) takes its element j as the B operand: column n of that MFMA is pixel 4n + j.  After the four MFMAs of a cout
//   block the lane holds, for couts 4*kq + r, the pixels 4n..4n+3: one 16-byte store per (cout block, r), 256-byte runs per cout.
//   A = the filter, W[16*cb + n][4*ks + kq], NCB x NKS registers per lane, read once per wave from the sc_pack_weights layout.
// No LDS staging, no barrier before the epilogue; the K loop is straight-line (exact vmcnt) with a ring of PD float4 loads in flight.
// Work-group = 4 waves: CP = 1: four pixel groups (256 pixels, two statistics rows); CP = 2: two pixel groups x two cout parts.
// Statistics rows as k_conv_mfma<1> (SC_STAT_CONV1: one row per 128 pixels).  Same fp32 products as v_mfma_f32_32x32x2_f32.
#ifndef SC_PWS_PD
#define SC_PWS_PD 8
#endif
struct PwsP {
  const float* x; const float* aux; const float* cst; int act;      // aux: the raw tensor y of a BatchNorm-backward source (BNB)
  const float* wpk; int co_t, Kpad;
  float* out; float* stats;
  int HW, K, M;
};

template <int NCB, int NKS, int CP, bool BNB = false>
__global__ __launch_bounds__(256) void k_pw_stream(const PwsP p) {
  constexpr int PD = NKS < SC_PWS_PD ? NKS : SC_PWS_PD;    // float4 loads in flight per lane
  constexpr int NPG = 4 / CP;                              // pixel groups per work-group
  constexpr int CW = BNB ? 8 : 2;                          // forward (scale, shift) | backward (scale, shift, A, B, D, -, -, -)
  __shared__ __attribute__((aligned(16))) float s_cst[NKS * 4 * CW];
  __shared__ float s_red[4][NCB * 16][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kq = lane >> 4;
  const int K = p.K, M = p.M, HW = p.HW;
  for (int c = tid; c < NKS * 4; c += 256) {
    if constexpr (BNB) {
      float4 c0 = make_float4(1.f, 0.f, 0.f, 0.f); float c4 = 0.f;        // (channels past K: any finite value, their filter entries are 0)
      if (c < K) { c0 = *reinterpret_cast<const float4*>(p.cst + (size_t)c * SC_CST); c4 = p.cst[(size_t)c * SC_CST + 4]; }
      *reinterpret_cast<float4*>(&s_cst[8 * c]) = c0;
      *reinterpret_cast<float4*>(&s_cst[8 * c + 4]) = make_float4(c4, 0.f, 0.f, 0.f);
    } else {
      float sc = 1.f, sh = 0.f;
      if (c < K && p.cst != nullptr) { sc = p.cst[(size_t)c * SC_CST]; sh = p.cst[(size_t)c * SC_CST + 1]; }
      s_cst[2 * c] = sc; s_cst[2 * c + 1] = sh;
    }
  }
  const int pg = wave % NPG, part = wave / NPG;
  const long gpx = ((long)blockIdx.x * NPG + pg) * 64;     // first pixel of the wave's group, over the whole batch (HW % 64 == 0)
  const int img = (int)(gpx / HW), px0 = (int)(gpx - (long)img * HW) + 4 * n16;
  const float* xb = p.x + (size_t)img * K * HW + px0;
  // ---- the ring's first PD requests, then the filter (both in flight while the constants settle)
  const float* yb = BNB ? p.aux + (size_t)img * K * HW + px0 : nullptr;
  float4 xr[PD], yr[BNB ? PD : 1];
#pragma unroll
  for (int u = 0; u < PD; ++u) {
    const int k = 4 * u + kq;
    xr[u] = *reinterpret_cast<const float4*>(xb + (size_t)(k < K ? k : 0) * HW);
    if constexpr (BNB) yr[u] = *reinterpret_cast<const float4*>(yb + (size_t)(k < K ? k : 0) * HW);
  }
  float A[NCB][NKS];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int m = (part * NCB + cb) * 16 + n16;
    const float* wp = p.wpk + ((size_t)(m / p.co_t) * p.Kpad) * p.co_t + (m % p.co_t);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int k = 4 * ks + kq;
      A[cb][ks] = (m < M && k < K) ? wp[(size_t)k * p.co_t] : 0.f;
    }
  }
  floatx4 acc[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[cb][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  const float lo = sc_act_lo(p.act), hi = sc_act_hi(p.act);
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const float4 v = xr[ks % PD];
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (BNB) y = yr[ks % PD];
    if (ks + PD < NKS) {
      const int k = 4 * (ks + PD) + kq;
      xr[ks % PD] = *reinterpret_cast<const float4*>(xb + (size_t)(k < K ? k : 0) * HW);
      if constexpr (BNB) yr[ks % PD] = *reinterpret_cast<const float4*>(yb + (size_t)(k < K ? k : 0) * HW);
    }
    float b[4];
    if constexpr (BNB) {
      const float4 c = *reinterpret_cast<const float4*>(&s_cst[8 * (4 * ks + kq)]);
      const float c4 = s_cst[8 * (4 * ks + kq) + 4];
      b[0] = sc_pro_bnbwd(v.x, y.x, c.x, c.y, c.z, c.w, c4, lo, hi); b[1] = sc_pro_bnbwd(v.y, y.y, c.x, c.y, c.z, c.w, c4, lo, hi);
      b[2] = sc_pro_bnbwd(v.z, y.z, c.x, c.y, c.z, c.w, c4, lo, hi); b[3] = sc_pro_bnbwd(v.w, y.w, c.x, c.y, c.z, c.w, c4, lo, hi);
    } else {
      const float2 c = *reinterpret_cast<const float2*>(&s_cst[2 * (4 * ks + kq)]);
      b[0] = sc_pro_affine(v.x, c.x, c.y, lo, hi); b[1] = sc_pro_affine(v.y, c.x, c.y, lo, hi);
      b[2] = sc_pro_affine(v.z, c.x, c.y, lo, hi); b[3] = sc_pro_affine(v.w, c.x, c.y, lo, hi);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[cb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cb][ks], b[j], acc[cb][j], 0, 0, 0);
  }
  // ---- store (16 bytes per lane and cout) and the per-cout sums of the wave's 64 pixels
  float* ob = p.out + (size_t)img * M * HW + px0;
  const bool want_stats = p.stats != nullptr;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = cb * 16 + 4 * kq + r, co = part * NCB * 16 + col;
      const float4 o = make_float4(acc[cb][0][r], acc[cb][1][r], acc[cb][2][r], acc[cb][3][r]);
      if (co < M) *reinterpret_cast<float4*>(ob + (size_t)co * HW) = o;
      if (want_stats) {
        float sv = (o.x + o.y) + (o.z + o.w), sq = fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, o.w * o.w)));
        sv = row_sum16(sv); sq = row_sum16(sq);          // (DPP adds: the 16 lanes of a row are the 16 pixel quads of this cout)
        if (n16 == 0) { s_red[wave][col][0] = sv; s_red[wave][col][1] = sq; }
      }
    }
  if (want_stats) {
    __syncthreads();
    // rows of 128 pixels = two pixel groups: CP = 1: waves (0, 1) and (2, 3); CP = 2: the two groups of each cout part
    constexpr int NROW = CP == 1 ? 2 : 1, NCOL = NCB * 16 * CP;
    for (int i = tid; i < NROW * NCOL * 2; i += 256) {
      const int k = i & 1, cc = (i >> 1) % NCOL, rw = (i >> 1) / NCOL;
      const int prt = cc / (NCB * 16), col = cc - prt * (NCB * 16);
      const int w0 = CP == 1 ? 2 * rw : prt * NPG;
      if (cc < M) {
        const size_t row = (size_t)blockIdx.x * NROW + rw;
        p.stats[(row * M + cc) * 2 + k] = s_red[w0][col][k] + s_red[w0 + 1][col][k];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight packing
__global__ void k_pack_weights(const float* __restrict__ w, float* __restrict__ wpk, int Cout, int Cin,
                               int taps, int co_t, int tflip, int Kpad, size_t total) {
  // destination-major: wpk[((mt*Kpad + k)*taps + tap)*co_t + col], Kpad = K rounded up to the kernel's chunk size
  //   forward : M = Cout, K = Cin,  value = w[((m*K + k)*taps) + tap]
  //   dgrad   : M = Cin,  K = Cout, value = w[((k*M + m)*taps) + (taps-1-tap)]
  const int M = tflip ? Cin : Cout, K = tflip ? Cout : Cin;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % co_t);
    size_t r = i / co_t;
    const int tap = (int)(r % taps); r /= taps;
    const int k = (int)(r % Kpad);
    const int mt = (int)(r / Kpad);
    const int m = mt * co_t + col;
    float v = 0.f;
    if (m < M && k < K) v = tflip ? w[((size_t)k * M + m) * taps + (taps - 1 - tap)] : w[((size_t)m * K + k) * taps + tap];
    wpk[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient.  GEMM view: D[co][ci] (per tap) = sum_pixels dy[co][pix] * in[ci][pix + d(tap)]
//   A: lane -> dy_lds[co = l&31][pix = 2q + (l>>5)],  B: lane -> in_lds[ci = l&31][pix(+tap)]
// Work-group tile (32*WM couts) x (32*WN cins), WK = 4/(WM*WN) waves split the pixel rows of a stage.
#ifndef SC_WG1_OCC
#define SC_WG1_OCC 1
#endif
struct WgradP {
  SrcD dy, s0, s1;
  int N, H, W, Cout, Cin;
  float* part;
  int nsl;          // K slices (gridDim.x)
  int CoP, CiP;     // padded dims of the partial buffer
};

template <int KS, int WM, int WN>
__global__ __launch_bounds__(256, KS == 1 ? SC_WG1_OCC : 1) void k_wgrad_mfma(const WgradP p) {
  constexpr int TAPS = KS * KS;
  constexpr int WK = 4 / (WM * WN);
  constexpr int SR = (WK == 4) ? 4 : 2;        // pixel rows (of 32) per stage
  constexpr int RW = SR / WK;                  // rows per wave
  constexpr int COT = 32 * WM, CIT = 32 * WN;
  constexpr int PA = SR * 32 + 1;              // dy pitch per channel (odd -> conflict-free)
  constexpr int PRW = SR + (KS == 3 ? 2 : 0);
  constexpr int PCW = (KS == 3) ? 34 : 32;
  constexpr int PB = (PRW * PCW) | 1;

  __shared__ float s_a[COT * PA];
  __shared__ float s_b[CIT * PB];
  __shared__ __attribute__((aligned(16))) float s_ca[COT * SC_CST];
  __shared__ __attribute__((aligned(16))) float s_cb[CIT * SC_CST];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave % WM, wn = (wave / WM) % WN, wk = wave / (WM * WN);
  const int cit = blockIdx.y, cot = blockIdx.z;
  const int H = p.H, W = p.W;
  const int C0 = p.s0.C;

  // per-WG constants -> LDS
  for (int i = tid; i < COT * SC_CST; i += 256) {
    const int ch = cot * COT + i / SC_CST;
    s_ca[i] = (p.dy.cst && p.dy.mode != SC_SRC_RAW && ch < p.Cout) ? p.dy.cst[(size_t)ch * SC_CST + (i % SC_CST)] : ((i % SC_CST) == 0 ? 1.f : 0.f);
  }
  for (int i = tid; i < CIT * SC_CST; i += 256) {
    const int ch = cit * CIT + i / SC_CST;
    float v = (i % SC_CST) == 0 ? 1.f : 0.f;      // RAW == affine(1, 0)
    if (ch < p.Cin) {
      const bool second = ch >= C0;
      const float* cp = second ? p.s1.cst : p.s0.cst;
      const int md = second ? p.s1.mode : p.s0.mode;
      if (cp && md != SC_SRC_RAW) v = cp[(size_t)(second ? ch - C0 : ch) * SC_CST + (i % SC_CST)];
    }
    s_cb[i] = v;
  }

  floatx16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // stage enumeration
  int tiles_x, per_img;
  if (KS == 3) {
    tiles_x = (W + 31) >> 5; per_img = tiles_x * ((H + SR - 1) / SR);
  } else {
    tiles_x = 1; per_img = (H * W + SR * 32 - 1) / (SR * 32);
  }
  const long T = (long)p.N * per_img;
  const long t_begin = T * blockIdx.x / p.nsl, t_end = T * (blockIdx.x + 1) / p.nsl;
  const int dymode = p.dy.mode, dyact = p.dy.act;

  // ---- register-staged prefetch of one stage: all loads issued back to back (clamped addresses, no exec-mask
  //      branches), consumed after the MFMAs with branch-free prologues.
  // dy tile : thread owns pixel a_px of the stage and channels a_c0 + ACH*i   (address = base + i*stride)
  // x patch : half-wave per channel (b_c0 + 8k), lane covers patch positions b_q + 32j
  constexpr int APX = SR * 32, ACH = 256 / APX, NAI = COT / ACH;
  constexpr int BPOS = PRW * PCW, NBJ = (BPOS + 31) / 32, NBK = CIT / 8;
  const int a_px = tid % APX, a_c0 = tid / APX;
  const int b_q = tid & 31, b_c0 = tid >> 5;
  float ag[NAI], ay[NAI], bx[NBK][NBJ];
  int sn = 0, sy0 = 0, sx0 = 0, sp0 = 0;     // coordinates of the prefetched stage

  auto a_pix = [&](int y0, int x0, int p0, bool& ok) -> int {
    if (KS == 3) {
      const int y = y0 + (a_px >> 5), x = x0 + (a_px & 31);
      ok = (y < H) && (x < W);
      return y * W + x;
    }
    const int pix = p0 + a_px;
    ok = pix < H * W;
    return pix;
  };
  auto b_pos = [&](int j, int y0, int x0, int p0, int up, bool& ok) -> int {
    const int e = b_q + 32 * j;
    if (KS == 3) {
      const int pr = e / PCW, pc = e - pr * PCW;
      const int y = y0 - 1 + pr, x = x0 - 1 + pc;
      ok = (e < BPOS) && (y >= 0) && (y < H) && (x >= 0) && (x < W);
      return (y >> up) * (W >> up) + (x >> up);
    }
    const int pix = p0 + e;
    ok = (e < BPOS) && (pix < H * W);
    return pix;
  };

  const float dlo = sc_act_lo(dyact), dhi = sc_act_hi(dyact);
  auto load_stage = [&](long t) {
    // 32-bit division (the host keeps the stage count below 2^31) and readfirstlane: the division runs on the VALU and leaves
    // its uniform results in VGPRs, which would drag every address of the stage into 64-bit lane arithmetic
    const int n = __builtin_amdgcn_readfirstlane((int)((unsigned)t / (unsigned)per_img));
    const int rem = __builtin_amdgcn_readfirstlane((int)((unsigned)t - (unsigned)n * (unsigned)per_img));
    int y0 = 0, x0 = 0, p0 = 0;
    if (KS == 3) { const int ty = __builtin_amdgcn_readfirstlane(rem / tiles_x); y0 = ty * SR; x0 = (rem - ty * tiles_x) * 32; }
    else p0 = rem * SR * 32;
    sn = n; sy0 = y0; sx0 = x0; sp0 = p0;
    const size_t HW = (size_t)H * W;
    {
      bool okp;
      const int po = a_pix(y0, x0, p0, okp);
      const int ch0 = cot * COT + a_c0;
      const size_t base = ((size_t)n * p.Cout + ch0) * HW + (okp ? po : 0);
      const float* gp = p.dy.x + base;
      const float* yp = (dymode == SC_SRC_BNBWD) ? p.dy.aux + base : gp;
#pragma unroll
      for (int i = 0; i < NAI; ++i) {
        const size_t o = (ch0 + ACH * i < p.Cout) ? (size_t)(ACH * i) * HW : 0;     // clamped, unconditional loads
        ag[i] = gp[o];
        ay[i] = yp[o];
      }
    }
    {
      int off0[NBJ], off1[NBJ];
#pragma unroll
      for (int j = 0; j < NBJ; ++j) {
        bool ok;
        const int o0 = b_pos(j, y0, x0, p0, p.s0.up, ok);
        off0[j] = ok ? o0 : 0;
        const int o1 = b_pos(j, y0, x0, p0, p.s1.up, ok);
        off1[j] = ok ? o1 : 0;
      }
#pragma unroll
      for (int k = 0; k < NBK; ++k) {
        const int chr = cit * CIT + b_c0 + 8 * k;
        const int ch = chr < p.Cin ? chr : 0;
        const bool second = ch >= C0;
        const int cs = second ? ch - C0 : ch;
        const int Cs = second ? p.s1.C : p.s0.C;
        const int up = second ? p.s1.up : p.s0.up;
        const float* xp = (second ? p.s1.x : p.s0.x) + ((size_t)n * Cs + cs) * ((size_t)(H >> up) * (W >> up));
#pragma unroll
        for (int j = 0; j < NBJ; ++j) bx[k][j] = xp[second ? off1[j] : off0[j]];
      }
    }
  };

  auto store_stage = [&]() {
    {
      bool okp;
      (void)a_pix(sy0, sx0, sp0, okp);
      if (dymode == SC_SRC_BNBWD) {
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
          const int chl = a_c0 + ACH * i;
          const float4 c0 = *reinterpret_cast<const float4*>(&s_ca[chl * SC_CST]);
          const float v = sc_pro_bnbwd(ag[i], ay[i], c0.x, c0.y, c0.z, c0.w, s_ca[chl * SC_CST + 4], dlo, dhi);
          s_a[chl * PA + a_px] = (okp && (cot * COT + chl < p.Cout)) ? v : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
          const int chl = a_c0 + ACH * i;
          const float2 c0 = *reinterpret_cast<const float2*>(&s_ca[chl * SC_CST]);
          const float v = sc_pro_affine(ag[i], c0.x, c0.y, dlo, dhi);
          s_a[chl * PA + a_px] = (okp && (cot * COT + chl < p.Cout)) ? v : 0.f;
        }
      }
    }
    bool okj[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) (void)b_pos(j, sy0, sx0, sp0, 0, okj[j]);
#pragma unroll
    for (int k = 0; k < NBK; ++k) {
      const int chl = b_c0 + 8 * k;
      const int ch = cit * CIT + chl;
      const bool second = ch >= C0;
      const int act = second ? p.s1.act : p.s0.act;
      const float lo = sc_act_lo(act), hi = sc_act_hi(act);
      const float2 c0 = *reinterpret_cast<const float2*>(&s_cb[chl * SC_CST]);
      const bool okc = ch < p.Cin;
#pragma unroll
      for (int j = 0; j < NBJ; ++j) {
        const int e = b_q + 32 * j;
        if (e < BPOS) s_b[chl * PB + e] = (okj[j] && okc) ? sc_pro_affine(bx[k][j], c0.x, c0.y, lo, hi) : 0.f;
      }
    }
  };

  if (t_begin < t_end) {
    load_stage(t_begin);
    __syncthreads();          // constants in LDS
    store_stage();
    __syncthreads();
  }
  for (long t = t_begin; t < t_end; ++t) {
    const bool more = (t + 1) < t_end;
    if (more) load_stage(t + 1);
    // ---- MFMA on the staged tile ----
#pragma unroll
    for (int rr0 = 0; rr0 < RW; ++rr0) {
      const int rr = wk * RW + rr0;
      if constexpr (KS == 1) {
        // all 32 operand reads of the row first, then its 16 MFMAs (otherwise: read, wait lgkmcnt(0), MFMA per K step)
        float av[16], bv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          av[q] = s_a[(wm * 32 + l31) * PA + rr * 32 + 2 * q + lhi];
          bv[q] = s_b[(wn * 32 + l31) * PB + rr * 32 + 2 * q + lhi];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc[0], 0, 0, 0);
        continue;
      }
#pragma unroll 4
      for (int q = 0; q < 16; ++q) {
        const float a = s_a[(wm * 32 + l31) * PA + rr * 32 + 2 * q + lhi];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          float b;
          if (KS == 3) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            b = s_b[(wn * 32 + l31) * PB + (rr + kh) * PCW + 2 * q + lhi + kw];
          } else {
            b = s_b[(wn * 32 + l31) * PB + rr * 32 + 2 * q + lhi];
          }
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tap], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) store_stage();
    __syncthreads();
  }
  // 1x1: the WK pixel-row parts of a work-group are summed through LDS (fixed order) before the store: one partial per K slice
  // instead of WK -- half / a quarter of the 0.5 GB of partials the pointwise layers wrote and the batched reduction re-read per step
  constexpr int WKP = (KS == 1) ? 1 : WK;      // partials per work-group
  if constexpr (KS == 1 && WK > 1) {
    constexpr int NPAIR = WM * WN;
    static_assert((WK - 1) * NPAIR * 1024 <= COT * PA, "reduction scratch must fit the dy tile");
    float* red = s_a;                          // (the loop ended with a barrier: the staging tiles are free)
    const int pairw = wave % NPAIR;
    if (wk > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wk - 1) * NPAIR + pairw) * 1024 + r * 64 + lane] = acc[0][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int k = 1; k < WK; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] += red[((k - 1) * NPAIR + pairw) * 1024 + r * 64 + lane];
  }
  // ---- partial store: part[((slice*WKP + wk)*TAPS + tap)*CoP*CiP + co*CiP + ci] ----
  const int ci = cit * CIT + wn * 32 + l31;
  const size_t plane = (size_t)p.CoP * p.CiP;
  float* pb = p.part + ((size_t)blockIdx.x * WKP + (KS == 1 ? 0 : wk)) * TAPS * plane;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * COT + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (co < p.CoP && ci < p.CiP) pb[tap * plane + (size_t)co * p.CiP + ci] = acc[tap][r];
    }
  }
}

// Thin layers (Cout <= 16, ks = 3): weight gradient on v_mfma_f32_16x16x4_f32.  D[co][ci] per tap and ci block;
// A: lane -> dy[co = l&15][px = 4q + (l>>4)],  B: lane -> in[ci = 16*cb + (l&15)][px + d(tap)];  D row = 4*(l>>4)+r (co),
// col = l&15 (ci).  Stage = 4 rows x 32 px, wave w owns row w (its own K slice), LDS pitches == 2 (mod 32): conflict-free.
template <int NCB>
__global__ __launch_bounds__(256, 2) void k_wgrad_mfma16(const WgradP p) {
  constexpr int TAPS = 9, SR = 4, COT = 16, CIT = 16 * NCB;
  constexpr int PA = 130;
  constexpr int PRW = 6, PCW = 34, BPOS = PRW * PCW, PB = 226;
  constexpr int APX = 128, ACH = 2, NAI = COT / ACH;
  constexpr int NBJ = (BPOS + 31) / 32, NBK = CIT / 8;

  __shared__ float s_a[COT * PA];
  __shared__ float s_b[CIT * PB];
  __shared__ __attribute__((aligned(16))) float s_ca[COT * SC_CST];
  __shared__ __attribute__((aligned(16))) float s_cb[CIT * SC_CST];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int cit = blockIdx.y;
  const int H = p.H, W = p.W;
  const int C0 = p.s0.C;

  for (int i = tid; i < COT * SC_CST; i += 256) {
    const int ch = i / SC_CST;
    s_ca[i] = (p.dy.cst && p.dy.mode != SC_SRC_RAW && ch < p.Cout) ? p.dy.cst[(size_t)ch * SC_CST + (i % SC_CST)] : ((i % SC_CST) == 0 ? 1.f : 0.f);
  }
  for (int i = tid; i < CIT * SC_CST; i += 256) {
    const int ch = cit * CIT + i / SC_CST;
    float v = (i % SC_CST) == 0 ? 1.f : 0.f;
    if (ch < p.Cin) {
      const bool second = ch >= C0;
      const float* cp = second ? p.s1.cst : p.s0.cst;
      const int md = second ? p.s1.mode : p.s0.mode;
      if (cp && md != SC_SRC_RAW) v = cp[(size_t)(second ? ch - C0 : ch) * SC_CST + (i % SC_CST)];
    }
    s_cb[i] = v;
  }

  floatx4 acc[TAPS][NCB];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int b = 0; b < NCB; ++b) acc[t][b] = (floatx4){0.f, 0.f, 0.f, 0.f};

  const int tiles_x = (W + 31) >> 5;
  const int per_img = tiles_x * ((H + SR - 1) / SR);
  const long T = (long)p.N * per_img;
  const long t_begin = T * blockIdx.x / p.nsl, t_end = T * (blockIdx.x + 1) / p.nsl;
  const int dymode = p.dy.mode;
  const float dlo = sc_act_lo(p.dy.act), dhi = sc_act_hi(p.dy.act);

  const int a_px = tid % APX, a_c0 = tid / APX;
  const int b_q = tid & 31, b_c0 = tid >> 5;
  float ag[NAI], ay[NAI], bx[NBK][NBJ];
  int sy0 = 0, sx0 = 0;

  auto b_pos = [&](int j, int y0, int x0, int up, bool& ok) -> int {
    const int e = b_q + 32 * j;
    const int pr = e / PCW, pc = e - pr * PCW;
    const int y = y0 - 1 + pr, x = x0 - 1 + pc;
    ok = (e < BPOS) && (y >= 0) && (y < H) && (x >= 0) && (x < W);
    return (y >> up) * (W >> up) + (x >> up);
  };

  auto load_stage = [&](long t) {
    const int n = (int)(t / per_img);
    const int rem = (int)(t - (long)n * per_img);
    const int ty = rem / tiles_x;
    const int y0 = ty * SR, x0 = (rem - ty * tiles_x) * 32;
    sy0 = y0; sx0 = x0;
    const size_t HW = (size_t)H * W;
    {
      const int y = y0 + (a_px >> 5), x = x0 + (a_px & 31);
      const bool okp = (y < H) && (x < W);
      const size_t base = ((size_t)n * p.Cout + a_c0) * HW + (okp ? y * W + x : 0);
      const float* gp = p.dy.x + base;
      const float* yp = (dymode == SC_SRC_BNBWD) ? p.dy.aux + base : gp;
#pragma unroll
      for (int i = 0; i < NAI; ++i) {
        const size_t o = (a_c0 + ACH * i < p.Cout) ? (size_t)(ACH * i) * HW : 0;
        ag[i] = gp[o]; ay[i] = yp[o];
      }
    }
    int off0[NBJ], off1[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      bool ok;
      const int o0 = b_pos(j, y0, x0, p.s0.up, ok);
      off0[j] = ok ? o0 : 0;
      const int o1 = b_pos(j, y0, x0, p.s1.up, ok);
      off1[j] = ok ? o1 : 0;
    }
#pragma unroll
    for (int k = 0; k < NBK; ++k) {
      const int chr = cit * CIT + b_c0 + 8 * k;
      const int ch = chr < p.Cin ? chr : 0;
      const bool second = ch >= C0;
      const int cs = second ? ch - C0 : ch;
      const int Cs = second ? p.s1.C : p.s0.C;
      const int up = second ? p.s1.up : p.s0.up;
      const float* xp = (second ? p.s1.x : p.s0.x) + ((size_t)n * Cs + cs) * ((size_t)(H >> up) * (W >> up));
#pragma unroll
      for (int j = 0; j < NBJ; ++j) bx[k][j] = xp[second ? off1[j] : off0[j]];
    }
  };

  auto store_stage = [&]() {
    {
      const int y = sy0 + (a_px >> 5), x = sx0 + (a_px & 31);
      const bool okp = (y < H) && (x < W);
#pragma unroll
      for (int i = 0; i < NAI; ++i) {
        const int chl = a_c0 + ACH * i;
        const float4 c0 = *reinterpret_cast<const float4*>(&s_ca[chl * SC_CST]);
        const float v = (dymode == SC_SRC_BNBWD) ? sc_pro_bnbwd(ag[i], ay[i], c0.x, c0.y, c0.z, c0.w, s_ca[chl * SC_CST + 4], dlo, dhi)
                                                 : sc_pro_affine(ag[i], c0.x, c0.y, dlo, dhi);
        s_a[chl * PA + a_px] = (okp && chl < p.Cout) ? v : 0.f;
      }
    }
    bool okj[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) (void)b_pos(j, sy0, sx0, 0, okj[j]);
#pragma unroll
    for (int k = 0; k < NBK; ++k) {
      const int chl = b_c0 + 8 * k;
      const int ch = cit * CIT + chl;
      const bool second = ch >= C0;
      const int act = second ? p.s1.act : p.s0.act;
      const float lo = sc_act_lo(act), hi = sc_act_hi(act);
      const float2 c0 = *reinterpret_cast<const float2*>(&s_cb[chl * SC_CST]);
      const bool okc = ch < p.Cin;
#pragma unroll
      for (int j = 0; j < NBJ; ++j) {
        const int e = b_q + 32 * j;
        if (e < BPOS) s_b[chl * PB + e] = (okj[j] && okc) ? sc_pro_affine(bx[k][j], c0.x, c0.y, lo, hi) : 0.f;
      }
    }
  };

  if (t_begin < t_end) {
    load_stage(t_begin);
    __syncthreads();
    store_stage();
    __syncthreads();
  }
  for (long t = t_begin; t < t_end; ++t) {
    const bool more = (t + 1) < t_end;
    if (more) load_stage(t + 1);
    const int rr = wave;
#pragma unroll 2
    for (int q = 0; q < 8; ++q) {
      const float a = s_a[l15 * PA + rr * 32 + 4 * q + lq];
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const float b = s_b[(cb * 16 + l15) * PB + (rr + kh) * PCW + 4 * q + lq + kw];
          acc[tap][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tap][cb], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) store_stage();
    __syncthreads();
  }
  const size_t plane = (size_t)p.CoP * p.CiP;
  float* pb = p.part + ((size_t)blockIdx.x * 4 + wave) * TAPS * plane;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * lq + r, ci = cit * CIT + cb * 16 + l15;
        if (ci < p.CiP) pb[tap * plane + (size_t)co * p.CiP + ci] = acc[tap][cb][r];
      }
}

// dw[co][ci][tap] = sum_s part[s][tap][co][ci]
__global__ void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int nparts, int taps,
                               int Cout, int Cin, int CoP, int CiP) {
  const size_t total = (size_t)taps * Cout * Cin;
  const size_t plane = (size_t)CoP * CiP;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    size_t r = i / Cin;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout);
    const float* src = part + (size_t)tap * plane + (size_t)co * CiP + ci;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += src[(size_t)k * taps * plane];
    dw[((size_t)co * Cin + ci) * taps + tap] = s;
  }
}

struct WgradPlan { int wm, wn, wk, sr, nsl, CoP, CiP, co_tiles, ci_tiles; long stages; };

WgradPlan plan_wgrad(int N, int H, int W, int Cout, int Cin, int ks) {
  WgradPlan pl;
  if (ks == 3 && Cout <= 16) {          // thin layers: k_wgrad_mfma16<NCB>, 4 waves = 4 K slices
    pl.wm = 0; pl.wn = Cin > 16 ? 2 : 1; pl.wk = 4; pl.sr = 4;
    pl.CoP = 32; pl.CiP = (Cin + 31) / 32 * 32;
    pl.co_tiles = 1; pl.ci_tiles = (Cin + 16 * pl.wn - 1) / (16 * pl.wn);
    pl.stages = (long)N * ((W + 31) / 32) * ((H + 3) / 4);
    long want = 512 / pl.ci_tiles;       // x4 K parts per work-group: 2048 partial rows (8192 rows cost 0.6 GB of partial traffic per step)
    if (want < 1) want = 1;
    if (want > pl.stages) want = pl.stages;
    pl.nsl = (int)want;
    return pl;
  }
  if (Cout > 32 && Cin > 32) { pl.wm = 2; pl.wn = 2; }
  else if (Cout > 32) { pl.wm = 2; pl.wn = 1; }
  else if (Cin > 32) { pl.wm = 1; pl.wn = 2; }
  else { pl.wm = 1; pl.wn = 1; }
  pl.wk = 4 / (pl.wm * pl.wn);
  pl.sr = (pl.wk == 4) ? 4 : 2;
  pl.CoP = (Cout + 31) / 32 * 32;
  pl.CiP = (Cin + 31) / 32 * 32;
  pl.co_tiles = (Cout + 32 * pl.wm - 1) / (32 * pl.wm);
  pl.ci_tiles = (Cin + 32 * pl.wn - 1) / (32 * pl.wn);
  if (ks == 3) pl.stages = (long)N * ((W + 31) / 32) * ((H + pl.sr - 1) / pl.sr);
  else pl.stages = (long)N * (((long)H * W + pl.sr * 32 - 1) / (pl.sr * 32));
  constexpr long wgs_env = 512L;       // work-groups per launch aimed at: 512 = one full round at two per CU (measured 1.65 vs 1.84 ms at 1024, 1.96 at 768, 2.1 at 256 or 2048)
  long want = wgs_env / ((long)pl.co_tiles * pl.ci_tiles);
  if (want < 1) want = 1;
  if (want > pl.stages) want = pl.stages;
  if (want > 512) want = 512;
  pl.nsl = (int)want;
  return pl;
}

}  // namespace

extern "C" size_t sc_packed_weight_floats(int Cout, int Cin, int ks, int co_t, int transpose_flip) {
  const int M = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  const size_t mt = (M + co_t - 1) / co_t;
  const int kc = ks == 3 ? 8 : 16;
  const size_t Kpad = (size_t)(K + kc - 1) / kc * kc;
  return mt * Kpad * ks * ks * co_t;
}

extern "C" int sc_pack_weights(const float* w, float* wpk, int Cout, int Cin, int ks, int co_t,
                               int transpose_flip, sc_stream stream) {
  SC_REQUIRE(ks == 1 || ks == 3, "sc_pack_weights: ks must be 1 or 3 (got %d)", ks);
  SC_REQUIRE(co_t == 16 || co_t == 32 || co_t == 64, "sc_pack_weights: co_t must be 16, 32 or 64 (got %d)", co_t);
  const size_t total = sc_packed_weight_floats(Cout, Cin, ks, co_t, transpose_flip);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  const int Kg = transpose_flip ? Cout : Cin, kc = ks == 3 ? 8 : 16;
  hipLaunchKernelGGL(k_pack_weights, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpk, Cout, Cin,
                     ks * ks, co_t, transpose_flip, (Kg + kc - 1) / kc * kc, total);
  SC_LAUNCH_OK("sc_pack_weights");
  return SC_OK;
}

// [SC_IDENTITY_MAXC][SC_CST] floats on the current device: scale 1, everything else 0.  Allocated and filled on first use (an eager
// call: every capture is preceded by warm-up steps), one table per device of the process.
static const float* sc_identity_cst(int C) {
  constexpr int SC_IDENTITY_MAXC = 4096, MAXDEV = 16;
  static float* table[MAXDEV] = {};
  int dev = 0;
  if (C > SC_IDENTITY_MAXC || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  if (!table[dev]) {
    std::vector<float> h((size_t)SC_IDENTITY_MAXC * SC_CST, 0.f);
    for (int c = 0; c < SC_IDENTITY_MAXC; ++c) h[(size_t)c * SC_CST] = 1.f;
    float* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    table[dev] = d;
  }
  return table[dev];
}

const float* sc_identity_cst_table(int C) { return sc_identity_cst(C); }

extern "C" int sc_conv2d_mfma(const sc_conv_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_conv2d_mfma: null args");
  SC_REQUIRE(a->ks == 1 || a->ks == 3, "sc_conv2d_mfma: ks must be 1 or 3 (got %d)", a->ks);
  SC_REQUIRE(a->co_t == 16 || a->co_t == 32 || a->co_t == 64, "sc_conv2d_mfma: co_t must be 16, 32 or 64 (got %d)", a->co_t);
  SC_REQUIRE(a->co_t != 16 || (a->ks == 3 && a->Cout <= 16 && a->csplit == a->Cout),
             "sc_conv2d_mfma: co_t=16 is the thin-layer kernel: ks=3, Cout<=16, single output");
  SC_REQUIRE(a->nsrc == 1 || a->nsrc == 2, "sc_conv2d_mfma: nsrc must be 1 or 2");
  const int C0 = a->src[0].C, C1 = a->nsrc == 2 ? a->src[1].C : 0;
  SC_REQUIRE(C0 > 0 && C0 % 8 == 0 && C1 % 8 == 0, "sc_conv2d_mfma: source channels must be multiples of 8 (got %d,%d)", C0, C1);
  SC_REQUIRE(a->nsrc == 1 || a->ks == 3, "sc_conv2d_mfma: a channel concat of two sources needs ks=3");
  SC_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cout > 0, "sc_conv2d_mfma: bad shape");
  SC_REQUIRE(a->csplit > 0 && a->csplit <= a->Cout, "sc_conv2d_mfma: bad csplit");
  SC_REQUIRE(a->csplit == a->Cout || (a->add0 == nullptr && a->add1 == nullptr), "sc_conv2d_mfma: add tensors need a single output");
  for (int s = 0; s < a->nsrc; ++s) {
    SC_REQUIRE(a->src[s].up == 0 || (a->ks == 3 && a->H % 2 == 0 && a->W % 2 == 0), "sc_conv2d_mfma: upsampled source needs ks=3 and even H,W");
    SC_REQUIRE(a->src[s].mode == SC_SRC_RAW || a->src[s].cst != nullptr, "sc_conv2d_mfma: source %d needs constants", s);
    SC_REQUIRE(a->src[s].mode != SC_SRC_BNBWD || a->src[s].aux != nullptr, "sc_conv2d_mfma: BNBWD source needs aux");
    SC_REQUIRE(a->src[s].mode != SC_SRC_BNBWD || a->nsrc == 1, "sc_conv2d_mfma: a BNBWD source cannot be part of a concat");
  }
  if (a->ks == 1) {
    // large planes, forward: the streaming kernel (k_pw_stream) where its shape conditions hold
    static const bool pws_off = [] { const char* e = getenv("STARCOP_PWS"); return e && atoi(e) == 0; }();      // (same-box A/B)
    const long HWl = (long)a->H * a->W;
    const int K = a->src[0].C, M = a->Cout;
    const bool bnb = a->src[0].mode == SC_SRC_BNBWD;      // data gradients of the projections: few dy channels, many outputs; no bnr epilogue here
    const bool plain = a->nsrc == 1 && (a->src[0].mode == SC_SRC_RAW || a->src[0].mode == SC_SRC_AFFINE || (bnb && a->bnr == nullptr && a->stats == nullptr)) &&
                       a->src[0].up == 0 && a->csplit == a->Cout && !a->accum0 && a->add0 == nullptr && a->add1 == nullptr && a->out0 != nullptr;
    const bool aligned = (((uintptr_t)a->src[0].x | (uintptr_t)a->out0 | (bnb ? (uintptr_t)a->src[0].aux : 0)) & 15) == 0;
    const int nks = K / 4;
    // short contractions only: the long-K projections (96 / 144 -> 24 at 128^2) measured a tie without and 3-8 us slower with the
    // statistics epilogue (a lane keeps NKS filter registers per cout block: two waves per SIMD)
    const bool ks_ok = bnb ? (nks == 4 || nks == 6 || nks == 8) : (nks == 6 || nks == 8);
    if (!pws_off && plain && aligned && ks_ok && (M <= 32 || (M <= 192 && nks <= 8)) && HWl >= 4096 && HWl % 256 == 0 && HWl < (1L << 30) && a->co_t != 16) {
      PwsP q;
      q.x = a->src[0].x; q.aux = a->src[0].aux; q.cst = a->src[0].mode == SC_SRC_RAW ? nullptr : a->src[0].cst;
      q.act = a->src[0].mode == SC_SRC_RAW ? (int)SC_ACT_NONE : a->src[0].act;
      q.wpk = a->wpk; q.co_t = a->co_t; q.Kpad = (K + 15) / 16 * 16;
      q.out = a->out0; q.stats = a->stats; q.HW = (int)HWl; q.K = K; q.M = M;
      hipStream_t st = (hipStream_t)stream;
      const long groups = (long)a->N * HWl / 64;
#define SC_PWS(NCB_, NKS_, CP_) hipLaunchKernelGGL((k_pw_stream<NCB_, NKS_, CP_>), dim3((unsigned)(groups / (4 / CP_))), dim3(256), 0, st, q)
#define SC_PWS_K(NCB_, CP_) do { if (nks == 6) SC_PWS(NCB_, 6, CP_); else SC_PWS(NCB_, 8, CP_); } while (0)
      if (bnb) {
#define SC_PWSB(NCB_, CP_) do { if (nks == 4) hipLaunchKernelGGL((k_pw_stream<NCB_, 4, CP_, true>), dim3((unsigned)(groups / (4 / CP_))), dim3(256), 0, st, q); \
                                else if (nks == 6) hipLaunchKernelGGL((k_pw_stream<NCB_, 6, CP_, true>), dim3((unsigned)(groups / (4 / CP_))), dim3(256), 0, st, q); \
                                else hipLaunchKernelGGL((k_pw_stream<NCB_, 8, CP_, true>), dim3((unsigned)(groups / (4 / CP_))), dim3(256), 0, st, q); } while (0)
        if (M <= 32) SC_PWSB(2, 1);
        else if (M <= 96) SC_PWSB(3, 2);
        else if (M <= 160) SC_PWSB(5, 2);
        else SC_PWSB(6, 2);
#undef SC_PWSB
      }
      else if (M <= 16) SC_PWS_K(1, 1);
      else if (M <= 32) SC_PWS_K(2, 1);
      else if (M <= 160) SC_PWS_K(5, 2);                   // (the expansions: few input channels, up to 160 / 192 couts in two parts)
      else SC_PWS_K(6, 2);
#undef SC_PWS_K
#undef SC_PWS
      SC_LAUNCH_OK("sc_conv2d_mfma(k_pw_stream)");
      return SC_OK;
    }
  }
  ConvP p;
  p.s0 = to_srcd(a->src[0]);
  p.s1 = a->nsrc == 2 ? to_srcd(a->src[1]) : empty_srcd();
  if (a->ks == 1) {
    // RAW sources of the 1x1 kernel read identity constants (scale 1, shift 0): see load_chunk
    if (p.s0.mode == SC_SRC_RAW) { p.s0.cst = sc_identity_cst(p.s0.C); SC_REQUIRE(p.s0.cst != nullptr, "sc_conv2d_mfma: identity constants unavailable (C = %d)", p.s0.C); }
    if (a->nsrc == 2 && p.s1.mode == SC_SRC_RAW) { p.s1.cst = sc_identity_cst(p.s1.C); SC_REQUIRE(p.s1.cst != nullptr, "sc_conv2d_mfma: identity constants unavailable (C = %d)", p.s1.C); }
  }
  p.wpk = a->wpk; p.N = a->N; p.H = a->H; p.W = a->W; p.Cout = a->Cout;
  p.out0 = a->out0; p.out1 = a->out1; p.csplit = a->csplit; p.accum0 = a->accum0; p.accum1 = a->accum1;
  p.add0 = a->add0; p.add1 = a->add1; p.stats = a->stats;
  const int co_tiles = (a->Cout + a->co_t - 1) / a->co_t;
  dim3 grid;
  if (a->ks == 3) grid = dim3(((a->W + 31) / 32) * ((a->H + 3) / 4), co_tiles, a->N);
  else grid = dim3((a->H * a->W + 127) / 128, co_tiles, a->N);
  hipStream_t st = (hipStream_t)stream;
  p.xcdmap = 0;
  constexpr int xcdmap_env = 0;   // measured: no gain for the 1x1 layers (2.82 vs 2.89 ms per step): off
  if (a->co_t != 16 && xcdmap_env) {
    const long total = (long)grid.x * a->N, per_xcd = (total + 7) / 8;
    SC_REQUIRE(per_xcd * 8 * co_tiles < (1L << 31), "sc_conv2d_mfma: grid too large");
    grid = dim3((unsigned)(per_xcd * 8 * co_tiles));
    p.xcdmap = 1;
  }
  if (a->co_t == 16) {
    const bool bnb = a->src[0].mode == SC_SRC_BNBWD;
    if (a->H >= 8) {          // 8-row tiles (two rows per wave): measured 0.58 -> 0.49 ms on 32->16 channels at 512^2
      dim3 g2(((a->W + 31) / 32) * ((a->H + 7) / 8), 1, a->N);
      if (bnb) hipLaunchKernelGGL((k_conv_mfma16<true, 2>), g2, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((k_conv_mfma16<false, 2>), g2, dim3(256), 0, st, p);
    } else {
      if (bnb) hipLaunchKernelGGL((k_conv_mfma16<true, 1>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((k_conv_mfma16<false, 1>), grid, dim3(256), 0, st, p);
    }
  }
  else {
    const bool bnb = a->src[0].mode == SC_SRC_BNBWD;
#define SC_CM(KS_, RM_)                                                                               \
    do {                                                                                              \
      if (bnb) hipLaunchKernelGGL((k_conv_mfma<KS_, RM_, true>), grid, dim3(256), 0, st, p);          \
      else hipLaunchKernelGGL((k_conv_mfma<KS_, RM_, false>), grid, dim3(256), 0, st, p);             \
    } while (0)
    constexpr int pf_env = 1;          // K chunks of look-ahead (2 measured slower twice: 2.93 vs 2.78 ms)
#define SC_CM2(RM_)                                                                                  \
    do {                                                                                              \
      if (bnb) hipLaunchKernelGGL((k_conv_mfma<1, RM_, true, 2>), grid, dim3(256), 0, st, p);         \
      else hipLaunchKernelGGL((k_conv_mfma<1, RM_, false, 2>), grid, dim3(256), 0, st, p);            \
    } while (0)
    if (a->ks == 3 && a->co_t == 64) SC_CM(3, 2);
    else if (a->ks == 3) SC_CM(3, 1);
    else if (a->co_t == 64) { if (pf_env == 2) SC_CM2(2); else SC_CM(1, 2); }
    else { if (pf_env == 2) SC_CM2(1); else SC_CM(1, 1); }
#undef SC_CM2
#undef SC_CM
  }
  SC_LAUNCH_OK("sc_conv2d_mfma");
  return SC_OK;
}

extern "C" int sc_conv1x1_ksplit(const sc_conv_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_conv1x1_ksplit: null args");
  SC_REQUIRE(a->ks == 1 && a->nsrc == 1, "sc_conv1x1_ksplit: ks must be 1 with a single source");
  SC_REQUIRE(a->co_t == 32 || a->co_t == 64, "sc_conv1x1_ksplit: co_t must be 32 or 64 (got %d)", a->co_t);
  SC_REQUIRE(a->src[0].C > 0 && a->src[0].up == 0, "sc_conv1x1_ksplit: bad source");
  SC_REQUIRE(a->src[0].mode != SC_SRC_NORM, "sc_conv1x1_ksplit: NORM sources are the stem's");
  SC_REQUIRE(a->src[0].mode == SC_SRC_RAW || a->src[0].cst != nullptr, "sc_conv1x1_ksplit: source needs constants");
  SC_REQUIRE(a->src[0].mode != SC_SRC_BNBWD || a->src[0].aux != nullptr, "sc_conv1x1_ksplit: BNBWD source needs aux");
  SC_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cout > 0, "sc_conv1x1_ksplit: bad shape");
  SC_REQUIRE(a->csplit > 0 && a->csplit <= a->Cout, "sc_conv1x1_ksplit: bad csplit");
  SC_REQUIRE(a->csplit == a->Cout || (a->add0 == nullptr && a->add1 == nullptr), "sc_conv1x1_ksplit: add tensors need a single output");
  ConvP p;
  p.xcdmap = 0;
  p.s0 = to_srcd(a->src[0]); p.s1 = empty_srcd();
  if (p.s0.mode == SC_SRC_RAW) { p.s0.cst = sc_identity_cst(p.s0.C); SC_REQUIRE(p.s0.cst != nullptr, "sc_conv1x1_ksplit: identity constants unavailable (C = %d)", p.s0.C); }
  p.wpk = a->wpk; p.N = a->N; p.H = a->H; p.W = a->W; p.Cout = a->Cout;
  p.out0 = a->out0; p.out1 = a->out1; p.csplit = a->csplit; p.accum0 = a->accum0; p.accum1 = a->accum1;
  p.add0 = a->add0; p.add1 = a->add1; p.stats = a->stats;
  dim3 grid((a->H * a->W + 31) / 32, (a->Cout + a->co_t - 1) / a->co_t, a->N);
  hipStream_t st = (hipStream_t)stream;
  const bool bnb = a->src[0].mode == SC_SRC_BNBWD;
  if (a->co_t == 64 && bnb) hipLaunchKernelGGL((k_conv1_ksplit<2, true>), grid, dim3(256), 0, st, p);
  else if (a->co_t == 64) hipLaunchKernelGGL((k_conv1_ksplit<2, false>), grid, dim3(256), 0, st, p);
  else if (bnb) hipLaunchKernelGGL((k_conv1_ksplit<1, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_conv1_ksplit<1, false>), grid, dim3(256), 0, st, p);
  SC_LAUNCH_OK("sc_conv1x1_ksplit");
  return SC_OK;
}

extern "C" size_t sc_wgrad_workspace_floats(int N, int H, int W, int Cout, int Cin, int ks) {
  const WgradPlan pl = plan_wgrad(N, H, W, Cout, Cin, ks);
  const size_t E = (size_t)ks * ks * pl.CoP * pl.CiP;
  const int nparts = pl.nsl * (ks == 1 ? 1 : pl.wk);      // 1x1: the K parts of a work-group are summed in the kernel
  return (size_t)nparts * E + sc_reduce_scratch_floats(nparts, E);
}

static int wgrad_mfma_launch(const sc_wgrad_args* a, sc_stream stream, sc_wgrad_pending* pending);

extern "C" int sc_conv2d_wgrad_mfma(const sc_wgrad_args* a, sc_stream stream) { return wgrad_mfma_launch(a, stream, nullptr); }

extern "C" int sc_conv2d_wgrad_mfma_deferred(const sc_wgrad_args* a, sc_wgrad_pending* pending, sc_stream stream) {
  SC_REQUIRE(pending != nullptr, "sc_conv2d_wgrad_mfma_deferred: null descriptor");
  return wgrad_mfma_launch(a, stream, pending);
}

static int wgrad_mfma_launch(const sc_wgrad_args* a, sc_stream stream, sc_wgrad_pending* pending) {
  SC_REQUIRE(a != nullptr, "sc_conv2d_wgrad_mfma: null args");
  SC_REQUIRE(a->ks == 1 || a->ks == 3, "sc_conv2d_wgrad_mfma: ks must be 1 or 3");
  SC_REQUIRE(a->nsrc == 1 || a->nsrc == 2, "sc_conv2d_wgrad_mfma: nsrc must be 1 or 2");
  const int C0 = a->src[0].C, C1 = a->nsrc == 2 ? a->src[1].C : 0;
  SC_REQUIRE(C0 + C1 == a->Cin, "sc_conv2d_wgrad_mfma: source channels (%d+%d) != Cin %d", C0, C1, a->Cin);
  SC_REQUIRE(a->dy.C == a->Cout, "sc_conv2d_wgrad_mfma: dy channels %d != Cout %d", a->dy.C, a->Cout);
  SC_REQUIRE(a->dy.up == 0, "sc_conv2d_wgrad_mfma: dy cannot be upsampled");
  SC_REQUIRE(a->dy.mode != SC_SRC_BNBWD || a->dy.aux != nullptr, "sc_conv2d_wgrad_mfma: BNBWD dy needs aux");
  for (int s = 0; s < a->nsrc; ++s) {
    SC_REQUIRE(a->src[s].mode != SC_SRC_BNBWD, "sc_conv2d_wgrad_mfma: input sources cannot be BNBWD");
    SC_REQUIRE(a->src[s].up == 0 || (a->ks == 3 && a->H % 2 == 0 && a->W % 2 == 0), "sc_conv2d_wgrad_mfma: upsampled source needs ks=3, even H,W");
  }
  const WgradPlan pl = plan_wgrad(a->N, a->H, a->W, a->Cout, a->Cin, a->ks);
  SC_REQUIRE(pl.stages < (1L << 31), "sc_conv2d_wgrad_mfma: too many stages (%ld) for the kernel's 32-bit stage arithmetic", pl.stages);
  const size_t need = sc_wgrad_workspace_floats(a->N, a->H, a->W, a->Cout, a->Cin, a->ks);
  SC_REQUIRE(a->part_floats >= need, "sc_conv2d_wgrad_mfma: workspace too small (%zu < %zu floats)", a->part_floats, need);
  WgradP p;
  p.dy = to_srcd(a->dy); p.s0 = to_srcd(a->src[0]); p.s1 = a->nsrc == 2 ? to_srcd(a->src[1]) : empty_srcd();
  p.N = a->N; p.H = a->H; p.W = a->W; p.Cout = a->Cout; p.Cin = a->Cin; p.part = a->part;
  p.nsl = pl.nsl; p.CoP = pl.CoP; p.CiP = pl.CiP;
  dim3 grid(pl.nsl, pl.ci_tiles, pl.co_tiles);
  hipStream_t st = (hipStream_t)stream;
#define SC_WG(KS, WM, WN) hipLaunchKernelGGL((k_wgrad_mfma<KS, WM, WN>), grid, dim3(256), 0, st, p)
  if (pl.wm == 0) {
    if (pl.wn == 2) hipLaunchKernelGGL((k_wgrad_mfma16<2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_wgrad_mfma16<1>), grid, dim3(256), 0, st, p);
  } else if (a->ks == 3) {
    if (pl.wm == 2 && pl.wn == 2) SC_WG(3, 2, 2);
    else if (pl.wm == 2) SC_WG(3, 2, 1);
    else if (pl.wn == 2) SC_WG(3, 1, 2);
    else SC_WG(3, 1, 1);
  } else {
    if (pl.wm == 2 && pl.wn == 2) SC_WG(1, 2, 2);
    else if (pl.wm == 2) SC_WG(1, 2, 1);
    else if (pl.wn == 2) SC_WG(1, 1, 2);
    else SC_WG(1, 1, 1);
  }
#undef SC_WG
  SC_LAUNCH_OK("sc_conv2d_wgrad_mfma");
  if (pending) {          // the caller sums the K-slice partials of many layers in one launch (sc_wgrad_reduce_batch)
    pending->part = a->part; pending->dw = a->dw;
    pending->nparts = pl.nsl * (a->ks == 1 ? 1 : pl.wk); pending->taps = a->ks * a->ks;
    pending->Cout = a->Cout; pending->Cin = a->Cin; pending->CoP = pl.CoP; pending->CiP = pl.CiP;
    pending->total = (uint64_t)pending->taps * a->Cout * a->Cin;
    return SC_OK;
  }
  return sc_wgrad_finish(a->part, pl.nsl * (a->ks == 1 ? 1 : pl.wk), a->ks * a->ks, a->Cout, a->Cin, pl.CoP, pl.CiP, a->dw, st);
}

namespace {
// dw[co][ci][tap] = sum_k part[k][tap][co][ci] for every pending layer: one thread per weight, partial rows summed in a fixed
// order (bit-reproducible), 8 loads in flight
__global__ __launch_bounds__(256) void k_wgrad_reduce_batch(const sc_wgrad_pending* __restrict__ descs, const unsigned* __restrict__ starts, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const sc_wgrad_pending d = descs[lo];
  const size_t i = (size_t)(blockIdx.x - starts[lo]) * 256 + threadIdx.x;
  if (i >= d.total) return;
  const int ci = (int)(i % d.Cin);
  size_t r = i / d.Cin;
  const int co = (int)(r % d.Cout);
  const int tap = (int)(r / d.Cout);
  const size_t plane = (size_t)d.CoP * d.CiP, stride = (size_t)d.taps * plane;
  const float* src = d.part + (size_t)tap * plane + (size_t)co * d.CiP + ci;
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= d.nparts; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < d.nparts; ++k) s += src[(size_t)k * stride];
  d.dw[((size_t)co * d.Cin + ci) * d.taps + tap] = s;
}
}  // namespace

extern "C" int sc_wgrad_reduce_batch(const sc_wgrad_pending* descs_dev, const uint32_t* block_starts_dev, int n, uint32_t total_blocks,
                                     sc_stream stream) {
  SC_REQUIRE(descs_dev && block_starts_dev && n >= 0, "sc_wgrad_reduce_batch: bad argument");
  if (n == 0 || total_blocks == 0) return SC_OK;
  hipLaunchKernelGGL(k_wgrad_reduce_batch, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, block_starts_dev, n);
  SC_LAUNCH_OK("sc_wgrad_reduce_batch");
  return SC_OK;
}

// two-level reduction of the K-slice partials part[nparts][taps][CoP][CiP] (scratch follows them), then the layout
// change to OIHW
int sc_wgrad_finish(float* part, int nparts, int taps, int Cout, int Cin, int CoP, int CiP, float* dw, hipStream_t st) {
  const size_t E = (size_t)taps * CoP * CiP;
  const float* rows; int nrows;
  int rc = sc_reduce_rows_partial(part, nparts, E, part + (size_t)nparts * E, &rows, &nrows, 64, st);
  if (rc != SC_OK) return rc;
  const size_t total = (size_t)taps * Cout * Cin;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(blocks), dim3(256), 0, st, rows, dw, nrows, taps, Cout, Cin, CoP, CiP);
  SC_LAUNCH_OK("sc_wgrad_reduce");
  return SC_OK;
}
