// Dense 3x3 NCHW conv2d (stride 1, pad 1) with fp32-level accuracy on the 16-bit matrix cores of gfx950.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the 16-bit MFMA rate, so the decoder's 3x3 convolutions (88 % of the U-Net's MACs:
// smp.Unet built at starcop/models/model_module.py:244-251) are MFMA-bound on it.  Here every fp32 operand is split exactly
// into 16-bit terms while it is staged and the leading partial products are accumulated in fp32:
//   two fp16 terms (default, template HF):  a*s = h0 + h1 (2 x 11 significand bits, power-of-two range scale s divided out in
//       the epilogue, see split2h below), products h0*g1 + h1*g0 + h0*g0 = 3 x v_mfma_f32_32x32x16_f16 per 32x32x16 block
//   three bf16 terms (NT = 3):  a = a0 + a1 + a2 (3 x 8 bits, fp32's exponent range), the six products of weight >= 2^-24
//       a1*b1 + a2*b0 + a0*b2 + a1*b0 + a0*b1 + a0*b0 = 6 x v_mfma_f32_32x32x16_bf16 (192 cycles; the fp32 MFMA needs 512)
//   NT = 2 / NT = 1 with bf16: the opt-in reduced-accuracy modes of the network ("fp32-bwd2", "fp32-2", "bf16")
// Measured against fp64 both full-accuracy splits are as close as the fp32 MFMA path (tests/test_gpu_ops.py::test_conv_bx3_*,
// test_conv_two_fp16_terms).
//
// Same "normalise on load" contract as conv_mfma.hip: the producer's BatchNorm+activation (forward), or the
// BatchNorm/activation backward (dgrad), nearest x2 upsampling and the channel concat are applied while the tile is
// staged; the split happens in the same pass.  The same kernel computes dgrad from transposed+flipped filters.
//
// GEMM view:  D[co][pixel] = sum_{tap} sum_{ci} Wp[tap][co][ci] * patch[ci][pixel + d(tap)],  K step = 16 channels
//   A (32 x 16): lane l -> W[co = l&31][ci = 8*(l>>5) .. +7]      (one 16-byte LDS read)
//   B (16 x 32): lane l -> patch[ci = 8*(l>>5) .. +7][pixel l&31]  (one 16-byte LDS read)
//   D: col = l&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(l>>5) (cout)
// Work-group = 4 waves, tile = 8 rows x 32 cols x (32*Q couts); wave w owns rows 2w, 2w+1 and all Q cout blocks
// (2 x Q accumulators: every A read is used twice, every B read Q times).
// LDS: patch [terms][2 channel halves][10 x 34 pixels] x 16 B (single buffer, next chunk prefetched in registers),
// filters per filter row kh [terms][3 kw][2 halves][32Q couts] x 16 B, double buffered: 51 KB (two terms) / 72 KB (three terms)
// at Q = 2 -> 2 work-groups per CU (the Q = 2 kernels need 220 VGPRs).
#include "sc_common.h"
#include "conv_sp_pack.h"
#include <cstdlib>
#include <type_traits>


namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) unsigned int uintx4;
typedef __attribute__((ext_vector_type(8))) _Float16 halfx8;
typedef __attribute__((ext_vector_type(2))) _Float16 halfx2;

// ---- two fp16 terms ("H" mode, terms code SC_TERMS_F16X2 = 4) ----
// a*s = h0 + h1 exactly to 22 significand bits (round-to-nearest conversions), products h0*g0 + h0*g1 + h1*g0: the dropped
// h1*g1 is 2^-22 |a||b|, below the fp32 accumulation error of a K >= 288 reduction.  fp16 has 5 exponent bits, so every
// operand is brought into range by an exact power-of-two scale that the epilogue divides out again:
//   filters     x 2^8  (|w| < 255; the absolute error 2^-25 of a sub-normal second term is 2^-33 in filter units)
//   activations x 2    (BatchNorm-normalised, ReLU6-clipped or ReLU: |x| < 32752; absolute error floor 2^-26)
//   gradients   x 2^(5-e), 2^e >= the tensor's max |A_c g| from the BatchNorm-backward reduction (args->absmax): the
//               prologue output A g + B y + D is bounded by (2 + max|x_hat|) times that, so anything up to
//               max|x_hat| = 2045 (the most a 4M-sample channel can reach is 2048) fits; error floor 2^-30 of the maximum
// Values beyond the range are clamped to +-65504 (a finite error, never an inf/NaN); the three-term bf16 split
// ("fp32-x3") has fp32's exponent range and no such limits.
constexpr float SC_H_SW = 256.f, SC_H_SX = 2.f, SC_H_MAX = 65504.f;
__device__ __forceinline__ float h_grad_scale(const float* absmax) {
  const float M = absmax ? *absmax : 0.f;
  if (!(M > 0.f) || !(M < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(M, &e);                                  // M = m * 2^e, m in [0.5, 1)
  e = 5 - e;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.f, e);                                // M * s in [16, 32)
}
// Activation operand scale (round 5: range-safe by construction).  xb0 / xb1: device floats >= max |activation| of the staged
// source(s) -- in training the bound |gamma| sqrt(n - 1) + |beta| that sc_bn_finalize leaves per BatchNorm'd tensor (no normalised
// sample of n can exceed sqrt(n - 1)), for residual sums the maximum sc_add_srcs_absmax records in every forward, in inference
// the sticky record of the streamed maxima; NULL: bounded by ReLU6 / unknown -> the default.  The scale is the default 2 whenever
// 2 M <= 32752 (bit-identical to the fixed scale of rounds 1-4) and the largest power of two with s M <= 32752 otherwise.
__device__ __forceinline__ float h_act_scale(const float* xb0, const float* xb1) {
  float M = fmaxf(xb0 ? *xb0 : 0.f, xb1 ? *xb1 : 0.f);
  if (!(M * SC_H_SX > 32752.f)) return SC_H_SX;
  M = fminf(M, 3.0e38f);
  int e;
  (void)frexpf(32752.f / M, &e);                        // 32752 / M = m * 2^e, m in [0.5, 1)  ->  2^(e-1) <= 32752 / M
  e = e - 1 < -120 ? -120 : e - 1;
  return ldexpf(1.f, e);
}
// The operand scale s and (forward sources) the clamp are folded into the prologue constants by the callers: s is a power of
// two, so  clamp(s * min(max(x*sc + sh, lo), hi))  ==  med3(x*(s*sc) + s*sh, max(s*lo, -65504), min(s*hi, 65504))  bit for bit
// (h_lo / h_hi / sc_pro_affine_h), and  s * (A g + B y + D)  ==  (sA) g + (sB) y + sD  -- 3 resp. 1 VALU less per staged value.
__device__ __forceinline__ float h_lo(float lo, float s) { return fmaxf(lo * s, -SC_H_MAX); }
__device__ __forceinline__ float h_hi(float hi, float s) { return fminf(hi * s, SC_H_MAX); }
__device__ __forceinline__ float sc_pro_affine_h(float x, float sc, float sh, float lo, float hi) {
  return __builtin_amdgcn_fmed3f(fmaf(x, sc, sh), lo, hi);
}
// The remainder a - h0 is ONE v_fma_mix_f32 per value (f16 half of the packed first term x -1 + a, the same single rounding as
// convert-back-and-subtract): a pair costs cvt_pk, 2 x fma_mix, cvt_pk instead of cvt_pk, 2 x cvt, (pk_)sub, cvt_pk.  The
// compiler does not form it by itself (it rewrites fma(h, -1, a) into the subtraction), hence the inline assembly.
#ifndef SC_SPLIT_MIX
#define SC_SPLIT_MIX 1
#endif
template <bool CLAMP = true>
__device__ __forceinline__ void split2h(float a, float b, unsigned& t0, unsigned& t1) {
  floatx2 v = {a, b};
  if constexpr (CLAMP) v = floatx2{__builtin_amdgcn_fmed3f(a, -SC_H_MAX, SC_H_MAX), __builtin_amdgcn_fmed3f(b, -SC_H_MAX, SC_H_MAX)};
  const halfx2 h0 = __builtin_convertvector(v, halfx2);
  t0 = __builtin_bit_cast(unsigned, h0);
#if SC_SPLIT_MIX
  float ra, rb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(t0), "v"(v[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(t0), "v"(v[1]));
  v = floatx2{ra, rb};
#else
  v -= __builtin_convertvector(h0, floatx2);
#endif
  const halfx2 h1 = __builtin_convertvector(v, halfx2);
  t1 = __builtin_bit_cast(unsigned, h1);
}
// The split for the producer waves of k_conv3_ws.  Without the fma_mix form it keeps its remainders scalar: beside the MFMAs
// of the consumer wave on the same SIMD a v_pk_add_f32 costs several issue slots (MI355X_MICROARCH.md: packed f32 VALU is "an
// anti-lever beside MFMAs"; measured -7 % there), while in the single-role kernels, whose staging phases are VALU-bound, the
// packed subtraction was the faster one (weight gradients +3-6 %, thin forward +2-5 % with the scalar form).
template <bool CLAMP = true>
__device__ __forceinline__ void split2h_scalar(float a, float b, unsigned& t0, unsigned& t1) {
#if SC_SPLIT_MIX
  split2h<CLAMP>(a, b, t0, t1);
#else
  if constexpr (CLAMP) {
    a = __builtin_amdgcn_fmed3f(a, -SC_H_MAX, SC_H_MAX);
    b = __builtin_amdgcn_fmed3f(b, -SC_H_MAX, SC_H_MAX);
  }
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;
  const float ra = a - (float)ha, rb = b - (float)hb;
  const _Float16 la = (_Float16)ra, lb = (_Float16)rb;
  t0 = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
  t1 = (unsigned)__builtin_bit_cast(unsigned short, la) | ((unsigned)__builtin_bit_cast(unsigned short, lb) << 16);
#endif
}
template <bool HF>
__device__ __forceinline__ floatx16 mfma_split(const bf16x8& a, const bf16x8& b, const floatx16& c) {
  if constexpr (HF) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(halfx8, a), __builtin_bit_cast(halfx8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

struct ConvXP {
  SrcD s0, s1;
  const uintx4* wpk;
  int N, H, W, Cout;
  float* out0; float* out1;
  int csplit, accum0, accum1;
  const float* add0; const float* add1;
  float* stats;
  int down0;
  const float* absmax;
  const float* xb0; const float* xb1;      // activation bounds of the sources (h_act_scale)
  int xcdmap;          // 1: 1-D grid, the cout tiles of a pixel tile numbered 8 apart (same XCD, back to back: its L2 serves the patch re-reads)
  // sc_bnr_args: BatchNorm-backward sums of out0's tensor from the data-gradient epilogue (bnr_y = NULL: off)
  const float* bnr_y; const float* bnr_cst; float* bnr_rows; float* bnr_absmax; int bnr_act;
};

static inline void set_bnr(ConvXP& p, const sc_bnr_args* b) {
  p.bnr_y = b ? b->y : nullptr; p.bnr_cst = b ? b->cst : nullptr; p.bnr_rows = b ? b->rows : nullptr;
  p.bnr_absmax = b ? b->absmax : nullptr; p.bnr_act = b ? b->act : SC_ACT_NONE;
}

__device__ __forceinline__ void wave_absmax_to(float mx, float* slot) {      // one atomic per wave, only when it raises the slot
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > __builtin_nontemporal_load(slot)) atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, mx));
}

// Data-gradient epilogue of k_conv3_bx3 for a cout tile of out0 WITH the BatchNorm-backward sums of out0's tensor (sc_bnr_args): the
// launch writes that tensor's complete gradient g (host: accum0 = 0, no add tensors), so what sc_bn_bwd_reduce(g, y) would compute
// in a pass of its own over both tensors -- {sum g', sum g' x_hat}, g' = g act'(BN(y)), and the range hint max |scale g'| -- costs one
// more read of y here.  Partial rows in the SC_STAT_CONV3 layout (two per work-group), summed in fp64 by sc_bn_bwd_finalize_rows32.
// The raw values y of a (q) block are requested together (16 or 32 loads in flight per lane) before the stores of the block.
template <int Q, bool HF>
__device__ __forceinline__ void bx3_epilogue_bnr(const ConvXP& p, const floatx16 (&acc)[2][Q], float hinv, int n, int cot, int ty, int tx, int y0,
                                                 int x0, int tiles_x, int wave, int l31, int lhi, int tid, float (&s_red)[4][32 * Q][2]) {
  constexpr int CO_T = 32 * Q;
  const int H = p.H, W = p.W, Cs = p.csplit, c0 = cot * CO_T;
  const int ox = x0 + l31;
  const float blo = sc_act_lo(p.bnr_act), bhi = sc_act_hi(p.bnr_act);
  const float* const bc = p.bnr_cst + (size_t)c0 * SC_CST;
  float mx = 0.f;
  if (p.down0) {
    const unsigned hq32 = (unsigned)((H >> 1) * (W >> 1));
    const size_t cbase = ((size_t)n * Cs + c0) * hq32;
    float* const ob = p.out0 + cbase;
    const float* const yb = p.bnr_y + cbase;
    const int oy = y0 + 2 * wave;
    const bool st = !(l31 & 1) && oy < H && ox < W;
    const unsigned loff = (unsigned)(4 * lhi) * hq32 + (unsigned)(st ? (oy >> 1) * (W >> 1) + (ox >> 1) : 0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float yv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = q * 32 + (r & 3) + 8 * (r >> 2);
        yv[r] = (st && c0 + cu + 4 * lhi < Cs) ? yb[loff + (unsigned)cu * hq32] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = q * 32 + (r & 3) + 8 * (r >> 2);
        const int col = cu + 4 * lhi;
        const bool okc = c0 + col < Cs;
        float v = ((oy < H) && (ox < W)) ? acc[0][q][r] : 0.f;
        v += ((oy + 1 < H) && (ox < W)) ? acc[1][q][r] : 0.f;
        if (HF) v *= hinv;
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
        const bool ok = st && okc;
        if (ok) ob[loff + (unsigned)cu * hq32] = v;
        const float4 cc = *reinterpret_cast<const float4*>(bc + (size_t)(okc ? col : 0) * SC_CST);
        const float yh = fmaf(yv[r], cc.x, cc.y);
        const float gq = (ok && yh > blo && yh < bhi) ? v : 0.f;
        mx = fmaxf(mx, fabsf(gq * cc.x));
        const float s = half_sum32(gq);
        const float ss = half_sum32(gq * ((yv[r] - cc.z) * cc.w));
        if (l31 == SC_HALF_SUM_LANE) { s_red[wave][col][0] = s; s_red[wave][col][1] = ss; }
      }
    }
  } else {
    const size_t HWs = (size_t)H * W;
    const size_t cbase = ((size_t)n * Cs + c0) * HWs;
    float* const ob = p.out0 + cbase;
    const float* const yb = p.bnr_y + cbase;
    const unsigned hw32 = (unsigned)HWs;
    unsigned loff[2]; bool okp[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int oy = y0 + 2 * wave + pp;
      okp[pp] = (oy < H) && (ox < W);
      loff[pp] = (unsigned)(4 * lhi) * hw32 + (unsigned)(okp[pp] ? oy * W + ox : 0);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float yv[2][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = q * 32 + (r & 3) + 8 * (r >> 2);
        const bool okc = c0 + cu + 4 * lhi < Cs;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) yv[pp][r] = (okp[pp] && okc) ? yb[loff[pp] + (unsigned)cu * hw32] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = q * 32 + (r & 3) + 8 * (r >> 2);
        const int col = cu + 4 * lhi;
        const bool okc = c0 + col < Cs;
        const float4 cc = *reinterpret_cast<const float4*>(bc + (size_t)(okc ? col : 0) * SC_CST);
        float sv = 0.f, sq = 0.f;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const bool ok = okp[pp] && okc;
          float v = ok ? acc[pp][q][r] : 0.f;
          if (HF) v *= hinv;
          if (ok) ob[loff[pp] + (unsigned)cu * hw32] = v;
          const float yh = fmaf(yv[pp][r], cc.x, cc.y);
          const float gq = (ok && yh > blo && yh < bhi) ? v : 0.f;
          sv += gq;
          sq = fmaf(gq, (yv[pp][r] - cc.z) * cc.w, sq);
          mx = fmaxf(mx, fabsf(gq * cc.x));
        }
        const float s = half_sum32(sv);
        const float ss = half_sum32(sq);
        if (l31 == SC_HALF_SUM_LANE) { s_red[wave][col][0] = s; s_red[wave][col][1] = ss; }
      }
    }
  }
  if (p.bnr_absmax) wave_absmax_to(mx, p.bnr_absmax);
  __syncthreads();
  const int rows4 = (H + 3) >> 2;
  for (int i = tid; i < 2 * CO_T * 2; i += 256) {
    const int hh = i / (CO_T * 2), rem = i - hh * (CO_T * 2);
    const int col = rem >> 1, k = rem & 1;
    const int co = c0 + col;
    const int t4 = 2 * ty + hh;
    if (co < Cs && t4 < rows4) {
      const size_t row = ((size_t)n * rows4 + t4) * tiles_x + tx;
      p.bnr_rows[(row * Cs + co) * 2 + k] = s_red[2 * hh][col][k] + s_red[2 * hh + 1][col][k];
    }
  }
}

// exact three-term bf16 split of two floats; returns packed pairs (low half = first value)
__device__ __forceinline__ void split3x2(float a, float b, unsigned& t0, unsigned& t1, unsigned& t2) {
  floatx2 v = {a, b};
  const bf16x2 h0 = __builtin_convertvector(v, bf16x2);
  v -= __builtin_convertvector(h0, floatx2);
  const bf16x2 h1 = __builtin_convertvector(v, bf16x2);
  v -= __builtin_convertvector(h1, floatx2);
  const bf16x2 h2 = __builtin_convertvector(v, bf16x2);
  t0 = __builtin_bit_cast(unsigned, h0);
  t1 = __builtin_bit_cast(unsigned, h1);
  t2 = __builtin_bit_cast(unsigned, h2);
}

// BNB: the (single) source is a BatchNorm/activation-backward source (dgrad); otherwise affine/raw sources (forward)
// NT : number of bf16 terms per operand: 3 = fp32-accurate (six products), 2 = a0*b1 + a1*b0 + a0*b0 (three products, operand
//      error 2^-18: the opt-in "fp32-bwd2" / "fp32-2" modes), 1 = plain bf16 operands (one product; the "bf16" precision
//      mode of the network: bf16 matrix math, fp32 accumulation and fp32 tensors in HBM)
template <int Q, bool BNB, int NT, bool HF = false>
__global__ __launch_bounds__(256, Q == 2 ? 2 : 3) void k_conv3_bx3(const ConvXP p) {
  static_assert(!HF || NT == 2, "the fp16 mode has two terms");
  constexpr int PR = 10, PC = 34, NPX = PR * PC;     // 8 output rows + halo
  constexpr bool PAD = (Q == 2);                     // Q = 1 keeps LDS under 53 KB (3 work-groups per CU) with guarded stores
  constexpr int NPXP = PAD ? 384 : NPX;              // padded: 3 staging rounds x 128 threads store unconditionally
  constexpr int CO_T = 32 * Q;
  constexpr int WENT = 6 * NT * CO_T;                // 16-byte filter entries per (chunk, kh) stage
  constexpr int NWV = (WENT + 255) / 256;
  constexpr int WENTP = PAD ? NWV * 256 : WENT;      // padded likewise
  constexpr int NR = 3;
  // filters fetched two stages ahead: 12 more registers, so not in the 256-register three-term kernels and not in the
  // 32-cout BatchNorm-backward kernel (168 registers for three work-groups per CU: it would spill)
  constexpr bool W2 = (NT != 3) && !(Q == 1 && BNB);

  __shared__ uintx4 s_p[NT][2][NPXP];
  __shared__ uintx4 s_w[2][WENTP];
  __shared__ float s_red[4][CO_T][2];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  int n, cot, tile;
  if (p.xcdmap) {
    // work-groups go to the 8 XCDs round-robin by linear id and each XCD has its own L2: number the cout tiles of one pixel tile 8
    // apart so that they run on ONE XCD back to back and the patch they all stage is fetched from memory once, not once per cout tile
    const int ncot = (p.Cout + CO_T - 1) / CO_T;
    const int per_img = tiles_x * ((H + 7) >> 3);
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    int pt;
    if (p.xcdmap == 3) {
      // cout-major: every XCD owns a few cout tiles (or, with fewer than 8 tiles, 8/ncot XCDs share one and split the pixel
      // tiles) and walks all pixel tiles for them.  For layers whose packed filters exceed an XCD's 4 MB L2 (decoder.blocks.0:
      // 12.7 MB) the pixel-major orders re-stream ALL filters from memory for every group of pixel tiles (0.8 GB per launch,
      // the work-groups then run at the ~9 B/clk/CU of memory-side loads); this way an XCD's filters stay in its L2 and the
      // (much smaller) patches are what is fetched 8 times
      const int total = per_img * p.N;
      if (ncot >= 8) {
        const int mine = (ncot - xcd + 7) >> 3;             // cout tiles xcd, xcd + 8, ...
        const int k = slot / total;
        if (k >= mine) return;
        cot = xcd + 8 * k; pt = slot - k * total;
      } else {
        const int g = 8 / ncot;                             // ncot in {1, 2, 4} (host-checked)
        const int per = (total + g - 1) / g;
        cot = xcd / g; pt = (xcd - cot * g) * per + slot;
        if (slot >= per || pt >= total) return;
      }
      n = pt / per_img; tile = pt - n * per_img;
    } else {
    if (p.xcdmap == 2) {
      // each XCD walks a CONTIGUOUS eighth of the pixel tiles in order: x- and y-neighbouring tiles (which share halo columns /
      // rows and the 128-byte lines their misaligned 34-pixel row segments straddle) run on one XCD within its ~64 resident
      // work-groups, so those lines are L2 hits instead of separate memory requests
      const int total = per_img * p.N, per_xcd = (total + 7) >> 3;
      const int j = slot / ncot;
      pt = xcd * per_xcd + j;
      if (j >= per_xcd || pt >= total) return;
    } else {
      pt = (slot / ncot) * 8 + xcd;
      if (pt >= per_img * p.N) return;
    }
    cot = slot % ncot;
    n = pt / per_img; tile = pt - n * per_img;
    }
  } else {
    n = blockIdx.z; cot = blockIdx.y; tile = blockIdx.x;
  }
  // (integer division has no scalar form: its uniform results come back in VGPRs and drag every address derived from them into
  // 64-bit VALU arithmetic; readfirstlane says that they are uniform)
  n = __builtin_amdgcn_readfirstlane(n); cot = __builtin_amdgcn_readfirstlane(cot); tile = __builtin_amdgcn_readfirstlane(tile);
  const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
  const int y0 = ty * 8, x0 = tx * 32;
  const int C0 = p.s0.C;
  const int Cin = C0 + p.s1.C;
  const int nk = (Cin + 15) >> 4;                    // packed filters are zero-padded to nk*16 input channels
  const uintx4* wbase = p.wpk + (size_t)cot * nk * 3 * WENT;

  // fp16 mode: operand scale of the staged tensor and the factor that removes it (and the filters' 2^8) again
  const float hsx = !HF ? 1.f : (BNB ? h_grad_scale(p.absmax) : h_act_scale(p.xb0, p.xb1));
  const float hinv = !HF ? 1.f : 1.f / (hsx * SC_H_SW);

  floatx16 acc[2][Q];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pp][q][r] = 0.f;

  // ---- staging state ----
  const int hw = __builtin_amdgcn_readfirstlane(wave >> 1);     // channel half staged by this wave (uniform)
  const int sidx = tid & 127;
  unsigned off0[NR], off1[NR];        // clamped pixel offsets in source 0 / source 1 (they may differ in `up`)
  unsigned inb = 0;
  {
    const int up0 = p.s0.up, up1 = p.s1.up;
    const int Ws0 = W >> up0, Ws1 = W >> up1;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = sidx + 128 * r;
      const int pr = e / PC, pc = e - pr * PC;
      const int y = y0 - 1 + pr, x = x0 - 1 + pc;
      const bool ok = (e < NPX) && (y >= 0) && (y < H) && (x >= 0) && (x < W);
      off0[r] = ok ? (unsigned)((y >> up0) * Ws0 + (x >> up0)) : 0u;
      off1[r] = ok ? (unsigned)((y >> up1) * Ws1 + (x >> up1)) : 0u;
      inb |= ok ? (1u << r) : 0u;
    }
  }
  float xv[NR][8], av[BNB ? NR : 1][8];
  uintx4 wvA[NWV], wvB[W2 ? NWV : 1];    // filter stages s+1 and (W2) s+2, see the pipeline below
  // per-chunk source description (uniform)
  const float* xb = nullptr; const float* ab = nullptr; const float* cb = nullptr;
  size_t plane = 0; int nch = 0; bool second = false;
  float slo = 0.f, shi = 0.f;

  auto select_chunk = [&](int kc) {
    second = kc * 16 >= C0;
    const SrcD& s = second ? p.s1 : p.s0;
    slo = sc_act_lo(s.act); shi = sc_act_hi(s.act);
    if constexpr (HF && !BNB) { slo = h_lo(slo, hsx); shi = h_hi(shi, hsx); }      // (operand scale + clamp folded, see split2h)
    plane = (size_t)(H >> s.up) * (W >> s.up);
    const int cbase = kc * 16 + hw * 8 - (second ? C0 : 0);       // first channel (source space) staged by this wave
    nch = s.C - cbase;                                            // channels j < nch exist
    const size_t o = ((size_t)n * s.C + (nch > 0 ? cbase : 0)) * plane;
    xb = s.x + o;
    ab = BNB ? s.aux + o : nullptr;
    cb = (s.mode != SC_SRC_RAW) ? s.cst + (size_t)(nch > 0 ? cbase : 0) * SC_CST : nullptr;
  };
  auto load_round = [&](int r) {
    const unsigned o = second ? off1[r] : off0[r];
    // channels beyond the source's last one re-read its LAST channel (masked to zero at conversion): clamping to channel 0
    // instead makes the address equal to j = 0's, which the compiler turns into "load channel 0, wait for it, copy, branch
    // around the other loads" -- a memory round trip at issue time in every staging round
    const int jmax = nch > 0 ? nch - 1 : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t cj = (size_t)(j < jmax ? j : jmax) * plane;
      xv[r][j] = xb[cj + o];
      if (BNB) av[r][j] = ab[cj + o];
    }
  };
  // prologue + three-term split of two channels (2jp, 2jp+1) of staging round r -> one dword of each term vector.
  // The twelve units of a chunk ride along with the MFMA steps of filter rows 1 and 2 (interleaved with them by
  // sched_group_barrier), so that only the LDS writes remain between the two barriers at the end of a chunk.
  uintx4 pt[NR][NT];
  float cs0[8], cs1[8], cs2[8], cs3[8], cs4[8];         // per-channel constants of the chunk being staged
  auto load_consts = [&]() {
    const int jmax = nch > 0 ? nch - 1 : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cj = j < jmax ? j : jmax;
      cs0[j] = BNB ? 1.f : hsx; cs1[j] = 0.f; cs2[j] = 0.f; cs3[j] = 0.f; cs4[j] = 0.f;
      if (cb) {
        const float4 c = *reinterpret_cast<const float4*>(cb + (size_t)cj * SC_CST);
        if (BNB) { cs0[j] = c.x; cs1[j] = c.y; cs2[j] = c.z; cs3[j] = c.w; cs4[j] = cb[(size_t)cj * SC_CST + 4]; }
        else { cs0[j] = c.x * hsx; cs1[j] = c.y * hsx; }
      }
    }
  };
  auto convert_unit = [&](int r, int jp) {
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 2 * jp + h;
      const float t = BNB ? sc_pro_bnbwd(xv[r][j], av[BNB ? r : 0][j], cs0[j], cs1[j], cs2[j], cs3[j], cs4[j], slo, shi)
                          : (HF ? sc_pro_affine_h(xv[r][j], cs0[j], cs1[j], slo, shi) : sc_pro_affine(xv[r][j], cs0[j], cs1[j], slo, shi));
      v[h] = (((inb >> r) & 1u) && j < nch) ? t : 0.f;
    }
    unsigned t[3];
    if constexpr (HF) {      // (BNB: the constants are wave-uniform SGPR values here, scaling them would move them to VGPRs)
      if constexpr (BNB) split2h(v[0] * hsx, v[1] * hsx, t[0], t[1]); else split2h<false>(v[0], v[1], t[0], t[1]);
      t[2] = 0u;
    }
    else split3x2(v[0], v[1], t[0], t[1], t[2]);
#pragma unroll
    for (int c = 0; c < NT; ++c) pt[r][c][jp] = t[c];
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = sidx + 128 * r;
      if (PAD || e < NPX) {
#pragma unroll
        for (int c = 0; c < NT; ++c) s_p[c][hw][e] = pt[r][c];
      }
    }
  };
  auto load_w = [&](uintx4 (&wv)[NWV], int s) {
    const uintx4* src = wbase + (size_t)s * WENT;
#pragma unroll
    for (int j = 0; j < NWV; ++j) {
      const int i = tid + 256 * j;
      wv[j] = src[i < WENT ? i : WENT - 1];
    }
  };
  auto store_w = [&](const uintx4 (&wv)[NWV], int buf) {
#pragma unroll
    for (int j = 0; j < NWV; ++j)
      if (PAD || tid + 256 * j < WENT) s_w[buf][tid + 256 * j] = wv[j];
  };
  // operand fetches of one step are issued before the 6*Q MFMAs of the previous step (explicit software pipeline: the
  // scheduler barrier keeps the LDS reads ~6*Q*32 cycles ahead of their use)
  auto load_A = [&](bf16x8 (&A)[Q][NT], int buf, int kw) {
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int c = 0; c < NT; ++c)
        A[q][c] = __builtin_bit_cast(bf16x8, s_w[buf][((c * 3 + kw) * 2 + lhi) * CO_T + q * 32 + l31]);
  };
  auto load_B = [&](bf16x8 (&B)[NT], int kh, int kw, int pp) {
    const int e = (2 * wave + pp + kh) * PC + l31 + kw;
#pragma unroll
    for (int c = 0; c < NT; ++c) B[c] = __builtin_bit_cast(bf16x8, s_p[c][lhi][e]);
  };
  // six partial products, cout blocks interleaved (independent accumulators back to back)
#define SC_BX3_STEP(A, B, PP, TA, TB)                                                                              \
  _Pragma("unroll") for (int q = 0; q < Q; ++q)                                                                    \
      acc[PP][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q][TA], B[TB], acc[PP][q], 0, 0, 0);
#define SC_BX3_MFMAS(A, B, PP)                                                                                     \
  SC_BX3_STEP(A, B, PP, 1, 1) SC_BX3_STEP(A, B, PP, 2, 0) SC_BX3_STEP(A, B, PP, 0, 2) SC_BX3_STEP(A, B, PP, 1, 0)  \
  SC_BX3_STEP(A, B, PP, 0, 1) SC_BX3_STEP(A, B, PP, 0, 0)
  // One step = the 6*Q MFMAs of (A, B) into acc[PP][*] with, when CV, one conversion unit (r, jp) cut into five slices
  // that are pinned between the MFMAs by scheduler barriers: each slice (<= 7 VALU) issues in the shadow of one MFMA.
  auto step = [&](const bf16x8 (&A)[Q][NT], const bf16x8 (&B)[NT], auto ppc, auto cvt, int unit) {
    constexpr int PP = decltype(ppc)::value;
    constexpr bool CV = decltype(cvt)::value && NT == 3;     // (measured: no gain from interleaving the two-term conversions)
    constexpr int TA[6] = {NT == 3 ? 1 : 0, NT == 3 ? 2 : 1, 0, 1, 0, 0}, TB[6] = {NT == 3 ? 1 : (NT == 2 ? 1 : 0), 0, NT == 3 ? 2 : 0, 0, 1, 0};
    constexpr int NM = (NT == 3 ? 6 : (NT == 2 ? 3 : 1)) * Q;
    constexpr int G = NM >= 6 ? NM / 6 : 1;         // MFMAs between slices
    const int r = unit >> 2, jp = unit & 3;
    float v0 = 0.f, v1 = 0.f;
    floatx2 vv = {0.f, 0.f};
    bf16x2 h0 = {}, h1 = {}, h2 = {};
    auto mf = [&](int i) {
      const int pr = i / Q, q = i - pr * Q;
      acc[PP][q] = mfma_split<HF>(A[q][TA[pr]], B[TB[pr]], acc[PP][q]);
    };
    auto pro = [&](int j) {
      const float t = BNB ? sc_pro_bnbwd(xv[r][j], av[BNB ? r : 0][j], cs0[j], cs1[j], cs2[j], cs3[j], cs4[j], slo, shi)
                          : sc_pro_affine(xv[r][j], cs0[j], cs1[j], slo, shi);
      return (((inb >> r) & 1u) && j < nch) ? t : 0.f;
    };
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      mf(i);
      if (CV && (i + 1) % G == 0 && i + 1 < NM) {
        const int sl = (i + 1) / G - 1;             // 0..4
        __builtin_amdgcn_sched_barrier(0);
        if (sl == 0) v0 = pro(2 * jp);
        if (sl == 1) v1 = pro(2 * jp + 1);
        {
          if (sl == 2) { vv = (floatx2){v0, v1}; h0 = __builtin_convertvector(vv, bf16x2); vv -= __builtin_convertvector(h0, floatx2); }
          if (sl == 3) { h1 = __builtin_convertvector(vv, bf16x2); vv -= __builtin_convertvector(h1, floatx2); }
          if (sl == 4) {
            h2 = __builtin_convertvector(vv, bf16x2);
            pt[r][0][jp] = __builtin_bit_cast(unsigned, h0);
            pt[r][NT > 1 ? 1 : 0][jp] = __builtin_bit_cast(unsigned, NT > 1 ? h1 : h0);
            pt[r][NT > 2 ? 2 : 0][jp] = __builtin_bit_cast(unsigned, NT > 2 ? h2 : h0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // conversion units 6*cv .. 6*cv+5 ride along with the six steps of a filter row
  auto compute = [&](int kh, int buf, auto cvt, int cv) {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    bf16x8 A0[Q][NT], A1[Q][NT], B0[NT], B1[NT];
    load_A(A0, buf, 0); load_B(B0, kh, 0, 0);
    load_B(B1, kh, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, B0, P0{}, cvt, 6 * cv + 0);
    __builtin_amdgcn_sched_barrier(0);
    load_A(A1, buf, 1); load_B(B0, kh, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, B1, P1{}, cvt, 6 * cv + 1);
    __builtin_amdgcn_sched_barrier(0);
    load_B(B1, kh, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    step(A1, B0, P0{}, cvt, 6 * cv + 2);
    __builtin_amdgcn_sched_barrier(0);
    load_A(A0, buf, 2); load_B(B0, kh, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    step(A1, B1, P1{}, cvt, 6 * cv + 3);
    __builtin_amdgcn_sched_barrier(0);
    load_B(B1, kh, 2, 1);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, B0, P0{}, cvt, 6 * cv + 4);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, B1, P1{}, cvt, 6 * cv + 5);
  };
#undef SC_BX3_MFMAS
#undef SC_BX3_STEP
  // the same six steps with every operand requested TWO steps before its use (three patch-operand buffers, 8 more registers):
  // one step of 3*Q MFMAs (96 / 192 cycles) does not cover the LDS latency once 8-12 waves per CU are reading
  auto compute_deep = [&](int kh, int buf) {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    constexpr std::false_type nocv{};
    bf16x8 A0[Q][NT], A1[Q][NT], Ba[NT], Bb[NT], Bc[NT];
    load_A(A0, buf, 0); load_B(Ba, kh, 0, 0); load_B(Bb, kh, 0, 1);
    load_A(A1, buf, 1); load_B(Bc, kh, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, Ba, P0{}, nocv, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_B(Ba, kh, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, Bb, P1{}, nocv, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_A(A0, buf, 2); load_B(Bb, kh, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    step(A1, Bc, P0{}, nocv, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_B(Bc, kh, 2, 1);
    __builtin_amdgcn_sched_barrier(0);
    step(A1, Ba, P1{}, nocv, 0);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, Bb, P0{}, nocv, 0);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, Bc, P1{}, nocv, 0);
  };

  // ---- pipeline: filters double-buffered per filter row, patch single-buffered with register prefetch ----
  // vmcnt retires in order: waiting for a filter stage that was requested AFTER the next chunk's patch waits for the patch as
  // well.  W2: the filters of stage s+2 are requested at the top of stage s, BEFORE the patch loads of that stage, so the stage
  // ends of a chunk only ever wait for requests older than the patch and the patch has the whole chunk (three stages of MFMAs)
  // to arrive.  Without W2 (three-term kernels, no registers left) it has one stage.
  select_chunk(0);
#pragma unroll
  for (int r = 0; r < NR; ++r) load_round(r);
  load_w(wvA, 0);
  load_consts();
#pragma unroll
  for (int u = 0; u < 12; ++u) convert_unit(u >> 2, u & 3);
  store_patch();
  store_w(wvA, 0);
  const int nst = 3 * nk;
  if constexpr (W2) load_w(wvA, 1);             // nst >= 3
  __syncthreads();
  auto next_patch = [&](int kc) {
    select_chunk(kc + 1);
#pragma unroll
    for (int r = 0; r < NR; ++r) load_round(r);        // the whole next patch: a chunk of MFMAs hides the HBM latency
    load_consts();
  };
  if constexpr (W2) {
    // stage s: request W(s+2) into WB, [patch], MFMAs from buffer s&1, W(s+1) (in WA since the previous stage) -> buffer (s+1)&1
    auto stage = [&](int s, int kh, uintx4 (&WA)[NWV], uintx4 (&WB)[NWV], int patch_kc) {
      load_w(WB, s + 2 < nst ? s + 2 : nst - 1);
      if (patch_kc >= 0) next_patch(patch_kc);
      compute_deep(kh, s & 1);
      store_w(WA, (s + 1) & 1);
      __syncthreads();
    };
    auto chunk = [&](int kc, uintx4 (&WA)[NWV], uintx4 (&WB)[NWV]) {
      const int s = 3 * kc;
      const bool more = kc + 1 < nk;
      stage(s, 0, WA, WB, more ? kc : -1);
      stage(s + 1, 1, WB, WA, -1);
      stage(s + 2, 2, WA, WB, -1);
      if (more) {
#pragma unroll
        for (int u = 0; u < 12; ++u) convert_unit(u >> 2, u & 3);
        store_patch();
        __syncthreads();
      }
    };
    for (int kc = 0; kc < nk; kc += 2) {
      chunk(kc, wvA, wvB);
      if (kc + 1 < nk) chunk(kc + 1, wvB, wvA);
    }
  } else {
    auto stage = [&](int s, int kh, auto cvt, int cv) {
      load_w(wvA, s + 1 < nst ? s + 1 : s);          // the last stage reloads itself (unused)
      if constexpr (NT != 3) compute_deep(kh, s & 1); else compute(kh, s & 1, cvt, cv);
      store_w(wvA, (s + 1) & 1);
      __syncthreads();
    };
    for (int kc = 0; kc < nk; ++kc) {
      const int s = 3 * kc;
      if (kc + 1 < nk) {
        next_patch(kc);
        stage(s, 0, std::false_type{}, 0);
        if (BNB) {   // the longer BatchNorm-backward prologue does not fit the MFMA shadows (and the registers): convert after
          stage(s + 1, 1, std::false_type{}, 0);
          stage(s + 2, 2, std::false_type{}, 0);
#pragma unroll
          for (int u = 0; u < 12; ++u) convert_unit(u >> 2, u & 3);
        } else {
          stage(s + 1, 1, std::true_type{}, 0);
          stage(s + 2, 2, std::true_type{}, 1);
        }
        store_patch();
        __syncthreads();
      } else {
        stage(s, 0, std::false_type{}, 0);
        stage(s + 1, 1, std::false_type{}, 0);
        stage(s + 2, 2, std::false_type{}, 0);
      }
    }
  }

  // ---- epilogue ----
  const size_t HWs = (size_t)H * W;
  const bool want_stats = p.stats != nullptr;
  const int ox = x0 + l31;
  if (p.csplit == p.Cout || p.csplit % CO_T == 0) {
    // The cout tile lies entirely in one output (always, in this network: a split falls on a tile boundary): one uniform base
    // pointer per work-group and a 32-bit lane offset (< 64 channels x H x W), so a store is one add + one saddr global_store
    // instead of the 64-bit multiply chain per element of the general path below -- which costs as much as two K chunks on the
    // 32- and 64-channel layers.
    const bool first = cot * CO_T < p.csplit;
    const int Cs = first ? p.csplit : p.Cout - p.csplit;                 // channels of the output this tile goes to
    const int c0 = first ? cot * CO_T : cot * CO_T - p.csplit;            // first channel of the tile in it
    const bool accum = first ? p.accum0 != 0 : p.accum1 != 0;
    const int Climit = first ? p.csplit : p.Cout;
    if constexpr (BNB) {
      if (first && p.bnr_y != nullptr) {       // + the BatchNorm-backward sums of out0's tensor (uniform branch)
        bx3_epilogue_bnr<Q, HF>(p, acc, hinv, n, cot, ty, tx, y0, x0, tiles_x, wave, l31, lhi, tid, s_red);
        return;
      }
    }
    if (first && p.down0) {
      // backward of nearest x2 upsampling fused into the store: the wave's two rows are a vertical pixel pair, adjacent lanes a
      // horizontal one -> sum the 2x2 block and store it at half resolution (no full-resolution temporary)
      const unsigned hq32 = (unsigned)((H >> 1) * (W >> 1));
      float* const ob = p.out0 + ((size_t)n * Cs + c0) * hq32;
      const int oy = y0 + 2 * wave;
      const bool st = !(l31 & 1) && oy < H && ox < W;
      const unsigned loff = (unsigned)(4 * lhi) * hq32 + (unsigned)(st ? (oy >> 1) * (W >> 1) + (ox >> 1) : 0);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = q * 32 + (r & 3) + 8 * (r >> 2);
          float v = ((oy < H) && (ox < W)) ? acc[0][q][r] : 0.f;
          v += ((oy + 1 < H) && (ox < W)) ? acc[1][q][r] : 0.f;
          if (HF) v *= hinv;
          v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
          if (st && cot * CO_T + cu + 4 * lhi < Climit) {
            const unsigned off = loff + (unsigned)cu * hq32;
            if (accum) v += ob[off];
            ob[off] = v;
          }
        }
      }
    } else {
      const size_t cbase = ((size_t)n * Cs + c0) * HWs;
      float* const ob = (first ? p.out0 : p.out1) + cbase;
      const float* const a0 = p.add0 ? p.add0 + cbase : nullptr;
      const float* const a1 = p.add1 ? p.add1 + cbase : nullptr;
      const unsigned hw32 = (unsigned)HWs;
      unsigned loff[2]; bool okp[2];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int oy = y0 + 2 * wave + pp;
        okp[pp] = (oy < H) && (ox < W);
        loff[pp] = (unsigned)(4 * lhi) * hw32 + (unsigned)(okp[pp] ? oy * W + ox : 0);
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = q * 32 + (r & 3) + 8 * (r >> 2);            // compile-time part of the channel
          const int col = cu + 4 * lhi;
          const bool okc = cot * CO_T + col < Climit;
          float sv = 0.f, sq = 0.f;
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            const bool ok = okp[pp] && okc;
            float v = ok ? acc[pp][q][r] : 0.f;
            if (HF) v *= hinv;
            sv += v; sq = fmaf(v, v, sq);
            if (ok) {
              const unsigned off = loff[pp] + (unsigned)cu * hw32;
              if (a0) v += a0[off];
              if (a1) v += a1[off];
              if (accum) v += ob[off];
              ob[off] = v;
            }
          }
          if (want_stats) {
            const float s = half_sum32(sv);
            const float ss = half_sum32(sq);
            if (l31 == SC_HALF_SUM_LANE) { s_red[wave][col][0] = s; s_red[wave][col][1] = ss; }
          }
        }
      }
    }
  } else
#pragma unroll
  for (int q = 0; q < Q; ++q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int co = cot * CO_T + col;
      float sv = 0.f, sq = 0.f;
      if (p.down0 && co < p.csplit) {
        // backward of nearest x2 upsampling fused into the store: the wave's two rows are a vertical pixel pair, adjacent
        // lanes a horizontal one -> sum the 2x2 block and store it at half resolution (no full-resolution temporary)
        const int oy = y0 + 2 * wave;
        float v = ((oy < H) && (ox < W)) ? acc[0][q][r] : 0.f;
        v += ((oy + 1 < H) && (ox < W)) ? acc[1][q][r] : 0.f;
        if (HF) v *= hinv;
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
        if (!(l31 & 1) && oy < H && ox < W) {
          const size_t idx = (((size_t)n * p.csplit + co) * (H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1);
          if (p.accum0) v += p.out0[idx];
          p.out0[idx] = v;
        }
        continue;
      }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int oy = y0 + 2 * wave + pp;
        const bool ok = (oy < H) && (ox < W) && (co < p.Cout);
        float v = ok ? acc[pp][q][r] : 0.f;
        if (HF) v *= hinv;
        sv += v; sq = fmaf(v, v, sq);
        if (ok) {
          const size_t opix = (size_t)oy * W + ox;
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
      if (want_stats) {
        const float s = half_sum32(sv);
        const float ss = half_sum32(sq);
        if (l31 == SC_HALF_SUM_LANE) { s_red[wave][col][0] = s; s_red[wave][col][1] = ss; }
      }
    }
  }
  if (want_stats) {
    // two partial rows per work-group, laid out exactly like the 4-row tiles of k_conv_mfma<3> (SC_STAT_CONV3)
    __syncthreads();
    const int rows4 = (H + 3) >> 2;
    for (int i = tid; i < 2 * CO_T * 2; i += 256) {
      const int hh = i / (CO_T * 2), rem = i - hh * (CO_T * 2);
      const int col = rem >> 1, k = rem & 1;
      const int co = cot * CO_T + col;
      const int t4 = 2 * ty + hh;
      if (co < p.Cout && t4 < rows4) {
        const float t = s_red[2 * hh][col][k] + s_red[2 * hh + 1][col][k];
        const size_t row = ((size_t)n * rows4 + t4) * tiles_x + tx;
        p.stats[(row * p.Cout + co) * 2 + k] = t;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Wave-specialised variant of k_conv3_bx3 for the two-fp16-term mode: 8 waves, waves 0-3 CONSUME (LDS operand reads + MFMAs,
// the tile mapping of k_conv3_bx3: wave w = output rows 2w, 2w+1, all Q cout blocks), waves 4-7 PRODUCE (global loads,
// prologue, split, LDS stores of the next chunk's patch and of the next filter row).  The patch is double-buffered, so the
// two halves only meet at ONE barrier per filter-row stage.  Why: in the single-role kernel a wave's MFMA phases and its
// staging phases alternate, the other work-group on the CU is mostly in the same phase, and the matrix pipe sits at 23-38 %
// (SQ_VALU_MFMA_BUSY_CYCLES); removing the MFMAs alone (operand reads kept) left 65-75 % of the time, i.e. the two parts ran
// back to back.  Here one consumer and one producer wave share each SIMD and overlap by construction.
// Producer pipeline per stage s = 3 kc + j (all straight-line, so the compiler's vmcnt counting stays exact):
//   request filters W(s+2)  ->  convert + store round j of patch kc+1 (requested one chunk ago)  ->  [j == 2: constants of
//   chunk kc+2]  ->  request round j of patch kc+2 into the same registers  ->  store W(s+1)  ->  barrier
// (filters are requested BEFORE the patch round of the stage: vmcnt retires in order, and this way waiting for a filter row
// never waits for a younger patch request).
template <int Q, bool BNB>
__global__ __launch_bounds__(512, 4) void k_conv3_ws(const ConvXP p) {
  constexpr int NT = 2;
  constexpr bool HF = true;
  constexpr int PR = 10, PC = 34, NPX = PR * PC;     // 8 output rows + halo
  constexpr int NPXP = NPX;                          // (guarded stores in the third staging round: two work-groups per CU need < 80 KB)
  constexpr int CO_T = 32 * Q;
  constexpr int WENT = 6 * NT * CO_T;                // 16-byte filter entries per (chunk, kh) stage
  constexpr int NWV = (WENT + 255) / 256;
  constexpr int WENTP = NWV * 256;
  constexpr int NR = 3;

  __shared__ uintx4 s_p[2][NT][2][NPXP];
  __shared__ uintx4 s_w[2][WENTP];
  __shared__ float s_red[4][CO_T][2];
  constexpr int WS_BNB_MAXC = 256;                   // BatchNorm-backward sources: the constants of all channels live in LDS (host-checked)
  __shared__ __attribute__((aligned(16))) float s_cst[BNB ? WS_BNB_MAXC * SC_CST : 4];

  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);      // 0: consumer, 1: producer
  const int tid = threadIdx.x & 255;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  int n, cot, tile;
  if (p.xcdmap) {
    // work-groups go to the 8 XCDs round-robin by linear id and each XCD has its own L2: number the cout tiles of one pixel tile 8
    // apart so that they run on ONE XCD back to back and the patch they all stage is fetched from memory once, not once per cout tile
    const int ncot = (p.Cout + CO_T - 1) / CO_T;
    const int per_img = tiles_x * ((H + 7) >> 3);
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    int pt;
    if (p.xcdmap == 3) {
      // cout-major: every XCD owns a few cout tiles (or, with fewer than 8 tiles, 8/ncot XCDs share one and split the pixel
      // tiles) and walks all pixel tiles for them.  For layers whose packed filters exceed an XCD's 4 MB L2 (decoder.blocks.0:
      // 12.7 MB) the pixel-major orders re-stream ALL filters from memory for every group of pixel tiles (0.8 GB per launch,
      // the work-groups then run at the ~9 B/clk/CU of memory-side loads); this way an XCD's filters stay in its L2 and the
      // (much smaller) patches are what is fetched 8 times
      const int total = per_img * p.N;
      if (ncot >= 8) {
        const int mine = (ncot - xcd + 7) >> 3;             // cout tiles xcd, xcd + 8, ...
        const int k = slot / total;
        if (k >= mine) return;
        cot = xcd + 8 * k; pt = slot - k * total;
      } else {
        const int g = 8 / ncot;                             // ncot in {1, 2, 4} (host-checked)
        const int per = (total + g - 1) / g;
        cot = xcd / g; pt = (xcd - cot * g) * per + slot;
        if (slot >= per || pt >= total) return;
      }
      n = pt / per_img; tile = pt - n * per_img;
    } else {
    if (p.xcdmap == 2) {
      // each XCD walks a CONTIGUOUS eighth of the pixel tiles in order: x- and y-neighbouring tiles (which share halo columns /
      // rows and the 128-byte lines their misaligned 34-pixel row segments straddle) run on one XCD within its ~64 resident
      // work-groups, so those lines are L2 hits instead of separate memory requests
      const int total = per_img * p.N, per_xcd = (total + 7) >> 3;
      const int j = slot / ncot;
      pt = xcd * per_xcd + j;
      if (j >= per_xcd || pt >= total) return;
    } else {
      pt = (slot / ncot) * 8 + xcd;
      if (pt >= per_img * p.N) return;
    }
    cot = slot % ncot;
    n = pt / per_img; tile = pt - n * per_img;
    }
  } else {
    n = blockIdx.z; cot = blockIdx.y; tile = blockIdx.x;
  }
  // (integer division has no scalar form: its uniform results come back in VGPRs and drag every address derived from them into
  // 64-bit VALU arithmetic; readfirstlane says that they are uniform)
  n = __builtin_amdgcn_readfirstlane(n); cot = __builtin_amdgcn_readfirstlane(cot); tile = __builtin_amdgcn_readfirstlane(tile);
  const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
  const int y0 = ty * 8, x0 = tx * 32;
  const int C0 = p.s0.C;
  const int Cin = C0 + p.s1.C;
  const int nk = (Cin + 15) >> 4;                    // packed filters are zero-padded to nk*16 input channels
  const uintx4* wbase = p.wpk + (size_t)cot * nk * 3 * WENT;

  // fp16 mode: operand scale of the staged tensor and the factor that removes it (and the filters' 2^8) again
  const float hsx = !HF ? 1.f : (BNB ? h_grad_scale(p.absmax) : h_act_scale(p.xb0, p.xb1));
  const float hinv = !HF ? 1.f : 1.f / (hsx * SC_H_SW);

  const int nst = 3 * nk;
  const bool want_stats = p.stats != nullptr;
  if constexpr (BNB) {
    for (int i = (int)threadIdx.x; i < p.s0.C * SC_CST; i += 512) {
      const int f = i & (SC_CST - 1);                   // A, B, D (fields 2..4) carry the operand scale, see split2h
      s_cst[i] = p.s0.cst[i] * ((HF && f >= 2 && f <= 4) ? hsx : 1.f);
    }
    __syncthreads();
  }
  if (role == 1) {
    // =========================== producers ===========================
  // ---- staging state (producers) ----
  const int hw = __builtin_amdgcn_readfirstlane(wave >> 1);     // channel half staged by this wave (uniform)
  const int sidx = tid & 127;
  unsigned off0[NR], off1[NR];        // clamped pixel offsets in source 0 / source 1 (they may differ in `up`)
  unsigned inb = 0;
  {
    const int up0 = p.s0.up, up1 = p.s1.up;
    const int Ws0 = W >> up0, Ws1 = W >> up1;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = sidx + 128 * r;
      const int pr = e / PC, pc = e - pr * PC;
      const int y = y0 - 1 + pr, x = x0 - 1 + pc;
      const bool ok = (e < NPX) && (y >= 0) && (y < H) && (x >= 0) && (x < W);
      off0[r] = ok ? (unsigned)((y >> up0) * Ws0 + (x >> up0)) : 0u;
      off1[r] = ok ? (unsigned)((y >> up1) * Ws1 + (x >> up1)) : 0u;
      inb |= ok ? (1u << r) : 0u;
    }
  }
  float xv[NR][8], av[BNB ? NR : 1][8];
  uintx4 wr[3][NWV];                                   // filter rows s+1 .. s+3, each requested one chunk before its LDS store
  struct Chunk { const float* xb; const float* ab; const float* cb; size_t plane; int nch, cbase; bool second, raw; float slo, shi; };
  auto make_chunk = [&](int kc) {
    Chunk c;
    c.second = kc * 16 >= C0;
    const SrcD& s = c.second ? p.s1 : p.s0;
    c.slo = sc_act_lo(s.act); c.shi = sc_act_hi(s.act);
    if constexpr (HF && !BNB) { c.slo = h_lo(c.slo, hsx); c.shi = h_hi(c.shi, hsx); }     // (scale + clamp folded, see split2h)
    c.plane = (size_t)(H >> s.up) * (W >> s.up);
    const int cbase = kc * 16 + hw * 8 - (c.second ? C0 : 0);       // first channel (source space) staged by this wave
    c.nch = s.C - cbase;                                              // channels j < nch exist
    c.cbase = c.nch > 0 ? cbase : 0;
    const size_t o = ((size_t)n * s.C + (c.nch > 0 ? cbase : 0)) * c.plane;
    c.xb = s.x + o;
    c.ab = BNB ? s.aux + o : nullptr;
    // RAW sources load (and ignore) a few of their own values: a load under a branch would make the compiler drain vmcnt at the
    // merge, and a pointer to a __device__ table of identity constants turns the loads into FLAT ones (vmcnt(0) again)
    c.raw = s.mode == SC_SRC_RAW;
    c.cb = c.raw ? s.x : s.cst + (size_t)(c.nch > 0 ? cbase : 0) * SC_CST;
    return c;
  };
  auto load_round = [&](const Chunk& c, int r) {
    const unsigned o = c.second ? off1[r] : off0[r];
    const int jmax = c.nch > 0 ? c.nch - 1 : 0;       // channels beyond the last re-read it (masked at conversion), see k_conv3_bx3
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t cj = (size_t)(j < jmax ? j : jmax) * c.plane;
      xv[r][j] = c.xb[cj + o];
      if (BNB) av[r][j] = c.ab[cj + o];
    }
  };
  struct Cst { float sc[BNB ? 1 : 8], sh[BNB ? 1 : 8]; };    // forward: scale / shift of a chunk's eight channels (BNB: constants in LDS)
  Cst csA, csB;
  auto load_consts = [&](const Chunk& c, Cst& k) {
    if constexpr (!BNB) {
      const int jmax = c.nch > 0 ? c.nch - 1 : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cj = j < jmax ? j : jmax;
        const float2 v = *reinterpret_cast<const float2*>(c.cb + (size_t)cj * SC_CST);     // (never a conditional load: see make_chunk)
        k.sc[j] = c.raw ? hsx : v.x * hsx; k.sh[j] = c.raw ? 0.f : v.y * hsx;     // (scale folded, see split2h)
      }
    }
  };
  // prologue + split of the eight channels of round r, stored to patch buffer pb
  auto convert_store_round = [&](const Chunk& c, const Cst& k, int r, int pb) {
    uintx4 t0, t1;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * jp + h;
        float t;
        if constexpr (BNB) {
          const int ch = c.cbase + (j < c.nch ? j : 0);             // (uniform address: an LDS broadcast)
          const float4 k4 = *reinterpret_cast<const float4*>(&s_cst[ch * SC_CST]);
          t = sc_pro_bnbwd(xv[r][j], av[r][j], k4.x, k4.y, k4.z, k4.w, s_cst[ch * SC_CST + 4], c.slo, c.shi);
        } else {
          t = HF ? sc_pro_affine_h(xv[r][j], k.sc[BNB ? 0 : j], k.sh[BNB ? 0 : j], c.slo, c.shi)
                 : sc_pro_affine(xv[r][j], k.sc[BNB ? 0 : j], k.sh[BNB ? 0 : j], c.slo, c.shi);
        }
        v[h] = (((inb >> r) & 1u) && j < c.nch) ? t : 0.f;
      }
      unsigned a, b;
      split2h_scalar<BNB>(v[0], v[1], a, b);
      t0[jp] = a; t1[jp] = b;
    }
    const int e = sidx + 128 * r;
    if (e < NPX) { s_p[pb][0][hw][e] = t0; s_p[pb][1][hw][e] = t1; }
  };
  auto load_w = [&](uintx4 (&wv)[NWV], int s) {
    const uintx4* src = wbase + (size_t)s * WENT;
#pragma unroll
    for (int j = 0; j < NWV; ++j) {
      const int i = tid + 256 * j;
      wv[j] = src[i < WENT ? i : WENT - 1];
    }
  };
  auto store_w = [&](const uintx4 (&wv)[NWV], int buf) {
#pragma unroll
    for (int j = 0; j < NWV; ++j) s_w[buf][tid + 256 * j] = wv[j];
  };
    auto clampk = [&](int kc) { return kc < nk ? kc : nk - 1; };
    auto clamps = [&](int st) { return st < nst ? st : nst - 1; };
    // Every request (patch round, filter row, constants) is consumed exactly one chunk after it was issued and in issue order,
    // so each wait is "all but the requests of the last chunk" and nothing is ever waited for early (vmcnt retires in order).
    Chunk cnext = make_chunk(clampk(1));                // the chunk whose raw values sit in xv / av
    {
      const Chunk c0 = make_chunk(0);
#pragma unroll
      for (int r = 0; r < NR; ++r) load_round(c0, r);
      load_w(wr[0], 0);
      load_consts(c0, csA);
#pragma unroll
      for (int r = 0; r < NR; ++r) convert_store_round(c0, csA, r, 0);
      store_w(wr[0], 0);
      load_consts(cnext, csB);
#pragma unroll
      for (int r = 0; r < NR; ++r) { load_round(cnext, r); load_w(wr[r], clamps(1 + r)); }
    }
    __syncthreads();
    int pb = 1;                                         // patch buffer being filled
    // chunk kc (stages s = 3 kc + j): K1 = constants of chunk kc+1 (in use), K2 receives those of chunk kc+2
    auto chunk = [&](int kc, Cst& K1, Cst& K2) {
      const Chunk cn2 = make_chunk(clampk(kc + 2));
      load_consts(cn2, K2);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int s = 3 * kc + j;
        convert_store_round(cnext, K1, j, pb);
        load_round(cn2, j);
        store_w(wr[j], (s + 1) & 1);
        load_w(wr[j], clamps(s + 4));
        __syncthreads();
      }
      cnext = cn2;
      pb ^= 1;
    };
    // (always pairs of chunks, the second one past the end is a harmless repeat of the last: a conditional call would make the
    // compiler merge two request histories at the join and fall back to vmcnt(0))
    for (int kc = 0; kc < nk; kc += 2) {
      chunk(kc, csB, csA);
      chunk(kc + 1, csA, csB);
    }
    if (want_stats) __syncthreads();                    // the consumers' statistics barrier
    return;
  }
  // =========================== consumers ===========================
  floatx16 acc[2][Q];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pp][q][r] = 0.f;

  int pbuf = 0;                                         // patch buffer the consumers read
  auto load_A = [&](bf16x8 (&A)[Q][NT], int buf, int kw) {
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int c = 0; c < NT; ++c)
        A[q][c] = __builtin_bit_cast(bf16x8, s_w[buf][((c * 3 + kw) * 2 + lhi) * CO_T + q * 32 + l31]);
  };
  auto load_B = [&](bf16x8 (&B)[NT], int kh, int kw, int pp) {   // reads patch buffer `pbuf`
    const int e = (2 * wave + pp + kh) * PC + l31 + kw;
#pragma unroll
    for (int c = 0; c < NT; ++c) B[c] = __builtin_bit_cast(bf16x8, s_p[pbuf][c][lhi][e]);
  };
  // One step = the 3*Q MFMAs of (A, B) into acc[PP][*]
  auto step = [&](const bf16x8 (&A)[Q][NT], const bf16x8 (&B)[NT], auto ppc) {
    constexpr int PP = decltype(ppc)::value;
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[PP][q] = mfma_split<true>(A[q][0], B[1], acc[PP][q]);
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[PP][q] = mfma_split<true>(A[q][1], B[0], acc[PP][q]);
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[PP][q] = mfma_split<true>(A[q][0], B[0], acc[PP][q]);
  };
  // the six steps of one filter row; operands are requested one step (3*Q MFMAs) ahead: a third patch-operand buffer
  // measured no gain in the single-role kernel and would not fit the 128 registers of two work-groups per CU
  auto compute = [&](int kh, int buf) {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    bf16x8 A0[Q][NT], A1[Q][NT], B0[NT], B1[NT];
    load_A(A0, buf, 0); load_B(B0, kh, 0, 0);
    load_B(B1, kh, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, B0, P0{});
    __builtin_amdgcn_sched_barrier(0);
    load_A(A1, buf, 1); load_B(B0, kh, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, B1, P1{});
    __builtin_amdgcn_sched_barrier(0);
    load_B(B1, kh, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    step(A1, B0, P0{});
    __builtin_amdgcn_sched_barrier(0);
    load_A(A0, buf, 2); load_B(B0, kh, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    step(A1, B1, P1{});
    __builtin_amdgcn_sched_barrier(0);
    load_B(B1, kh, 2, 1);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, B0, P0{});
    __builtin_amdgcn_sched_barrier(0);
    step(A0, B1, P1{});
  };

  __syncthreads();                                      // chunk 0 staged
  for (int kc = 0; kc < ((nk + 1) & ~1); ++kc) {        // (the producers work in pairs of chunks)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (kc < nk) compute(j, (3 * kc + j) & 1);
      __syncthreads();
    }
    pbuf ^= 1;
  }

  // ---- epilogue: the four consumer waves hold the tile ----
  const size_t HWs = (size_t)H * W;
  const int ox = x0 + l31;
  if (p.csplit == p.Cout || p.csplit % CO_T == 0) {
    // The cout tile lies entirely in one output (always, in this network: a split falls on a tile boundary): one uniform base
    // pointer per work-group and a 32-bit lane offset (< 64 channels x H x W), so a store is one add + one saddr global_store
    // instead of the 64-bit multiply chain per element of the general path below -- which costs as much as two K chunks on the
    // 32- and 64-channel layers.
    const bool first = cot * CO_T < p.csplit;
    const int Cs = first ? p.csplit : p.Cout - p.csplit;                 // channels of the output this tile goes to
    const int c0 = first ? cot * CO_T : cot * CO_T - p.csplit;            // first channel of the tile in it
    const bool accum = first ? p.accum0 != 0 : p.accum1 != 0;
    const int Climit = first ? p.csplit : p.Cout;
    if (first && p.down0) {
      // backward of nearest x2 upsampling fused into the store: the wave's two rows are a vertical pixel pair, adjacent lanes a
      // horizontal one -> sum the 2x2 block and store it at half resolution (no full-resolution temporary)
      const unsigned hq32 = (unsigned)((H >> 1) * (W >> 1));
      float* const ob = p.out0 + ((size_t)n * Cs + c0) * hq32;
      const int oy = y0 + 2 * wave;
      const bool st = !(l31 & 1) && oy < H && ox < W;
      const unsigned loff = (unsigned)(4 * lhi) * hq32 + (unsigned)(st ? (oy >> 1) * (W >> 1) + (ox >> 1) : 0);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = q * 32 + (r & 3) + 8 * (r >> 2);
          float v = ((oy < H) && (ox < W)) ? acc[0][q][r] : 0.f;
          v += ((oy + 1 < H) && (ox < W)) ? acc[1][q][r] : 0.f;
          if (HF) v *= hinv;
          v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
          if (st && cot * CO_T + cu + 4 * lhi < Climit) {
            const unsigned off = loff + (unsigned)cu * hq32;
            if (accum) v += ob[off];
            ob[off] = v;
          }
        }
      }
    } else {
      const size_t cbase = ((size_t)n * Cs + c0) * HWs;
      float* const ob = (first ? p.out0 : p.out1) + cbase;
      const float* const a0 = p.add0 ? p.add0 + cbase : nullptr;
      const float* const a1 = p.add1 ? p.add1 + cbase : nullptr;
      const unsigned hw32 = (unsigned)HWs;
      unsigned loff[2]; bool okp[2];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int oy = y0 + 2 * wave + pp;
        okp[pp] = (oy < H) && (ox < W);
        loff[pp] = (unsigned)(4 * lhi) * hw32 + (unsigned)(okp[pp] ? oy * W + ox : 0);
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = q * 32 + (r & 3) + 8 * (r >> 2);            // compile-time part of the channel
          const int col = cu + 4 * lhi;
          const bool okc = cot * CO_T + col < Climit;
          float sv = 0.f, sq = 0.f;
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            const bool ok = okp[pp] && okc;
            float v = ok ? acc[pp][q][r] : 0.f;
            if (HF) v *= hinv;
            sv += v; sq = fmaf(v, v, sq);
            if (ok) {
              const unsigned off = loff[pp] + (unsigned)cu * hw32;
              if (a0) v += a0[off];
              if (a1) v += a1[off];
              if (accum) v += ob[off];
              ob[off] = v;
            }
          }
          if (want_stats) {
            const float s = half_sum32(sv);
            const float ss = half_sum32(sq);
            if (l31 == SC_HALF_SUM_LANE) { s_red[wave][col][0] = s; s_red[wave][col][1] = ss; }
          }
        }
      }
    }
  } else
#pragma unroll
  for (int q = 0; q < Q; ++q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int co = cot * CO_T + col;
      float sv = 0.f, sq = 0.f;
      if (p.down0 && co < p.csplit) {
        // backward of nearest x2 upsampling fused into the store: the wave's two rows are a vertical pixel pair, adjacent
        // lanes a horizontal one -> sum the 2x2 block and store it at half resolution (no full-resolution temporary)
        const int oy = y0 + 2 * wave;
        float v = ((oy < H) && (ox < W)) ? acc[0][q][r] : 0.f;
        v += ((oy + 1 < H) && (ox < W)) ? acc[1][q][r] : 0.f;
        if (HF) v *= hinv;
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
        if (!(l31 & 1) && oy < H && ox < W) {
          const size_t idx = (((size_t)n * p.csplit + co) * (H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1);
          if (p.accum0) v += p.out0[idx];
          p.out0[idx] = v;
        }
        continue;
      }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int oy = y0 + 2 * wave + pp;
        const bool ok = (oy < H) && (ox < W) && (co < p.Cout);
        float v = ok ? acc[pp][q][r] : 0.f;
        if (HF) v *= hinv;
        sv += v; sq = fmaf(v, v, sq);
        if (ok) {
          const size_t opix = (size_t)oy * W + ox;
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
      if (want_stats) {
        const float s = half_sum32(sv);
        const float ss = half_sum32(sq);
        if (l31 == SC_HALF_SUM_LANE) { s_red[wave][col][0] = s; s_red[wave][col][1] = ss; }
      }
    }
  }

  if (want_stats) {
    // two partial rows per work-group, laid out exactly like the 4-row tiles of k_conv_mfma<3> (SC_STAT_CONV3)
    __syncthreads();
    const int rows4 = (H + 3) >> 2;
    for (int i = tid; i < 2 * CO_T * 2; i += 256) {
      const int hh = i / (CO_T * 2), rem = i - hh * (CO_T * 2);
      const int col = rem >> 1, k = rem & 1;
      const int co = cot * CO_T + col;
      const int t4 = 2 * ty + hh;
      if (co < p.Cout && t4 < rows4) {
        const float t = s_red[2 * hh][col][k] + s_red[2 * hh + 1][col][k];
        const size_t row = ((size_t)n * rows4 + t4) * tiles_x + tx;
        p.stats[(row * p.Cout + co) * 2 + k] = t;
      }
    }
  }
}

// filters -> [co tile][chunk of 16 ci][kh][term][kw][ci half][co][8 ci] bf16; forward or transposed+flipped (dgrad)
// the bf16 / fp16 term bits of one filter value (fp16: scaled by 2^8, see split2h)
__device__ __forceinline__ void split_filter(float v, bool half, unsigned short (&t)[3]) {
  if (half) {
    const float vs = __builtin_amdgcn_fmed3f(v * SC_H_SW, -SC_H_MAX, SC_H_MAX);
    const _Float16 h0 = (_Float16)vs;
    const _Float16 h1 = (_Float16)(vs - (float)h0);
    t[0] = __builtin_bit_cast(unsigned short, h0); t[1] = __builtin_bit_cast(unsigned short, h1); t[2] = 0;
    return;
  }
  const __bf16 t0 = (__bf16)v;
  float rr = v - (float)t0;
  const __bf16 t1 = (__bf16)rr;
  rr -= (float)t1;
  const __bf16 t2 = (__bf16)rr;
  t[0] = __builtin_bit_cast(unsigned short, t0); t[1] = __builtin_bit_cast(unsigned short, t1); t[2] = __builtin_bit_cast(unsigned short, t2);
}

// ------------------------------------------------------------------------------------------
// Thin 3x3 layers (<= 16 output channels, 16 or 32 input channels: decoder.blocks.4, 512x512) with two fp16 terms on
// v_mfma_f32_16x16x32_f16.  The whole filter bank (<= 16 x 32 x 9) lives in REGISTERS in MFMA-operand form (A: lane -> cout l&15,
// k = 8*(l>>4)..+7; 5 or 9 K steps x 2 terms x 4 VGPRs), all input channels of the 10x34 patch are staged in ONE pass (no K loop,
// one barrier), and a K step covers 32 = (taps x channels): with 16 input channels two taps per MFMA.
//   B: lane -> pixel l&15 of a 16-pixel block, k group l>>4 -> (tap, channel half): one 16-byte LDS read
//   D: lane -> pixel l&15, couts 4*(l>>4) .. +3
// Work-group = 4 waves, tile 8 rows x 32 px; wave = 2 rows = four 16-pixel blocks.  Single source (may be upsampled), single
// output, optional statistics rows (SC_STAT_CONV3 layout); anything else stays on the other kernels.
// UPS: the source is the nearest-neighbour 2x up-sampling of a half-resolution tensor (decoder.blocks.4.conv1) and the patch is staged
// AT THE SOURCE RESOLUTION -- 6 x (TW/2 + 2) entries per channel group instead of 10 x (TW + 2): every source value is requested,
// run through the prologue, split and written to LDS once instead of 3.1 times (the staging was more VALU work than the kernel's MFMAs);
// the operand reads address the source entry of their output pixel: row (r + 1) >> 1, column (c + 1) >> 1 of the patch (tile origins
// are even, so output row / column -1 and H / W fall on source -1 and H/2 / W/2: the same zero padding).
template <int CIN, bool BNB, bool UPS = false>
__global__ __launch_bounds__(256, 2) void k_conv3_thin_h(const ConvXP p) {
  // Tile 8 rows x TW px.  16 input channels: TW = 64 -- the runs a work-group reads and writes per (plane, row) are 264 / 256 B instead
  // of 136 / 128 B, worth 14-17 % at 16 x 512^2 (forward 177 -> 152 us, backward 269 -> 224 us; these layers run at 3 TB/s and neither the
  // MFMAs -- removed: same time -- nor exposed latency -- persistent work-groups with the next patch in flight: same time -- limit them:
  // what is left is how DRAM likes the access pattern).  32 channels: the 85 KB patch would leave one work-group per CU (205 -> 308 us).
  constexpr int TW = CIN == 16 ? 64 : 32, PBW = TW / 16, NPB = 2 * PBW;
  constexpr int PR = UPS ? 6 : 10, PC = UPS ? TW / 2 + 2 : TW + 2, NPX = PR * PC;
  constexpr int NH = CIN / 8;                 // 8-channel groups
  constexpr int NG = 9 * NH, NS = (NG + 3) / 4;
  // Pitch between the 8-channel groups: a multiple of 16 entries (256 B).  ds_read_b128 is serviced in four NON-contiguous 16-lane
  // groups -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- over 64 banks (MI355X_MICROARCH.md, LDS): a group mixes the lanes of TWO
  // k groups lg, which read the same 16 consecutive patch entries of two channel groups one pitch apart.  With the pitch = NPX (660 or
  // 340 = 4 mod 16 entries) four of the sixteen 16-byte slots were hit twice in every group: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  // 0.29-0.41 on these kernels for five rounds (the operand reads were laid out for 8 contiguous lanes x 32 banks).
  constexpr int NPXP = (NPX + 15) & ~15;
  __shared__ uintx4 s_p[2][NH][NPXP];
  __shared__ float s_red[4][TW / 32][16][2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int H = p.H, W = p.W;
  const int tiles_x = (W + TW - 1) / TW;
  int n, tile;
  if (p.xcdmap) {      // each XCD walks a contiguous eighth of the pixel tiles (see k_conv3_bx3): halo lines shared in its L2
    const int per_img = tiles_x * ((H + 7) >> 3), total = per_img * p.N, per_xcd = (total + 7) >> 3;
    const int j = blockIdx.x >> 3, pt = (blockIdx.x & 7) * per_xcd + j;
    if (j >= per_xcd || pt >= total) return;
    n = pt / per_img; tile = pt - n * per_img;
  } else {
    n = blockIdx.z; tile = blockIdx.x;
  }
  n = __builtin_amdgcn_readfirstlane(n); tile = __builtin_amdgcn_readfirstlane(tile);      // uniform (see k_conv3_bx3)
  const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
  const int y0 = ty * 8, x0 = tx * TW;
  const float hsx = BNB ? h_grad_scale(p.absmax) : h_act_scale(p.xb0, p.xb1);
  const float hinv = 1.f / (hsx * SC_H_SW);

  const SrcD& src = p.s0;
  uintx4 A[NS][2];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) A[s][t] = p.wpk[(s * 2 + t) * 64 + lane];

  // ---- stage the patch.  A wave stages ONE 8-channel group (4 / NH waves per group), lanes = consecutive patch pixels: the
  // group -- hence the plane pointers and the per-channel constants -- is wave-uniform, so the constants are SCALAR loads into
  // SGPRs.  (First version: entry -> (group, pixel) per lane, constants fetched per element from global memory under the
  // `mode != RAW` branch: a second, fully exposed memory round trip behind the patch loads; through LDS: 48 ds_reads per thread.)
  constexpr int WPG = 4 / NH, TPG = 64 * WPG, NR = (NPX + TPG - 1) / TPG;
  const int grp = __builtin_amdgcn_readfirstlane(wave / WPG);
  const int g8 = grp * 8;
  const int e0 = tid - grp * TPG;
  const int up = UPS ? 1 : src.up;
  const int Ws = W >> up;
  const size_t plane = (size_t)(H >> up) * Ws;
  // (operand scale -- and for forward sources the clamp -- folded into the constants, see split2h)
  const float slo = BNB ? sc_act_lo(src.act) : h_lo(sc_act_lo(src.act), hsx), shi = BNB ? sc_act_hi(src.act) : h_hi(sc_act_hi(src.act), hsx);
  float xv[NR][8], av[BNB ? NR : 1][8];
  bool okv[NR];
  const float* const xg = src.x + ((size_t)n * CIN + g8) * plane;
  const float* const ag = BNB ? src.aux + ((size_t)n * CIN + g8) * plane : nullptr;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = e0 + TPG * r;
    const int pr = e / PC, pc = e - pr * PC;
    // (UPS: y, x are SOURCE coordinates and the patch covers source rows y0/2 - 1 .. y0/2 + 4)
    const int y = UPS ? (y0 >> 1) - 1 + pr : y0 - 1 + pr, x = UPS ? (x0 >> 1) - 1 + pc : x0 - 1 + pc;
    const bool ok = UPS ? ((e < NPX) && y >= 0 && y < (H >> 1) && x >= 0 && x < Ws) : ((e < NPX) && y >= 0 && y < H && x >= 0 && x < W);
    okv[r] = ok;
    const unsigned off = ok ? (UPS ? (unsigned)(y * Ws + x) : (unsigned)((y >> up) * Ws + (x >> up))) : 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xv[r][j] = xg[(size_t)j * plane + off];
      if (BNB) av[r][j] = ag[(size_t)j * plane + off];
    }
  }
  // the group's constants: requested behind the patch (the wait of a scalar load under the RAW branch then overlaps its flight)
  __builtin_amdgcn_sched_barrier(0);
  float4 cc[8]; float c4[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cc[j] = make_float4(BNB ? 1.f : hsx, 0.f, 0.f, 0.f); c4[j] = 0.f; }
  if (src.mode != SC_SRC_RAW) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cc[j] = *reinterpret_cast<const float4*>(src.cst + (size_t)(g8 + j) * SC_CST);
      if (BNB) c4[j] = src.cst[(size_t)(g8 + j) * SC_CST + 4];
      else { cc[j].x *= hsx; cc[j].y *= hsx; }
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = e0 + TPG * r;
    uintx4 t0, t1;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * jp + h;
        const float t = BNB ? sc_pro_bnbwd(xv[r][j], av[BNB ? r : 0][j], cc[j].x, cc[j].y, cc[j].z, cc[j].w, c4[j], slo, shi)
                            : sc_pro_affine_h(xv[r][j], cc[j].x, cc[j].y, slo, shi);
        v[h] = okv[r] ? (BNB ? t * hsx : t) : 0.f;
      }
      unsigned a0, a1;
      split2h<BNB>(v[0], v[1], a0, a1);
      t0[jp] = a0; t1[jp] = a1;
    }
    if (e < NPX) { s_p[0][grp][e] = t0; s_p[1][grp][e] = t1; }
  }
  __syncthreads();

  // (sc_bnr_args) the raw values y of the tensor whose gradient this launch writes -- the lane's 4 x NPB output positions -- are requested
  // HERE, ahead of the MFMA phase: requested in the epilogue (round 5) their round trip was exposed at the end of every work-group's
  // life and the launch lost more (197 -> 302 us) than the BatchNorm-backward pass it replaces costs (120 us)
  float yv[BNB ? 4 : 1][BNB ? NPB : 1];
  if constexpr (BNB) {
    if (p.bnr_y != nullptr) {
      const float* const yn = p.bnr_y + (size_t)n * p.Cout * ((size_t)H * W);
      const unsigned HWy = (unsigned)((size_t)H * W);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
          const int co = 4 * lg + r, oy = y0 + 2 * wave + pb / PBW, ox = x0 + 16 * (pb % PBW) + l15;
          yv[r][pb] = (oy < H && ox < W && co < p.Cout) ? yn[(unsigned)co * HWy + (unsigned)(oy * W + ox)] : 0.f;
        }
    }
  }

  // ---- MFMAs: four 16-pixel blocks per wave, all K steps from LDS, filters from registers
  floatx4 acc[NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) acc[pb] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int gi = 4 * s + lg;
    const int gic = gi < NG ? gi : 0;            // padded K groups: zero filters, any finite patch entry
    const int tap = gic / NH, half = gic - tap * NH;
    const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      const int e = UPS ? ((2 * wave + pb / PBW + kh + 1) >> 1) * PC + ((16 * (pb % PBW) + l15 + kw + 1) >> 1)
                        : (2 * wave + pb / PBW + kh) * PC + 16 * (pb % PBW) + l15 + kw;
      const halfx8 b0 = __builtin_bit_cast(halfx8, s_p[0][half][e]);
      const halfx8 b1 = __builtin_bit_cast(halfx8, s_p[1][half][e]);
      const halfx8 a0 = __builtin_bit_cast(halfx8, A[s][0]);
      const halfx8 a1 = __builtin_bit_cast(halfx8, A[s][1]);
      acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[pb], 0, 0, 0);
      acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[pb], 0, 0, 0);
      acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[pb], 0, 0, 0);
    }
  }

  // ---- epilogue: store, per-cout sums of the wave's 2 x 32 pixels
  const size_t HWs = (size_t)H * W;
  const bool want_stats = p.stats != nullptr;
  // uniform image base + 32-bit lane BYTE offsets (Cout * H * W < 2^30, host-checked): the 64-bit form cost ~60 VALU per thread
  float* const outn = p.out0 + (size_t)n * p.Cout * HWs;
  const unsigned HWu = (unsigned)HWs;
  if constexpr (BNB) {
    if (p.bnr_y != nullptr) {
      // data gradient + the BatchNorm-backward sums of the tensor it belongs to (sc_bnr_args, see bx3_epilogue_bnr): the lane's 4 x NPB
      // raw values y are requested together, then stores / masks / sums
      const float blo = sc_act_lo(p.bnr_act), bhi = sc_act_hi(p.bnr_act);
      float mx = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * lg + r;
        const float4 cc = *reinterpret_cast<const float4*>(p.bnr_cst + (size_t)(co < p.Cout ? co : 0) * SC_CST);
        float sv[TW / 32], sq[TW / 32];
#pragma unroll
        for (int h = 0; h < TW / 32; ++h) { sv[h] = 0.f; sq[h] = 0.f; }
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
          const int oy = y0 + 2 * wave + pb / PBW, ox = x0 + 16 * (pb % PBW) + l15;
          const bool ok = oy < H && ox < W && co < p.Cout;
          const float v = ok ? acc[pb][r] * hinv : 0.f;
          if (ok) *reinterpret_cast<float*>(reinterpret_cast<char*>(outn) + ((unsigned)co * HWu + (unsigned)(oy * W + ox)) * 4u) = v;
          const float yh = fmaf(yv[r][pb], cc.x, cc.y);
          const float gq = (ok && yh > blo && yh < bhi) ? v : 0.f;
          sv[(pb % PBW) >> 1] += gq;
          sq[(pb % PBW) >> 1] = fmaf(gq, (yv[r][pb] - cc.z) * cc.w, sq[(pb % PBW) >> 1]);
          mx = fmaxf(mx, fabsf(gq * cc.x));
        }
#pragma unroll
        for (int h = 0; h < TW / 32; ++h) {
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { sv[h] += __shfl_xor(sv[h], o, 64); sq[h] += __shfl_xor(sq[h], o, 64); }
          if (l15 == 0) { s_red[wave][h][co][0] = sv[h]; s_red[wave][h][co][1] = sq[h]; }
        }
      }
      if (p.bnr_absmax) wave_absmax_to(mx, p.bnr_absmax);
      __syncthreads();
      const int rows4 = (H + 3) >> 2, tiles32 = (W + 31) >> 5;
      if (tid < 2 * (TW / 32) * 16 * 2) {
        const int hh = tid / ((TW / 32) * 32), h = (tid >> 5) % (TW / 32), col = (tid >> 1) & 15, k = tid & 1;
        const int t4 = 2 * ty + hh, t32 = tx * (TW / 32) + h;
        if (col < p.Cout && t4 < rows4 && t32 < tiles32) {
          const size_t row = ((size_t)n * rows4 + t4) * tiles32 + t32;
          p.bnr_rows[(row * p.Cout + col) * 2 + k] = s_red[2 * hh][h][col][k] + s_red[2 * hh + 1][h][col][k];
        }
      }
      return;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = 4 * lg + r;
    float sv[TW / 32], sq[TW / 32];             // per 32-pixel statistics tile (SC_STAT_CONV3 rows are 4 rows x 32 px)
#pragma unroll
    for (int h = 0; h < TW / 32; ++h) { sv[h] = 0.f; sq[h] = 0.f; }
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      const int oy = y0 + 2 * wave + pb / PBW, ox = x0 + 16 * (pb % PBW) + l15;
      const bool ok = oy < H && ox < W && co < p.Cout;
      const float v = ok ? acc[pb][r] * hinv : 0.f;
      sv[(pb % PBW) >> 1] += v; sq[(pb % PBW) >> 1] = fmaf(v, v, sq[(pb % PBW) >> 1]);
      if (ok) *reinterpret_cast<float*>(reinterpret_cast<char*>(outn) + ((unsigned)co * HWu + (unsigned)(oy * W + ox)) * 4u) = v;
    }
    if (want_stats) {
#pragma unroll
      for (int h = 0; h < TW / 32; ++h) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { sv[h] += __shfl_xor(sv[h], o, 64); sq[h] += __shfl_xor(sq[h], o, 64); }
        if (l15 == 0) { s_red[wave][h][co][0] = sv[h]; s_red[wave][h][co][1] = sq[h]; }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    const int rows4 = (H + 3) >> 2, tiles32 = (W + 31) >> 5;
    if (tid < 2 * (TW / 32) * 16 * 2) {
      const int hh = tid / ((TW / 32) * 32), h = (tid >> 5) % (TW / 32), col = (tid >> 1) & 15, k = tid & 1;
      const int t4 = 2 * ty + hh, t32 = tx * (TW / 32) + h;
      if (col < p.Cout && t4 < rows4 && t32 < tiles32) {
        const size_t row = ((size_t)n * rows4 + t4) * tiles32 + t32;
        p.stats[(row * p.Cout + col) * 2 + k] = s_red[2 * hh][h][col][k] + s_red[2 * hh + 1][h][col][k];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// decoder.blocks.4.conv1 (32 -> <= 16 channels, source = nearest 2x up-sampling of a half-resolution tensor) as FOUR 2x2 convolutions
// on the half-resolution source -- the sub-pixel form of conv_sp.hip on the thin layer's 16x16x32 MFMA.  Output pixel (2y + py, 2x + px)
// sees, through its 3 x 3 window on the up-sampled image, only the 2 x 2 source pixels (y + py - 1 + sy, x + px - 1 + sx), sy, sx in
// {0, 1}; the filter of slot (sy, sx) of phase (py, px) is the sum of the taps that land on that source pixel (kh: py = 0 -> {0}, {1, 2};
// py = 1 -> {0, 1}, {2}; kw alike) -- summed in fp32 and split by the pack (entries behind the 3 x 3 ones, pack_thin_item).
//   16 MFMA K steps (4 phases x 4 slots) per 16 source pixels = 64 output pixels instead of 4 x 9: 2.25x fewer MFMAs, and each of the
//   nine source positions is read from LDS ONCE per 16 source pixels for all the phases that use it: 4.5x fewer operand reads
//   (k_conv3_thin_h spent 61 us of LDS reads and 46 us of MFMAs on this layer at 16 x 512^2 beside its 95 us of memory traffic);
//   the patch is staged at source resolution (6 x 34 entries per 8-channel group: every source value prologue'd and split once).
//   B: lane (n = l&15, lg = l>>4) -> source pixel n of the wave's 16-pixel block, channels 8 lg .. +7; D: lane -> couts 4 lg + r of that
//   source pixel, one accumulator per phase: the two px phases of a row make one 8-byte store (output columns 2x, 2x + 1).
// Work-group = 4 waves, output tile 8 rows x 64 columns = source 4 x 32; wave w = source row w, two blocks of 16 source pixels.
// Statistics rows as k_conv3_thin_h (SC_STAT_CONV3: 4 rows x 32 pixels).
__global__ __launch_bounds__(256, 2) void k_conv3_thin_sp(const ConvXP p) {
  constexpr int PC = 34, NPX = 6 * PC, NPXP = (NPX + 15) & ~15, NH = 4, NR = (NPX + 63) / 64, STEPS3 = 9;
  __shared__ uintx4 s_p[2][NH][NPXP];
  __shared__ float s_red[4][2][16][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int H = p.H, W = p.W, Hs = H >> 1, Ws = W >> 1;
  const int tiles_x = (W + 63) >> 6;
  int n, tile;
  if (p.xcdmap) {      // each XCD walks a contiguous eighth of the pixel tiles (see k_conv3_bx3)
    const int per_img = tiles_x * ((H + 7) >> 3), total = per_img * p.N, per_xcd = (total + 7) >> 3;
    const int j = blockIdx.x >> 3, pt = (blockIdx.x & 7) * per_xcd + j;
    if (j >= per_xcd || pt >= total) return;
    n = pt / per_img; tile = pt - n * per_img;
  } else {
    n = blockIdx.z; tile = blockIdx.x;
  }
  n = __builtin_amdgcn_readfirstlane(n); tile = __builtin_amdgcn_readfirstlane(tile);
  const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
  const int y0 = ty * 8, x0 = tx * 64;
  const float hsx = h_act_scale(p.xb0, p.xb1);
  const float hinv = 1.f / (hsx * SC_H_SW);
  const SrcD& src = p.s0;

  // ---- stage the source patch: wave = one 8-channel group, lanes = consecutive patch entries (k_conv3_thin_h)
  const int g8 = wave * 8;
  const size_t plane = (size_t)Hs * Ws;
  const float slo = h_lo(sc_act_lo(src.act), hsx), shi = h_hi(sc_act_hi(src.act), hsx);
  float xv[NR][8];
  bool okv[NR];
  const float* const xg = src.x + ((size_t)n * 32 + g8) * plane;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = lane + 64 * r;
    const int pr = e / PC, pc = e - pr * PC;
    const int y = (y0 >> 1) - 1 + pr, x = (x0 >> 1) - 1 + pc;
    const bool ok = (e < NPX) && y >= 0 && y < Hs && x >= 0 && x < Ws;
    okv[r] = ok;
    const unsigned off = ok ? (unsigned)(y * Ws + x) : 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[r][j] = xg[(size_t)j * plane + off];
  }
  __builtin_amdgcn_sched_barrier(0);
  float2 cc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cc[j] = make_float2(hsx, 0.f);
  if (src.mode != SC_SRC_RAW) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cc[j] = *reinterpret_cast<const float2*>(src.cst + (size_t)(g8 + j) * SC_CST);
      cc[j].x *= hsx; cc[j].y *= hsx;
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = lane + 64 * r;
    uintx4 t0, t1;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const float v0 = okv[r] ? sc_pro_affine_h(xv[r][2 * jp], cc[2 * jp].x, cc[2 * jp].y, slo, shi) : 0.f;
      const float v1 = okv[r] ? sc_pro_affine_h(xv[r][2 * jp + 1], cc[2 * jp + 1].x, cc[2 * jp + 1].y, slo, shi) : 0.f;
      unsigned a0, a1;
      split2h<false>(v0, v1, a0, a1);
      t0[jp] = a0; t1[jp] = a1;
    }
    if (e < NPX) { s_p[0][wave][e] = t0; s_p[1][wave][e] = t1; }
  }
  // the phase filters (16 K steps x 2 terms, behind the 3 x 3 entries of the pack), requested while the patch settles
  uintx4 A[16][2];
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) A[s][t] = p.wpk[((STEPS3 + s) * 2 + t) * 64 + lane];
  __syncthreads();

  floatx4 acc[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) acc[q][ph] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dyi = 0; dyi < 3; ++dyi)
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = (wave + dyi) * PC + 16 * q + l15 + dxi;
        const halfx8 b0 = __builtin_bit_cast(halfx8, s_p[0][lg][e]);
        const halfx8 b1 = __builtin_bit_cast(halfx8, s_p[1][lg][e]);
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const int sy = dyi - py;                          // source row offset dyi - 1 = py - 1 + sy
          if (sy < 0 || sy > 1) continue;
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int sx = dxi - px;
            if (sx < 0 || sx > 1) continue;
            const int ph = 2 * py + px, s = 4 * ph + 2 * sy + sx;
            const halfx8 a0 = __builtin_bit_cast(halfx8, A[s][0]);
            const halfx8 a1 = __builtin_bit_cast(halfx8, A[s][1]);
            acc[q][ph] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[q][ph], 0, 0, 0);
            acc[q][ph] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[q][ph], 0, 0, 0);
            acc[q][ph] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[q][ph], 0, 0, 0);
          }
        }
      }

  // ---- epilogue: 8-byte stores (the two px phases), per-cout sums of the wave's 2 output rows x 32 columns per block
  const size_t HWs = (size_t)H * W;
  const bool want_stats = p.stats != nullptr;
  float* const outn = p.out0 + (size_t)n * p.Cout * HWs;
  const unsigned HWu = (unsigned)HWs;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = 4 * lg + r;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float sv = 0.f, sq = 0.f;
      const int ox = x0 + 32 * q + 2 * l15;
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const int oy = y0 + 2 * wave + py;
        const bool ok = oy < H && ox < W && co < p.Cout;     // (W is even: columns ox, ox + 1 are inside or outside together)
        const float v0 = ok ? acc[q][2 * py][r] * hinv : 0.f, v1 = ok ? acc[q][2 * py + 1][r] * hinv : 0.f;
        sv += v0 + v1; sq = fmaf(v0, v0, fmaf(v1, v1, sq));
        if (ok) *reinterpret_cast<float2*>(reinterpret_cast<char*>(outn) + ((unsigned)co * HWu + (unsigned)(oy * W + ox)) * 4u) = make_float2(v0, v1);
      }
      if (want_stats) {
        sv = row_sum16(sv); sq = row_sum16(sq);
        if (l15 == 0) { s_red[wave][q][co][0] = sv; s_red[wave][q][co][1] = sq; }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    const int rows4 = (H + 3) >> 2, tiles32 = (W + 31) >> 5;
    if (tid < 2 * 2 * 16 * 2) {
      const int hh = tid / 64, h = (tid >> 5) & 1, col = (tid >> 1) & 15, k = tid & 1;
      const int t4 = 2 * ty + hh, t32 = 2 * tx + h;
      if (col < p.Cout && t4 < rows4 && t32 < tiles32) {
        const size_t row = ((size_t)n * rows4 + t4) * tiles32 + t32;
        p.stats[(row * p.Cout + col) * 2 + k] = s_red[2 * hh][h][col][k] + s_red[2 * hh + 1][h][col][k];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// The data gradient of decoder.blocks.4.conv1 (16 -> 32 channels, stored at HALF resolution: the backward of the nearest 2x up-sampling
// is the 2 x 2 sum) in the sub-pixel form: a source pixel (ys, xs) feeds the output pixels (2 ys + a', 2 xs + b') through the taps that
// land on it, so its gradient gathers dy over the 4 x 4 window  a, b in {-1, 0, 1, 2}:
//     dx[ci][ys][xs] = sum_{a, b, co} Wd[a][b][ci][co] * dy[co][2 ys + a][2 xs + b],   Wd[a][b] = sum_{kh in S(a), kw in S(b)} w[co][ci][kh][kw],
//     S(-1) = {2}, S(0) = {1, 2}, S(1) = {0, 1}, S(2) = {0}
// -- K = 16 positions x 16 channels = 8 steps of the 16x16x32 MFMA for 2 x 16 output rows instead of four output pixels x 9 taps
// (2.25x fewer MFMAs than sc_conv3x3_bx3 with down0, which also leaves every other lane idle in its summing store), and every dy
// entry is read from LDS once.  The filters Wd are summed in fp32 and split by the pack (pack_thin_item, transpose_flip with 32 input
// channels).  dy = the BatchNorm-backward source (g, y) formed on load as in k_conv3_thin_h<16, true>.
//   A: lane (m = l&15, lg) -> Wd[pos = 2 s + (lg >> 1)][ci = 16 mb + m][co = 8 (lg & 1) .. +7];  B: lane (n, lg) -> dy entry of source
//   pixel n at that position, channel half lg & 1.  The patch (10 rows x 66 columns at full resolution) is stored with its columns
//   DE-INTERLEAVED by parity, so the 16 source pixels of a block read 16 consecutive entries (stride-2 columns otherwise).
// Work-group = 4 waves, source tile 4 rows x 32 columns (= 8 x 64 of dy); wave w = source row w, two blocks of 16 source pixels.
__global__ __launch_bounds__(256, 2) void k_conv3_thin_spd(const ConvXP p) {
  constexpr int PR = 10, PC = 66, NPX = PR * PC, PCH = 34, NPL = PR * PCH, NPLP = (NPL + 15) & ~15;      // per parity plane: 10 x 34 entries
  constexpr int TPG = 128, NR = (NPX + TPG - 1) / TPG;
  __shared__ uintx4 s_p[2][2][2][NPLP];          // [term][channel half][column parity][row * 34 + column / 2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int H = p.H, W = p.W, Hs = H >> 1, Ws = W >> 1;
  const int tiles_x = (W + 63) >> 6;
  int n, tile;
  if (p.xcdmap) {
    const int per_img = tiles_x * ((H + 7) >> 3), total = per_img * p.N, per_xcd = (total + 7) >> 3;
    const int j = blockIdx.x >> 3, pt = (blockIdx.x & 7) * per_xcd + j;
    if (j >= per_xcd || pt >= total) return;
    n = pt / per_img; tile = pt - n * per_img;
  } else {
    n = blockIdx.z; tile = blockIdx.x;
  }
  n = __builtin_amdgcn_readfirstlane(n); tile = __builtin_amdgcn_readfirstlane(tile);
  const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
  const int y0 = ty * 8, x0 = tx * 64;                       // full-resolution origin of the tile (source origin y0 / 2, x0 / 2)
  const float hsx = h_grad_scale(p.absmax);
  const float hinv = 1.f / (hsx * SC_H_SW);
  const SrcD& src = p.s0;

  // ---- stage dy: two waves per 8-channel group, lanes = consecutive patch pixels (k_conv3_thin_h<16, true>)
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 1);
  const int g8 = grp * 8;
  const int e0 = tid - grp * TPG;
  const size_t plane = (size_t)H * W;
  const float slo = sc_act_lo(src.act), shi = sc_act_hi(src.act);
  float xv[NR][8], av[NR][8];
  bool okv[NR];
  const float* const xg = src.x + ((size_t)n * 16 + g8) * plane;
  const float* const ag = src.aux + ((size_t)n * 16 + g8) * plane;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = e0 + TPG * r;
    const int pr = e / PC, pc = e - pr * PC;
    const int y = y0 - 1 + pr, x = x0 - 1 + pc;
    const bool ok = (e < NPX) && y >= 0 && y < H && x >= 0 && x < W;
    okv[r] = ok;
    const unsigned off = ok ? (unsigned)(y * W + x) : 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) { xv[r][j] = xg[(size_t)j * plane + off]; av[r][j] = ag[(size_t)j * plane + off]; }
  }
  __builtin_amdgcn_sched_barrier(0);
  float4 cc[8]; float c4[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cc[j] = *reinterpret_cast<const float4*>(src.cst + (size_t)(g8 + j) * SC_CST);
    c4[j] = src.cst[(size_t)(g8 + j) * SC_CST + 4];
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = e0 + TPG * r;
    const int pr = e / PC, pc = e - pr * PC;
    uintx4 t0, t1;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * jp + h;
        const float t = sc_pro_bnbwd(xv[r][j], av[r][j], cc[j].x, cc[j].y, cc[j].z, cc[j].w, c4[j], slo, shi);
        v[h] = okv[r] ? t * hsx : 0.f;
      }
      unsigned a0, a1;
      split2h<true>(v[0], v[1], a0, a1);
      t0[jp] = a0; t1[jp] = a1;
    }
    if (e < NPX) { const int d = pr * PCH + (pc >> 1); s_p[0][grp][pc & 1][d] = t0; s_p[1][grp][pc & 1][d] = t1; }
  }
  // the gathered filters: 8 K steps x 2 row blocks x 2 terms
  uintx4 A[8][2][2];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int t = 0; t < 2; ++t) A[s][mb][t] = p.wpk[(((s * 2 + mb) * 2) + t) * 64 + lane];
  __syncthreads();

  floatx4 acc[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc[q][mb] = (floatx4){0.f, 0.f, 0.f, 0.f};
  // step s: positions 2 s, 2 s + 1 -> row a = (s >> 1) - 1, columns b = 2 (s & 1) - 1 + (lg >> 1): patch row 2 w + a + 1, patch column
  // 32 q + 2 n + b + 1 -> parity lg >> 1 (b = -1, 1: even columns; 0, 2: odd), entry 16 q + n + (s & 1) of that parity plane
  const int half = lg & 1, par = lg >> 1;
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int d = (2 * wave + (s >> 1)) * PCH + 16 * q + l15 + (s & 1);
      const halfx8 b0 = __builtin_bit_cast(halfx8, s_p[0][half][par][d]);
      const halfx8 b1 = __builtin_bit_cast(halfx8, s_p[1][half][par][d]);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const halfx8 a0 = __builtin_bit_cast(halfx8, A[s][mb][0]);
        const halfx8 a1 = __builtin_bit_cast(halfx8, A[s][mb][1]);
        acc[q][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[q][mb], 0, 0, 0);
        acc[q][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[q][mb], 0, 0, 0);
        acc[q][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[q][mb], 0, 0, 0);
      }
    }
  // ---- store at half resolution (optionally accumulating)
  const size_t HWs = (size_t)Hs * Ws;
  float* const outn = p.out0 + (size_t)n * p.Cout * HWs;
  const int ys = (y0 >> 1) + wave;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int xs = (x0 >> 1) + 16 * q + l15;
    if (ys < Hs && xs < Ws) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = 16 * mb + 4 * lg + r;
          if (ci < p.Cout) {
            float* o = outn + (size_t)ci * HWs + (size_t)ys * Ws + xs;
            const float v = acc[q][mb][r] * hinv;
            *o = p.accum0 ? *o + v : v;
          }
        }
    }
  }
}

// filters of a thin layer in the register layout of k_conv3_thin_h: entry ((s*2 + term)*64 + lane) of 8 halves
__device__ __forceinline__ void pack_thin_item(const float* __restrict__ w, unsigned short* __restrict__ out, size_t i, int Cout, int Cin,
                                               int tflip) {
  const int M = tflip ? Cin : Cout, K = tflip ? Cout : Cin;      // rows (<= 16), reduction channels (16 | 32)
  const int NH = K / 8, NG = 9 * NH;
  const int j = (int)(i & 7);
  const int lane = (int)((i >> 3) & 63);
  const int s = (int)(i >> 9);
  const int m = lane & 15, gi = 4 * s + (lane >> 4);
  float v = 0.f;
  if (tflip && M == 32) {
    // k_conv3_thin_spd (decoder.blocks.4.conv1's data gradient: K = 16 dy channels, M = 32 rows): 16 entries (K step, row block) x 2 terms
    //   value = Wd[pos = 2 ks + (lg >> 1)][ci = 16 mb + m][co = 8 (lg & 1) + j] = sum of the taps kh in S(a), kw in S(b), pos = 4 (a + 1) + b + 1
    const int ks = s >> 1, mb = s & 1, lgp = lane >> 4, pos = 2 * ks + (lgp >> 1), ai = pos >> 2, bi = pos & 3;
    const int ci = 16 * mb + m, co = 8 * (lgp & 1) + j;
    const int kh0 = ai == 0 ? 2 : (ai == 1 ? 1 : 0), kh1 = ai == 0 ? 2 : (ai == 1 ? 2 : (ai == 2 ? 1 : 0));
    const int kw0 = bi == 0 ? 2 : (bi == 1 ? 1 : 0), kw1 = bi == 0 ? 2 : (bi == 1 ? 2 : (bi == 2 ? 1 : 0));
    for (int kh = kh0; kh <= kh1; ++kh)
      for (int kw = kw0; kw <= kw1; ++kw) v += w[((size_t)co * M + ci) * 9 + kh * 3 + kw];
    unsigned short tq[3];
    split_filter(v, true, tq);
    out[((size_t)(s * 2 + 0) * 64 + lane) * 8 + j] = tq[0];
    out[((size_t)(s * 2 + 1) * 64 + lane) * 8 + j] = tq[1];
    return;
  }
  const int steps3 = (NG + 3) / 4;
  if (s >= steps3) {
    // the 16 phase filters of k_conv3_thin_sp (forward, 32 input channels): step = 4 * (2 py + px) + 2 sy + sx, value = the fp32 sum of the
    // taps (kh, kw) whose up-sampled position falls on source pixel (py - 1 + sy, px - 1 + sx)
    const int sp = s - steps3, ph = sp >> 2, py = ph >> 1, px = ph & 1, sy = (sp >> 1) & 1, sx = sp & 1;
    const int kh0 = (py == 0) ? (sy == 0 ? 0 : 1) : (sy == 0 ? 0 : 2), kh1 = (py == 0) ? (sy == 0 ? 0 : 2) : (sy == 0 ? 1 : 2);
    const int kw0 = (px == 0) ? (sx == 0 ? 0 : 1) : (sx == 0 ? 0 : 2), kw1 = (px == 0) ? (sx == 0 ? 0 : 2) : (sx == 0 ? 1 : 2);
    const int k = (lane >> 4) * 8 + j;
    if (m < M) {
      for (int kh = kh0; kh <= kh1; ++kh)
        for (int kw = kw0; kw <= kw1; ++kw) v += w[((size_t)m * K + k) * 9 + kh * 3 + kw];
    }
  } else if (gi < NG && m < M) {
    const int tap = gi / NH, k = (gi - tap * NH) * 8 + j;
    v = tflip ? w[((size_t)k * M + m) * 9 + (8 - tap)] : w[((size_t)m * K + k) * 9 + tap];
  }
  unsigned short t[3];
  split_filter(v, true, t);
  out[((size_t)(s * 2 + 0) * 64 + lane) * 8 + j] = t[0];
  out[((size_t)(s * 2 + 1) * 64 + lane) * 8 + j] = t[1];
}
__global__ void k_pack_weights_thin(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout, int Cin, int tflip, size_t total) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < total) pack_thin_item(w, wpk, i, Cout, Cin, tflip);
}

__global__ void k_pack_weights_bx3(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout, int Cin,
                                   int co_t, int tflip, int nchunk, int nt, int half, size_t total) {
  const int M = tflip ? Cin : Cout, K = tflip ? Cout : Cin;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int j = (int)(r % 8); r /= 8;
    const int col = (int)(r % co_t); r /= co_t;
    const int hf = (int)(r % 2); r /= 2;
    const int kw = (int)(r % 3); r /= 3;
    const int kh = (int)(r % 3); r /= 3;
    const int chunk = (int)(r % nchunk);
    const int mt = (int)(r / nchunk);
    const int m = mt * co_t + col, k = chunk * 16 + hf * 8 + j, tap = kh * 3 + kw;
    float v = 0.f;
    if (m < M && k < K) v = tflip ? w[((size_t)k * M + m) * 9 + (8 - tap)] : w[((size_t)m * K + k) * 9 + tap];
    unsigned short t[3];
    split_filter(v, half != 0, t);
    const size_t stage = ((size_t)mt * nchunk + chunk) * 3 + kh;
    for (int c = 0; c < nt; ++c) {
      const size_t d = ((((stage * nt + c) * 3 + kw) * 2 + hf) * co_t + col) * 8 + j;
      wpk[d] = t[c];
    }
  }
}


// ------------------------------------------------------------------------------------------
// Weight gradient of the same convolution on the split-bf16 MFMA.
// GEMM view (per tap):  D[co][ci] = sum_px dy[co][px] * in[ci][px + d(tap)],  K step = 16 pixels of one image row
//   A (32 x 16): lane l -> dy[co = l&31][px = 16j + 8*(l>>5) .. +7]            (one 16-byte LDS read per term)
//   B (16 x 32): lane l -> in[ci = l&31][px + kw .. +7]: an aligned 16-byte read + the next dword; kw = 1 is funnel-
//                shifted in registers (v_alignbit), kw = 2 is a register renaming
// Work-group = 12 waves; tile = (32*WM couts) x 64 cins x 9 taps; a stage = 2 image rows x 32 columns of one image.
// Wave = one (cout block, cin block) pair x one filter row kh (3 taps, 48 accumulator registers);
// WM = 2: 4 pairs x 3 kh, both rows of the stage; WM = 1: 2 pairs x 3 kh x 2 rows (two K parts).
// The input rows live in a 4-slot ring per channel (slot = (row + 1) & 3): walking down a 32-column strip only the two
// new rows are fetched per stage, every element is split into its three bf16 terms once.
struct WgradXP {
  const float* absmax;
  const float* xb0; const float* xb1;      // activation bounds of the input sources (h_act_scale)
  SrcD dy, s0, s1;
  int N, H, W, Cout, Cin;
  float* part;
  int nsl;          // K slices (gridDim.x)
  int CoP, CiP;     // padded dims of the partial buffer
};

// NCI: 32-wide cin blocks per tile (2: 64 cins; 1: 32 cins for channel counts that would waste most of a 64-wide tile)
// PIPE (fp16 mode): one barrier per stage.  The dy tile is double-buffered and the input ring has 8 slots, so stage t+1 is
// split and stored while other waves still run the MFMAs of stage t (the matrix pipe and the VALU overlap across the three
// waves of a SIMD instead of alternating in lock step), and every item's global load for stage t+2 is issued right after
// its registers were consumed for stage t+1: a whole stage in flight with no extra registers.
template <int WM, int NT, int NCI, bool HF = false, bool PIPE = false>
__global__ __launch_bounds__(768, 1) void k_wgrad3_bx3(const WgradXP p) {
  static_assert(!HF || NT == 2, "the fp16 mode has two terms");
  static_assert(!PIPE || HF, "the pipelined variant is the fp16 mode's");
  constexpr int NTH = 768;
  const float hsg = HF ? h_grad_scale(p.absmax) : 1.f;          // fp16 mode: scale of the gradient operand
  const float hsa = HF ? h_act_scale(p.xb0, p.xb1) : 1.f;        // ... and of the activation operand
  const float hinv = HF ? 1.f / (hsg * hsa) : 1.f;
  // KP K parts: rows and, at KP = 4, 16-pixel steps.  NCI = 3 (WM = 1: a 96-wide cin tile for 65..96 input channels -- decoder.blocks.3.conv1's
  // 64 up-sampled + 16 skip channels in ONE pass over dy instead of two): 3 pairs x 3 filter rows = 9 computing waves, each both rows and
  // both 16-pixel steps of a stage (KP = 1); the other three waves only stage
  constexpr int COT = 32 * WM, CIT = 32 * NCI, NPAIR = NCI * WM, KP = NPAIR == 3 ? 1 : 4 / NPAIR;
  static_assert(NPAIR == 1 || NPAIR == 2 || NPAIR == 3 || NPAIR == 4, "wave roles");
  constexpr int DYP = 72;              // dy pitch per cout in pixels (144 B: conflict-free 16-byte reads)
  constexpr int XRP = 40;              // input row pitch in pixels (34 used)
  constexpr int RING = PIPE ? 8 : 4;
  constexpr int XCP = RING * XRP + 8;  // input pitch per cin: ring rows + pad (336 / 656 B: conflict-free)
  constexpr int NDY = (COT * 32 + NTH - 1) / NTH;     // dy pixel pairs per thread per stage
  constexpr int NXI = (2 * 17 * CIT + NTH - 1) / NTH; // input pixel pairs per thread per two rows

  __shared__ __attribute__((aligned(16))) unsigned s_dy[PIPE ? 2 : 1][NT][COT * DYP / 2];
  __shared__ __attribute__((aligned(16))) unsigned s_x[NT][CIT * XCP / 2];
  __shared__ __attribute__((aligned(16))) float s_ca[COT * SC_CST];
  __shared__ __attribute__((aligned(16))) float s_cb[CIT * 4];      // scale, shift, lo, hi per cin

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int pair = wave % NPAIR, kh = (wave / NPAIR) % 3, kp = wave / (3 * NPAIR);
  const bool computing = wave < 3 * NPAIR * KP;            // (all twelve waves except for NPAIR = 3)
  const int wm = pair % WM, wn = pair / WM;
  const int cit = blockIdx.y, cot = blockIdx.z;
  const int H = p.H, W = p.W;
  const int C0 = p.s0.C;

  for (int i = tid; i < COT * SC_CST; i += NTH) {
    const int ch = cot * COT + i / SC_CST;
    // (the operand scale rides in the constants, see split2h: A, B, D of the BatchNorm-backward form, scale / shift otherwise)
    const int f = i % SC_CST;
    const float fs = (p.dy.mode == SC_SRC_BNBWD ? (f >= 2 && f <= 4) : (f < 2)) ? hsg : 1.f;
    s_ca[i] = fs * ((p.dy.cst && p.dy.mode != SC_SRC_RAW && ch < p.Cout) ? p.dy.cst[(size_t)ch * SC_CST + f] : (f == 0 ? 1.f : 0.f));
  }
  for (int i = tid; i < CIT; i += NTH) {
    const int ch = cit * CIT + i;
    float sc = 1.f, sh = 0.f, lo = -__builtin_inff(), hi = __builtin_inff();
    if (ch < p.Cin) {
      const bool second = ch >= C0;
      const float* cp = second ? p.s1.cst : p.s0.cst;
      const int md = second ? p.s1.mode : p.s0.mode;
      const int act = second ? p.s1.act : p.s0.act;
      if (cp && md != SC_SRC_RAW) { sc = cp[(size_t)(second ? ch - C0 : ch) * SC_CST]; sh = cp[(size_t)(second ? ch - C0 : ch) * SC_CST + 1]; }
      lo = sc_act_lo(act); hi = sc_act_hi(act);
    }
    if constexpr (HF) { sc *= hsa; sh *= hsa; lo = h_lo(lo, hsa); hi = h_hi(hi, hsa); }      // (see split2h)
    s_cb[i * 4] = sc; s_cb[i * 4 + 1] = sh; s_cb[i * 4 + 2] = lo; s_cb[i * 4 + 3] = hi;
  }

  floatx16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // stage enumeration: strips (image, 32-column strip) outer, row pairs inner
  // PIPE: every strip starts with a pre-stage at y0 = -2 that only brings rows -1 (zeros) and 0 into the ring, so that EVERY
  // stage fetches exactly two new input rows and the loop body has no conditional loads
  const int tiles_x = (W + 31) >> 5, RS = (H + 1) >> 1, RSS = PIPE ? RS + 1 : RS, YB = PIPE ? -2 : 0;
  const long T = (long)p.N * tiles_x * RSS;
  const long t_begin = T * blockIdx.x / p.nsl, t_end = T * (blockIdx.x + 1) / p.nsl;
  const int dymode = PIPE ? (int)SC_SRC_BNBWD : p.dy.mode;      // the pipelined variant is launched for BatchNorm-backward gradients only
  const float dls = dymode == SC_SRC_BNBWD ? 1.f : hsg;         // (limits of an affine gradient source scale with its constants)
  const float dlo = sc_act_lo(p.dy.act) * dls, dhi = sc_act_hi(p.dy.act) * dls;
  const size_t HW = (size_t)H * W;

  auto decode = [&](long t, int& n, int& y0, int& x0) {
    const int strip = (int)(t / RSS);
    const int ty = (int)(t - (long)strip * RSS);
    n = strip / tiles_x;
    x0 = (strip - n * tiles_x) * 32;
    y0 = ty * 2 + YB;
  };

  // ---- dy: thread owns pixel pairs (co = item >> 5, row = (item >> 4) & 1, cols 2*(item & 15), +1) ----
  float dg[NDY][2], dv[NDY][2];
  // PIPE: per-item address invariants in registers: the byte offset of the item's gradient channel plane (+ column), the
  // input item's channel plane pointer and which source it belongs to
  unsigned dyo[NDY];
  const char* xch[(2 * 17 * CIT + NTH - 1) / NTH];
  unsigned xsec = 0;
  unsigned istr0 = 0, istr1 = 0;
  if constexpr (PIPE) {
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
      int it = tid + NTH * k;
      if (it >= COT * 32) it -= COT * 32;
      const int co = cot * COT + ((it >> 5) & (COT - 1));
      dyo[k] = (unsigned)(co < p.Cout ? co : 0) * (unsigned)HW * 4u;
    }
    istr0 = (unsigned)((size_t)p.s0.C * (H >> p.s0.up) * (W >> p.s0.up) * 4);
    istr1 = (unsigned)((size_t)p.s1.C * (H >> p.s1.up) * (W >> p.s1.up) * 4);
#pragma unroll
    for (int k = 0; k < NXI; ++k) {
      int it = tid + NTH * k;
      if (it >= 2 * 17 * CIT) it -= 2 * 17 * CIT;
      const int cil = ((it / 17) >> 1) % CIT;
      const int chr = cit * CIT + cil;
      const int ch = chr < p.Cin ? chr : 0;
      const bool second = ch >= C0;
      const int cs = second ? ch - C0 : ch;
      const int up = second ? p.s1.up : p.s0.up;
      xch[k] = reinterpret_cast<const char*>((second ? p.s1.x : p.s0.x) + (size_t)cs * ((size_t)(H >> up) * (W >> up)));
      xsec |= second ? (1u << k) : 0u;
    }
  }
  // PIPE: no guarded items (the compiler's s_waitcnt counting gives up at every exec-mask branch): threads past the last item
  // redo one of the first items and store the same values to the same place
  auto dy_load_item = [&](int k, int n, int y0, int x0) {
    {
      int it = tid + NTH * k;
      if (PIPE && it >= COT * 32) it -= COT * 32;
      const int co = cot * COT + ((it >> 5) & (COT - 1)), row = (it >> 4) & 1, col = 2 * (it & 15);
      const int y = y0 + row, x = x0 + col;
      const bool okc = co < p.Cout && y < H && y >= 0;
      if constexpr (PIPE) {
        // one 8-byte load per tensor: W and col are even (host-checked), the byte offset within the image fits 32 bits
        const size_t io = (size_t)n * p.Cout * HW * 4;
        const unsigned off = dyo[k] + (unsigned)(((y >= 0 && y < H) ? y : 0) * W + ((x < W) ? x : 0)) * 4u;
        const float2 g = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p.dy.x) + io + off);
        const float2 a = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p.dy.aux) + io + off);
        dg[k][0] = g.x; dg[k][1] = g.y; dv[k][0] = a.x; dv[k][1] = a.y;
        return;
      }
      // uniform image base + 32-bit lane offset (< Cout * H * W): saddr loads, no 64-bit lane arithmetic
      const float* const gx = p.dy.x + (size_t)n * p.Cout * HW;
      const unsigned base = (unsigned)(okc ? co : 0) * (unsigned)HW + (unsigned)((okc ? y : 0) * W);
      const unsigned xa = (x < W) ? x : 0, xb = (x + 1 < W) ? x + 1 : 0;
      dg[k][0] = gx[base + xa]; dg[k][1] = gx[base + xb];
      if (dymode == SC_SRC_BNBWD) {
        const float* const ga = p.dy.aux + (size_t)n * p.Cout * HW;
        dv[k][0] = ga[base + xa]; dv[k][1] = ga[base + xb];
      } else { dv[k][0] = 0.f; dv[k][1] = 0.f; }
    }
  };
  auto dy_load = [&](int n, int y0, int x0) {
#pragma unroll
    for (int k = 0; k < NDY; ++k) dy_load_item(k, n, y0, x0);
  };
  auto dy_store_item = [&](int k, int buf, int y0, int x0) {
    {
      int it = tid + NTH * k;
      if (PIPE && it >= COT * 32) it -= COT * 32;
      const int col_l = (it >> 5) & (COT - 1), row = (it >> 4) & 1, col = 2 * (it & 15);
      const int y = y0 + row, x = x0 + col;
      const bool okc = (cot * COT + col_l < p.Cout) && y < H && y >= 0;
      const float4 c0 = *reinterpret_cast<const float4*>(&s_ca[col_l * SC_CST]);
      const float c4 = s_ca[col_l * SC_CST + 4];
      float v0, v1;
      if (dymode == SC_SRC_BNBWD) {
        v0 = sc_pro_bnbwd(dg[k][0], dv[k][0], c0.x, c0.y, c0.z, c0.w, c4, dlo, dhi);
        v1 = sc_pro_bnbwd(dg[k][1], dv[k][1], c0.x, c0.y, c0.z, c0.w, c4, dlo, dhi);
      } else {
        v0 = sc_pro_affine(dg[k][0], c0.x, c0.y, dlo, dhi);
        v1 = sc_pro_affine(dg[k][1], c0.x, c0.y, dlo, dhi);
      }
      v0 = (okc && x < W) ? v0 : 0.f;
      v1 = (okc && x + 1 < W) ? v1 : 0.f;
      unsigned t[3];
      if constexpr (HF) { split2h(v0, v1, t[0], t[1]); t[2] = 0u; }
      else split3x2(v0, v1, t[0], t[1], t[2]);
      const int d = (col_l * DYP + row * 32 + col) >> 1;
      if (PIPE || it < COT * 32) {
#pragma unroll
        for (int c = 0; c < NT; ++c) s_dy[buf][c][d] = t[c];
      }
    }
  };
  auto dy_store = [&](int buf, int y0, int x0) {
#pragma unroll
    for (int k = 0; k < NDY; ++k) dy_store_item(k, buf, y0, x0);
  };

  // ---- input rows R, R+1 (R may be -1): item -> (pair of columns pr in 0..16, row, ci) ----
  float xr[NXI][2];
  auto x_load_item = [&](float (&xr)[NXI][2], int k, int n, int R, int x0) {
    {
      int it = tid + NTH * k;
      if (PIPE && it >= 2 * 17 * CIT) it -= 2 * 17 * CIT;
      const int rc = it / 17, pr = it - rc * 17;
      const int rowi = rc & 1, cil = (rc >> 1) % CIT;
      if constexpr (PIPE) {
        const bool sec = (xsec >> k) & 1u;
        const int up = sec ? p.s1.up : p.s0.up;
        const int y = R + rowi, x = x0 - 1 + 2 * pr;
        const int yc = (y >= 0 && y < H) ? y : 0;
        const int xa = (x >= 0 && x < W) ? x : 0, xb = (x + 1 < W) ? x + 1 : 0;
        const unsigned ro = (unsigned)n * (sec ? istr1 : istr0) + (unsigned)((yc >> up) * (W >> up)) * 4u;
        xr[k][0] = *reinterpret_cast<const float*>(xch[k] + (ro + (unsigned)(xa >> up) * 4u));
        xr[k][1] = *reinterpret_cast<const float*>(xch[k] + (ro + (unsigned)(xb >> up) * 4u));
        return;
      }
      const int chr = cit * CIT + cil;
      const int ch = chr < p.Cin ? chr : 0;
      const bool second = ch >= C0;
      const int cs = second ? ch - C0 : ch;
      const int Cs = second ? p.s1.C : p.s0.C;
      const int up = second ? p.s1.up : p.s0.up;
      const int y = R + rowi, x = x0 - 1 + 2 * pr;
      const bool oky = (y >= 0) && (y < H);
      const int Ws = W >> up;
      const float* xp = (second ? p.s1.x : p.s0.x) + ((size_t)n * Cs + cs) * ((size_t)(H >> up) * Ws) + (size_t)((oky ? y : 0) >> up) * Ws;
      const int xa = (x >= 0 && x < W) ? x : 0, xb = (x + 1 < W) ? x + 1 : 0;
      xr[k][0] = xp[xa >> up]; xr[k][1] = xp[xb >> up];
    }
  };
  auto x_load = [&](float (&xr)[NXI][2], int n, int R, int x0) {
#pragma unroll
    for (int k = 0; k < NXI; ++k) x_load_item(xr, k, n, R, x0);
  };
  // sl0: ring slot of row R (PIPE: a running position, see the loop; otherwise (R + 1) & 3)
  auto x_store_item = [&](const float (&xr)[NXI][2], int k, int R, int x0, int sl0) {
    {
      int it = tid + NTH * k;
      if (PIPE && it >= 2 * 17 * CIT) it -= 2 * 17 * CIT;
      const int rc = it / 17, pr = it - rc * 17;
      const int rowi = rc & 1, cil = (rc >> 1) % CIT;
      const int y = R + rowi, x = x0 - 1 + 2 * pr;
      const bool okc = (cit * CIT + cil < p.Cin) && (y >= 0) && (y < H);
      const float4 c = *reinterpret_cast<const float4*>(&s_cb[cil * 4]);
      float v0 = HF ? sc_pro_affine_h(xr[k][0], c.x, c.y, c.z, c.w) : sc_pro_affine(xr[k][0], c.x, c.y, c.z, c.w);
      float v1 = HF ? sc_pro_affine_h(xr[k][1], c.x, c.y, c.z, c.w) : sc_pro_affine(xr[k][1], c.x, c.y, c.z, c.w);
      v0 = (okc && x >= 0 && x < W) ? v0 : 0.f;
      v1 = (okc && x + 1 < W) ? v1 : 0.f;
      unsigned t[3];
      if constexpr (HF) { split2h<false>(v0, v1, t[0], t[1]); t[2] = 0u; }
      else split3x2(v0, v1, t[0], t[1], t[2]);
      const int slot = (sl0 + rowi) & (RING - 1);
      const int d = ((cil * XCP + slot * XRP) >> 1) + pr;
      if (PIPE || it < 2 * 17 * CIT) {
#pragma unroll
        for (int c = 0; c < NT; ++c) s_x[c][d] = t[c];
      }
    }
  };
  auto x_store = [&](const float (&xr)[NXI][2], int R, int x0, int sl0) {
#pragma unroll
    for (int k = 0; k < NXI; ++k) x_store_item(xr, k, R, x0, sl0);
  };

  const unsigned opaque_zero = (unsigned)p.H >> 30;      // 0 (host: H < 2^30), unknown to the compiler
  auto compute = [&](int sl0, int buf) {          // sl0: ring slot of image row y0 - 1
    if (NPAIR == 3 && !computing) return;
#pragma unroll
    for (int rr = 0; rr < (KP == 1 ? 2 : 1); ++rr) {
      const int r = (KP == 1) ? rr : (kp & 1);
      const int slot = (sl0 + r + kh) & (RING - 1);
#pragma unroll
      for (int jj = 0; jj < (KP == 4 ? 1 : 2); ++jj) {
        const int j = (KP == 4) ? (kp >> 1) : jj;
        bf16x8 A[NT];
        const int da = ((wm * 32 + l31) * DYP + r * 32 + 16 * j + 8 * lhi) >> 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) A[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uintx4*>(&s_dy[buf][t][da]));
        const int dx = ((wn * 32 + l31) * XCP + slot * XRP + 16 * j + 8 * lhi) >> 1;
        bf16x8 B[3][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const uintx4 X0 = *reinterpret_cast<const uintx4*>(&s_x[t][dx]);
          // the dword after the chunk: read as part of the NEXT aligned 16 bytes.  A ds_read_b32 of one dword per cin row
          // hits only 8 of the 32 banks (rows are 16-byte aligned: 4-way conflict, 8 LDS cycles); a second ds_read_b128 is
          // conflict-free (4 cycles) -- measured by SQ_LDS_BANK_CONFLICT: 56 % of this kernel's LDS cycles before
          // (round 6: hipcc NARROWS an element-0 use of a 16-byte LDS read to ds_read_b32 -- the elimination builds under the counters,
          // tools/exp_wgrad3_pmc.sh, put 76 % of this kernel's bank-conflict cycles and a third of its LDS-active cycles on exactly
          // this read.  The second dword is kept alive through an AND with a zero the compiler cannot prove, one v_and_or_b32.)
          const uintx4 X1v = *reinterpret_cast<const uintx4*>(&s_x[t][dx + 4]);
          const unsigned X1 = X1v[0] | (X1v[1] & opaque_zero);
          uintx4 S1, S2;
          S1[0] = __builtin_amdgcn_alignbit(X0[1], X0[0], 16);
          S1[1] = __builtin_amdgcn_alignbit(X0[2], X0[1], 16);
          S1[2] = __builtin_amdgcn_alignbit(X0[3], X0[2], 16);
          S1[3] = __builtin_amdgcn_alignbit(X1, X0[3], 16);
          S2[0] = X0[1]; S2[1] = X0[2]; S2[2] = X0[3]; S2[3] = X1;
          B[0][t] = __builtin_bit_cast(bf16x8, X0);
          B[1][t] = __builtin_bit_cast(bf16x8, S1);
          B[2][t] = __builtin_bit_cast(bf16x8, S2);
        }
        // the six partial products, taps interleaved so that consecutive MFMAs hit different accumulators
#define SC_BX3_STEP(TA, TB)                                                                                         \
  _Pragma("unroll") for (int kw = 0; kw < 3; ++kw)                                                                  \
      acc[kw] = mfma_split<HF>(A[TA], B[kw][TB], acc[kw]);
        if constexpr (NT == 3) { SC_BX3_STEP(1, 1) SC_BX3_STEP(2, 0) SC_BX3_STEP(0, 2) SC_BX3_STEP(1, 0) SC_BX3_STEP(0, 1) }
        if constexpr (NT == 2) { SC_BX3_STEP(0, 1) SC_BX3_STEP(1, 0) }
        SC_BX3_STEP(0, 0)
#undef SC_BX3_STEP
      }
    }
  };

  __syncthreads();          // constants in LDS
  // stage coordinates of t and t+1, advanced incrementally (decode() is a 64-bit division: ~200 scalar instructions)
  int n = 0, y0 = 0, x0 = 0;
  if (t_begin < t_end) decode(t_begin, n, y0, x0);
  n = __builtin_amdgcn_readfirstlane(n); y0 = __builtin_amdgcn_readfirstlane(y0); x0 = __builtin_amdgcn_readfirstlane(x0);   // uniform: scalar address arithmetic
  int n1 = n, y1 = y0, x1 = x0;
  auto advance = [&](int& nn, int& yy, int& xx) {
    yy += 2;
    if (yy >= 2 * RS) { yy = YB; xx += 32; if (xx >= 32 * tiles_x) { xx = 0; ++nn; } }
  };
  advance(n1, y1, x1);
  if constexpr (!PIPE) {
    if (t_begin < t_end) {
      x_load(xr, n, y0 - 1, x0);
      dy_load(n, y0, x0);
      x_store(xr, y0 - 1, x0, y0 & 3);
      x_load(xr, n, y0 + 1, x0);
      dy_store(0, y0, x0);
      x_store(xr, y0 + 1, x0, (y0 + 2) & 3);
      __syncthreads();
    }
    for (long t = t_begin; t < t_end; ++t) {
      const bool more = (t + 1) < t_end;
      if (more) {
        dy_load(n1, y1, x1);
        x_load(xr, n1, y1 + 1, x1);
      }
      compute(y0 & 3, 0);
      __syncthreads();
      if (more) {
        dy_store(0, y1, x1);
        x_store(xr, y1 + 1, x1, (y1 + 2) & 3);
        if (y1 == 0) {            // new strip: the two rows above are not in the ring
          x_load(xr, n1, y1 - 1, x1);
          x_store(xr, y1 - 1, x1, y1 & 3);
        }
      }
      __syncthreads();
      n = n1; y0 = y1; x0 = x1;
      advance(n1, y1, x1);
    }
  } else {
    // rp: ring slot of image row y0 - 1 of the current stage, which reads slots rp .. rp+3 while the two new rows of stage t+1
    // are stored to rp+4, rp+5 and the dy tile to the other buffer: one barrier per stage.  Registers: the values of stage t+1
    // were requested one whole stage ago; each item's registers are re-used for its stage t+2 request right after the
    // conversion.  Past the last stage the coordinates stay on the last one (a harmless re-fetch, stored but never read).
    int rp = 0, buf = 0;
    const int nlast = p.N - 1;
    auto advance_sat = [&](int& nn, int& yy, int& xx) {
      const int pn = nn, py = yy, px = xx;
      advance(nn, yy, xx);
      if (nn > nlast) { nn = pn; yy = py; xx = px; }
    };
    if (n1 > nlast) { n1 = n; y1 = y0; x1 = x0; }
    int n2 = n1, y2 = y1, x2 = x1;
    advance_sat(n2, y2, x2);
    if (t_begin < t_end) {
      x_load(xr, n, y0 - 1, x0);
      dy_load(n, y0, x0);
      x_store(xr, y0 - 1, x0, rp);
      x_load(xr, n, y0 + 1, x0);
      dy_store(0, y0, x0);
      x_store(xr, y0 + 1, x0, rp + 2);
      dy_load(n1, y1, x1);
      x_load(xr, n1, y1 + 1, x1);
      __syncthreads();
    }
    for (long t = t_begin; t < t_end; ++t) {
      if (y0 >= 0) compute(rp, buf);          // (the pre-stage of a strip has no gradient rows)
#pragma unroll
      for (int k = 0; k < NDY; ++k) {
        dy_store_item(k, buf ^ 1, y1, x1);
        dy_load_item(k, n2, y2, x2);
      }
#pragma unroll
      for (int k = 0; k < NXI; ++k) {
        x_store_item(xr, k, y1 + 1, x1, rp + 4);
        x_load_item(xr, k, n2, y2 + 1, x2);
      }
      __syncthreads();
      rp = (rp + 2) & 7;
      buf ^= 1;
      n = n1; y0 = y1; x0 = x1;
      n1 = n2; y1 = y2; x1 = x2;
      advance_sat(n2, y2, x2);
    }
  }
  // PIPE: the KP K parts of a work-group are summed through LDS (fixed order, one round per part) before the store: one partial
  // per K slice instead of KP (the 32-cout layers wrote 1024 x 36 KB of partials per launch)
  constexpr bool RED = PIPE && (KP == 1 || 3 * NPAIR * 3072 <= NT * CIT * XCP / 2);      // (the scratch is the input ring; host: wgrad3_pipe_reduces)
  constexpr int KPP = RED ? 1 : KP;            // partials per work-group
  if constexpr (RED && KP > 1) {
    float* red = reinterpret_cast<float*>(&s_x[0][0]);       // the loop ended with a barrier: the ring is free
    const int slot = pair + NPAIR * kh;
#pragma unroll
    for (int k = 1; k < KP; ++k) {
      if (kp == k) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[slot * 3072 + (kw * 16 + r) * 64 + lane] = acc[kw][r];
      }
      __syncthreads();
      if (kp == 0) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[kw][r] += red[slot * 3072 + (kw * 16 + r) * 64 + lane];
      }
      if (k + 1 < KP) __syncthreads();
    }
    if (kp != 0) return;
  }
  if (NPAIR == 3 && !computing) return;
  // ---- partial store: part[((slice*KPP + kp)*9 + tap)*CoP*CiP + co*CiP + ci] ----
  const int ci = cit * CIT + wn * 32 + l31;
  const size_t plane = (size_t)p.CoP * p.CiP;
  float* pb = p.part + ((size_t)blockIdx.x * KPP + (RED ? 0 : kp)) * 9 * plane;
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * COT + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (co < p.CoP && ci < p.CiP) pb[(kh * 3 + kw) * plane + (size_t)co * p.CiP + ci] = HF ? acc[kw][r] * hinv : acc[kw][r];
    }
  }
}


// ------------------------------------------------------------------------------------------
// Weight gradient of the THIN 3x3 layers (Cout <= 16, Cin = 16 | 32: smp's decoder.blocks.4 at full resolution) with two fp16
// terms on v_mfma_f32_16x16x32_f16 (3 x 16-cycle MFMAs per 16 x 16 x 32 block; the fp32 kernel needs 8 x 32 cycles).
// GEMM view (per tap):  D[co][ci] = sum_px dy[co][px] * in[ci][px + d(tap)],  K step = 32 pixels of one image row
//   A (16 x 32): lane l -> dy[co = l&15][px = 8*(l>>4) .. +7]                  (one 16-byte LDS read per term)
//   B (32 x 16): lane l -> in[ci = l&15][px + kw + 8*(l>>4) .. +7]: rows are stored from image column x0 - 1, so kw = 0 is an aligned
//                16-byte read, kw = 1 a 16-bit funnel shift (v_alignbit) with the next dword, kw = 2 a register renaming
//   D: lane -> ci = l&15, co = 4*(l>>4) + r
// Work-group = 4 waves; stage = 4 image rows x 32 columns of one image, wave w owns row w (its own K part; the four are summed
// through LDS at the end).  Pipeline as in k_wgrad3_bx3<PIPE>: the input rows live in a 12-slot ring per channel (a stage reads
// six rows and the four new rows of the next stage are stored meanwhile), the dy tile is double-buffered: ONE barrier per stage,
// straight-line loop body (every strip has a pre-stage at y0 = -4 that only brings rows -3 .. 0 in), and each item's global
// load for stage t+2 is issued right behind its stage t+1 conversion -- a whole stage in flight with no second register set.
template <int CIN, bool BNB>
__global__ __launch_bounds__(256, CIN == 32 ? 2 : 3) void k_wgrad_thin_h(const WgradXP p) {
  constexpr int NCB = CIN / 16, SR = 4, RING = 12;
  // Operand pitches in 16-byte slots = 2 (mod 4).  A 16 x 16 x 32 operand read is lane -> (row l15, k group lg): slot = pitch * l15 + lg
  // (+ wave-uniform terms), and ds_read_b128 is serviced in the 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... over 64
  // banks = 16 slots (MI355X_MICROARCH.md, LDS): a group holds rows {0-3, 12-15} of one k group and rows {4-11} of the NEXT.  An odd
  // pitch (rounds 2-5: 17 and 61 slots, chosen for 8 contiguous lanes x 32 banks) always puts one row of each k group on a slot the
  // other uses (the row sets are complements, a translation cannot map one onto itself): every group took 2 LDS cycles instead of 1,
  // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.41-0.45.  With pitch = 2 (mod 4) one k group lands on the even and the other on the odd slots.
  // (CIN = 32: the dy pitch stays at 17 slots -- 2 of the 26 reads of a stage -- so that two work-groups still fit the CU's 160 KB.)
#ifndef SC_WTH_DYPAD
#define SC_WTH_DYPAD(CIN) ((CIN) == 16 ? 16 : 8)
#endif
#ifndef SC_WTH_XPAD
#define SC_WTH_XPAD 16
#endif
  constexpr int DYP = SR * 32 + SC_WTH_DYPAD(CIN);   // dy pitch per cout in halves (288 B = 18 slots; CIN = 32: 272 B)
  constexpr int XP = 40, XCP = RING * XP + SC_WTH_XPAD;  // input pitch per row (34 used) / per channel in halves (80 B / 992 B = 62 slots)
  constexpr int NDY = (16 * SR * 16) / 256;    // dy pixel pairs per thread per stage (exact)
  constexpr int XCNT = CIN * SR * 17;          // input pixel pairs per stage
  constexpr int NXI = (XCNT + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned s_dy[2][2][16 * DYP / 2];      // [buffer][term]
  __shared__ __attribute__((aligned(16))) unsigned s_x[2][CIN * XCP / 2];
  __shared__ __attribute__((aligned(16))) float s_ca[16 * SC_CST];
  __shared__ __attribute__((aligned(16))) float s_cb[CIN * 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int H = p.H, W = p.W;
  const float hsg = h_grad_scale(p.absmax);
  const float hsa = h_act_scale(p.xb0, p.xb1);
  const float hinv = 1.f / (hsg * hsa);

  for (int i = tid; i < 16 * SC_CST; i += 256) {
    const int ch = i / SC_CST, f = i % SC_CST;          // (operand scales folded into the constants, see split2h)
    const float fs = (BNB ? (f >= 2 && f <= 4) : (f < 2)) ? hsg : 1.f;
    s_ca[i] = fs * ((p.dy.cst && p.dy.mode != SC_SRC_RAW && ch < p.Cout) ? p.dy.cst[(size_t)ch * SC_CST + f] : (f == 0 ? 1.f : 0.f));
  }
  for (int i = tid; i < CIN; i += 256) {
    float sc = 1.f, sh = 0.f;
    if (p.s0.cst && p.s0.mode != SC_SRC_RAW) { sc = p.s0.cst[(size_t)i * SC_CST]; sh = p.s0.cst[(size_t)i * SC_CST + 1]; }
    s_cb[i * 4] = sc * hsa; s_cb[i * 4 + 1] = sh * hsa;
    s_cb[i * 4 + 2] = h_lo(sc_act_lo(p.s0.act), hsa); s_cb[i * 4 + 3] = h_hi(sc_act_hi(p.s0.act), hsa);
  }

  floatx4 acc[9][NCB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[t][cb] = (floatx4){0.f, 0.f, 0.f, 0.f};

  const int tiles_x = (W + 31) >> 5, RS = (H + SR - 1) / SR, RSS = RS + 1;
  const long T = (long)p.N * tiles_x * RSS;
  const long t_begin = T * blockIdx.x / p.nsl, t_end = T * (blockIdx.x + 1) / p.nsl;
  const float dlo = sc_act_lo(p.dy.act) * (BNB ? 1.f : hsg), dhi = sc_act_hi(p.dy.act) * (BNB ? 1.f : hsg);
  const size_t HW = (size_t)H * W;
  const int up = p.s0.up, Ws = W >> up;
  const unsigned istr = (unsigned)((size_t)CIN * (H >> up) * Ws * 4);      // bytes per image of the input (host-checked < 2^32 / N)

  // per-item address invariants (bytes within one image)
  unsigned dyo[NDY], xo[NXI];
#pragma unroll
  for (int k = 0; k < NDY; ++k) {
    const int co = (tid + 256 * k) >> 6;
    dyo[k] = (unsigned)(co < p.Cout ? co : 0) * (unsigned)HW * 4u;
  }
#pragma unroll
  for (int k = 0; k < NXI; ++k) {
    int it = tid + 256 * k;
    if (it >= XCNT) it -= XCNT;
    xo[k] = (unsigned)((it / 17) >> 2) * (unsigned)((H >> up) * Ws) * 4u;
  }

  float dg[NDY][2], dv[BNB ? NDY : 1][2], xr[NXI][2];
  // dy item: co = it >> 6, row = (it >> 4) & 3, columns 2 * (it & 15), +1: one 8-byte load per tensor (W even, host-checked)
  auto dy_load_item = [&](int k, int n, int y0, int x0) {
    const int it = tid + 256 * k;
    const int row = (it >> 4) & 3, col = 2 * (it & 15);
    const int y = y0 + row, x = x0 + col;
    const size_t io = (size_t)n * p.Cout * HW * 4;
    const unsigned off = dyo[k] + (unsigned)(((y >= 0 && y < H) ? y : 0) * W + ((x < W) ? x : 0)) * 4u;
    const float2 g = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p.dy.x) + io + off);
    dg[k][0] = g.x; dg[k][1] = g.y;
    if (BNB) {
      const float2 a = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p.dy.aux) + io + off);
      dv[k][0] = a.x; dv[k][1] = a.y;
    }
  };
  auto dy_store_item = [&](int k, int buf, int y0, int x0) {
    const int it = tid + 256 * k;
    const int co = it >> 6, row = (it >> 4) & 3, col = 2 * (it & 15);
    const int y = y0 + row, x = x0 + col;
    const bool ok = (co < p.Cout) && y >= 0 && y < H && x < W;
    const float4 c0 = *reinterpret_cast<const float4*>(&s_ca[co * SC_CST]);
    const float c4 = s_ca[co * SC_CST + 4];
    float v0, v1;
    if (BNB) {
      v0 = sc_pro_bnbwd(dg[k][0], dv[k][0], c0.x, c0.y, c0.z, c0.w, c4, dlo, dhi);
      v1 = sc_pro_bnbwd(dg[k][1], dv[k][1], c0.x, c0.y, c0.z, c0.w, c4, dlo, dhi);
    } else {
      v0 = sc_pro_affine(dg[k][0], c0.x, c0.y, dlo, dhi);
      v1 = sc_pro_affine(dg[k][1], c0.x, c0.y, dlo, dhi);
    }
    v0 = ok ? v0 : 0.f;
    v1 = ok ? v1 : 0.f;
    unsigned t0, t1;
    split2h(v0, v1, t0, t1);
    const int d = (co * DYP + row * 32 + col) >> 1;
    s_dy[buf][0][d] = t0; s_dy[buf][1][d] = t1;
  };
  // input item: pair pr in 0..16 (columns x0 - 1 + 2 pr, +1), row R + rowi, channel cil; sl0 = ring slot of row R
  auto x_load_item = [&](int k, int n, int R, int x0) {
    int it = tid + 256 * k;
    if (it >= XCNT) it -= XCNT;
    const int rc = it / 17, pr = it - rc * 17;
    const int rowi = rc & 3;
    const int y = R + rowi, x = x0 - 1 + 2 * pr;
    const int yc = (y >= 0 && y < H) ? y : 0;
    const int xa = (x >= 0 && x < W) ? x : 0, xb = (x + 1 < W) ? x + 1 : 0;
    const char* base = reinterpret_cast<const char*>(p.s0.x) + (size_t)n * istr;
    const unsigned ro = xo[k] + (unsigned)((yc >> up) * Ws) * 4u;
    xr[k][0] = *reinterpret_cast<const float*>(base + (ro + (unsigned)(xa >> up) * 4u));
    xr[k][1] = *reinterpret_cast<const float*>(base + (ro + (unsigned)(xb >> up) * 4u));
  };
  auto x_store_item = [&](int k, int R, int x0, int sl0) {
    int it = tid + 256 * k;
    if (it >= XCNT) it -= XCNT;
    const int rc = it / 17, pr = it - rc * 17;
    const int rowi = rc & 3, cil = rc >> 2;
    const int y = R + rowi, x = x0 - 1 + 2 * pr;
    const bool oky = (y >= 0) && (y < H);
    const float4 c = *reinterpret_cast<const float4*>(&s_cb[cil * 4]);
    float v0 = sc_pro_affine_h(xr[k][0], c.x, c.y, c.z, c.w);
    float v1 = sc_pro_affine_h(xr[k][1], c.x, c.y, c.z, c.w);
    v0 = (oky && x >= 0 && x < W) ? v0 : 0.f;
    v1 = (oky && x + 1 < W) ? v1 : 0.f;
    unsigned t0, t1;
    split2h<false>(v0, v1, t0, t1);
    int slot = sl0 + rowi;
    slot = slot >= RING ? slot - RING : slot;
    const int d = ((cil * XCP + slot * XP) >> 1) + pr;
    s_x[0][d] = t0; s_x[1][d] = t1;
  };
  auto ring = [](int v) { return v >= RING ? v - RING : v; };
  const unsigned opaque_zero = (unsigned)p.H >> 30;      // 0, unknown to the compiler
  // MFMAs of one stage: wave w = image row y0 + w; rp = ring slot of row y0 - 1
  auto compute = [&](int rp, int buf) {
    const int r = wave;
    halfx8 A[2];
    const int da = (l15 * DYP + r * 32 + 8 * lg) >> 1;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) A[tm] = __builtin_bit_cast(halfx8, *reinterpret_cast<const uintx4*>(&s_dy[buf][tm][da]));
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int slot = ring(rp + r + kh);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int dx = (((cb * 16 + l15) * XCP + slot * XP) >> 1) + 4 * lg;
        halfx8 B[3][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const uintx4 X0 = *reinterpret_cast<const uintx4*>(&s_x[tm][dx]);
          const uintx4 X1v = *reinterpret_cast<const uintx4*>(&s_x[tm][dx + 4]);      // (kept a 16-byte read: see k_wgrad3_bx3)
          const unsigned X1 = X1v[0] | (X1v[1] & opaque_zero);
          uintx4 S1, S2;
          S1[0] = __builtin_amdgcn_alignbit(X0[1], X0[0], 16);
          S1[1] = __builtin_amdgcn_alignbit(X0[2], X0[1], 16);
          S1[2] = __builtin_amdgcn_alignbit(X0[3], X0[2], 16);
          S1[3] = __builtin_amdgcn_alignbit(X1, X0[3], 16);
          S2[0] = X0[1]; S2[1] = X0[2]; S2[2] = X0[3]; S2[3] = X1;
          B[0][tm] = __builtin_bit_cast(halfx8, X0);
          B[1][tm] = __builtin_bit_cast(halfx8, S1);
          B[2][tm] = __builtin_bit_cast(halfx8, S2);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          floatx4 c = acc[kh * 3 + kw][cb];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0], B[kw][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1], B[kw][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0], B[kw][0], c, 0, 0, 0);
          acc[kh * 3 + kw][cb] = c;
        }
      }
    }
  };

  // stage coordinates (n, y0, x0) of t, t+1, t+2; past the last stage they stay on the last one (harmless re-fetch)
  const int nlast = p.N - 1;
  auto advance = [&](int& nn, int& yy, int& xx) {
    const int pn = nn, py = yy, px = xx;
    yy += SR;
    if (yy >= SR * RS) { yy = -SR; xx += 32; if (xx >= 32 * tiles_x) { xx = 0; ++nn; } }
    if (nn > nlast) { nn = pn; yy = py; xx = px; }
  };
  int n = 0, y0 = 0, x0 = 0;
  if (t_begin < t_end) {
    const int strip = (int)(t_begin / RSS), ty = (int)(t_begin - (long)strip * RSS);
    n = strip / tiles_x; x0 = (strip - n * tiles_x) * 32; y0 = (ty - 1) * SR;
  }
  n = __builtin_amdgcn_readfirstlane(n); y0 = __builtin_amdgcn_readfirstlane(y0); x0 = __builtin_amdgcn_readfirstlane(x0);   // uniform: scalar address arithmetic
  int n1 = n, y1 = y0, x1 = x0;
  advance(n1, y1, x1);
  int n2 = n1, y2 = y1, x2 = x1;
  advance(n2, y2, x2);
  __syncthreads();          // constants in LDS
  int rp = 0, buf = 0;
  if (t_begin < t_end) {
    // the first stage's six rows (two batches of four: rows y0 - 1 .. y0 + 6, the last two are the next stage's first two) and dy
#pragma unroll
    for (int k = 0; k < NXI; ++k) x_load_item(k, n, y0 - 1, x0);
#pragma unroll
    for (int k = 0; k < NDY; ++k) dy_load_item(k, n, y0, x0);
#pragma unroll
    for (int k = 0; k < NXI; ++k) { x_store_item(k, y0 - 1, x0, rp); x_load_item(k, n, y0 + 3, x0); }
#pragma unroll
    for (int k = 0; k < NDY; ++k) { dy_store_item(k, 0, y0, x0); dy_load_item(k, n1, y1, x1); }
#pragma unroll
    for (int k = 0; k < NXI; ++k) { x_store_item(k, y0 + 3, x0, rp + 4); x_load_item(k, n1, y1 + 1, x1); }
    __syncthreads();
  }
  for (long t = t_begin; t < t_end; ++t) {
    if (y0 >= 0) compute(rp, buf);            // (the pre-stage of a strip has no gradient rows)
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
      dy_store_item(k, buf ^ 1, y1, x1);
      dy_load_item(k, n2, y2, x2);
    }
    const int sl1 = ring(rp + 6);             // rows y1 + 1 .. y1 + 4 = y0 + 5 .. y0 + 8
#pragma unroll
    for (int k = 0; k < NXI; ++k) {
      x_store_item(k, y1 + 1, x1, sl1);
      x_load_item(k, n2, y2 + 1, x2);
    }
    __syncthreads();
    rp = ring(rp + 4);
    buf ^= 1;
    n = n1; y0 = y1; x0 = x1;
    n1 = n2; y1 = y2; x1 = x2;
    advance(n2, y2, x2);
  }
  // ---- the four waves' partial sums through LDS (two rounds), then part[slice][tap][co][ci]   (CoP = 16, CiP = CIN)
  float* red = reinterpret_cast<float*>(&s_x[0][0]);        // 2 x 9 x NCB x 256 floats = 18 / 36 KB of the 31 / 62 KB ring
  auto put = [&](int slot) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        *reinterpret_cast<floatx4*>(&red[(((slot * 9 + tap) * NCB + cb) * 64 + lane) * 4]) = acc[tap][cb];
  };
  auto take = [&](int slot) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[tap][cb] += *reinterpret_cast<const floatx4*>(&red[(((slot * 9 + tap) * NCB + cb) * 64 + lane) * 4]);
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) take(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave == 0) {
    take(0);
    const size_t plane = (size_t)16 * CIN;
    float* pb = p.part + (size_t)blockIdx.x * 9 * plane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[tap * plane + (size_t)(4 * lg + r) * CIN + cb * 16 + l15] = acc[tap][cb][r] * hinv;
  }
}

struct WgradXPlan { int wm, nci, kp, nsl, CoP, CiP, co_tiles, ci_tiles; };
WgradXPlan plan_wgrad_bx3(int N, int H, int W, int Cout, int Cin, bool allow96 = false) {
  WgradXPlan pl;
  pl.wm = Cout > 32 ? 2 : 1;
  // 32-wide cin tiles only for Cin <= 32: for 80 or 152 input channels the emptier 64-wide tiles still win (measured
  // 0.59 vs 0.66 ms and 0.38 vs 0.39 ms): twice the MFMA work per staged dy tile outweighs the padding
  pl.nci = Cin <= 32 ? 1 : 2;
  // <= 32 output and 65..96 input channels (two-fp16-term kernels): ONE 96-wide tile instead of a full and a mostly empty 64-wide one,
  // each of which stages dy
  if (allow96 && pl.wm == 1 && Cin > 64 && Cin <= 96) pl.nci = 3;
  // (measured, not kept: 33..64 output x 129..192 input channels as 32 x 96 tiles -- two passes over dy instead of three, but the input
  // rows staged twice: decoder.blocks.2.conv1 214 -> 282 us)
  pl.kp = pl.nci * pl.wm == 3 ? 1 : 4 / (pl.nci * pl.wm);
  pl.CoP = (Cout + 31) / 32 * 32;
  pl.CiP = (Cin + 31) / 32 * 32;
  pl.co_tiles = (Cout + 32 * pl.wm - 1) / (32 * pl.wm);
  pl.ci_tiles = (Cin + 32 * pl.nci - 1) / (32 * pl.nci);
  const long T = (long)N * ((W + 31) / 32) * ((H + 1) / 2);
  // K slices: one work-group per CU is resident (95 KB of LDS), so the launch runs in rounds of 256 work-groups.  Pick the
  // slice count that minimises  rounds x (stages per slice + fixed prologue/epilogue), e.g.
  // 16 tiles -> 16 slices (one full round) rather than 32; 88 tiles -> 5 slices (1.7 rounds) rather than 3 (1.03 rounds).
  const long tiles = (long)pl.co_tiles * pl.ci_tiles;
  const long nmax = T / 4 < 512 ? (T / 4 > 1 ? T / 4 : 1) : 512;
  auto cost = [&](long k) { return (double)((tiles * k + 255) / 256) * ((double)((T + k - 1) / k) + 6.0); };
  double best_cost = 1e30;
  for (long k = 1; k <= nmax; ++k) best_cost = cost(k) < best_cost ? cost(k) : best_cost;
  long best = 1;
  for (long k = 1; k <= nmax; ++k)
    if (cost(k) <= 1.03 * best_cost) { best = k; break; }        // fewest slices (least partial-sum traffic) within 3 %
  long want = best;
  pl.nsl = (int)want;
  return pl;
}


// ------------------------------------------------------------------------------------------
// all filter packs of the network in ONE launch (88 tiny launches per step otherwise): block -> descriptor by binary search
// over the block offsets, then the same index arithmetic as k_pack_weights (conv_mfma.hip) / k_pack_weights_bx3
struct PackDesc {
  const float* w; float* wpk;
  int Cout, Cin, ks, co_t, tflip, bx3;
  unsigned long long total;      // work items (fp32 layout: destination floats; bx3: destination bf16 per term)
};
static_assert(sizeof(PackDesc) == sizeof(sc_pack_desc), "sc_pack_desc layout");

__global__ __launch_bounds__(256) void k_pack_batch(const PackDesc* __restrict__ descs, const unsigned* __restrict__ starts, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {                      // last descriptor whose first block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = descs[lo];
  const size_t i = (size_t)(blockIdx.x - starts[lo]) * 256 + threadIdx.x;
  if (i >= d.total) return;
  if (d.bx3 == SC_PACK_THIN16) { pack_thin_item(d.w, reinterpret_cast<unsigned short*>(d.wpk), i, d.Cout, d.Cin, d.tflip); return; }
  if (d.bx3 == SC_PACK_SPD) { spd_pack_item(d.w, reinterpret_cast<unsigned short*>(d.wpk), i, d.Cout, d.Cin, d.co_t, (d.tflip & 3) == 2, (d.tflip & 4) != 0, (d.tflip & 3) == 3 ? d.Cin - d.co_t : 0); return; }      // tflip 2: + virtual skip channels; 3: + skip tiles; | 4: one bf16 term
  if (d.bx3 == SC_PACK_SP) { sp_pack_item(d.w, reinterpret_cast<unsigned short*>(d.wpk), i, d.Cout, d.co_t, d.Cin - d.co_t, (d.tflip & 4) != 0); return; }      // conv_sp.hip: co_t = up-sampled channels
  const int M = d.tflip ? d.Cin : d.Cout, K = d.tflip ? d.Cout : d.Cin;
  if (d.bx3 == SC_PACK_PW3) {
    // pointwise filters for k_pw3 (conv_pw3.hip): [cout block][k step][term][lane][8] bf16, lane -> cout l&31, k = 16*step + 8*(l>>5) + j;
    // three exact bf16 terms; cout blocks zero-padded to a multiple of 4
    const int nks = (K + 15) / 16;
    const int j = (int)(i % 8), lane = (int)((i / 8) % 64);
    const int ks = (int)((i / 512) % nks), cb = (int)(i / ((size_t)512 * nks));
    const int m = cb * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (m < M && k < K) v = d.tflip ? d.w[(size_t)k * M + m] : d.w[(size_t)m * K + k];
    unsigned short t[3];
    split_filter(v, false, t);
    unsigned short* out = reinterpret_cast<unsigned short*>(d.wpk);
    for (int c = 0; c < 3; ++c) out[((((size_t)cb * nks + ks) * 3 + c) * 64 + lane) * 8 + j] = t[c];
    return;
  }
  const int taps = d.ks * d.ks;
  if (!d.bx3) {
    const int kc = d.ks == 3 ? 8 : 16;
    const int Kpad = (K + kc - 1) / kc * kc;
    const int col = (int)(i % d.co_t);
    size_t r = i / d.co_t;
    const int tap = (int)(r % taps); r /= taps;
    const int k = (int)(r % Kpad);
    const int m = (int)(r / Kpad) * d.co_t + col;
    float v = 0.f;
    if (m < M && k < K) v = d.tflip ? d.w[((size_t)k * M + m) * taps + (taps - 1 - tap)] : d.w[((size_t)m * K + k) * taps + tap];
    d.wpk[i] = v;
    return;
  }
  const int nchunk = (K + 15) / 16;
  size_t r = i;
  const int j = (int)(r % 8); r /= 8;
  const int col = (int)(r % d.co_t); r /= d.co_t;
  const int half = (int)(r % 2); r /= 2;
  const int kw = (int)(r % 3); r /= 3;
  const int kh = (int)(r % 3); r /= 3;
  const int chunk = (int)(r % nchunk);
  const int mt = (int)(r / nchunk);
  const int m = mt * d.co_t + col, k = chunk * 16 + half * 8 + j, tap = kh * 3 + kw;
  float v = 0.f;
  if (m < M && k < K) v = d.tflip ? d.w[((size_t)k * M + m) * 9 + (8 - tap)] : d.w[((size_t)m * K + k) * 9 + tap];
  unsigned short t[3];
  const bool hmode = d.bx3 == SC_TERMS_F16X2;
  const int nt = hmode ? 2 : d.bx3;
  split_filter(v, hmode, t);
  const size_t stage = ((size_t)mt * nchunk + chunk) * 3 + kh;
  unsigned short* out = reinterpret_cast<unsigned short*>(d.wpk);
  for (int c = 0; c < nt; ++c) out[((((stage * nt + c) * 3 + kw) * 2 + half) * d.co_t + col) * 8 + j] = t[c];
}

}  // namespace

extern "C" size_t sc_packed_weight_floats_bx3(int Cout, int Cin, int co_t, int transpose_flip, int terms) {
  const int M = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  const size_t mt = (M + co_t - 1) / co_t, nchunk = (K + 15) / 16;
  return mt * nchunk * 3 * 6 * (size_t)(terms == SC_TERMS_F16X2 ? 2 : (terms >= 1 && terms <= 3 ? terms : 3)) * co_t * 4;     // 16-byte entries -> floats
}

extern "C" int sc_pack_weights_bx3(const float* w, float* wpk, int Cout, int Cin, int co_t, int transpose_flip,
                                   int terms, sc_stream stream) {
  SC_REQUIRE(terms >= 1 && terms <= 4, "sc_pack_weights_bx3: terms must be 1, 2, 3 or SC_TERMS_F16X2 (got %d)", terms);
  SC_REQUIRE(w && wpk && Cout > 0 && Cin > 0, "sc_pack_weights_bx3: bad argument");
  SC_REQUIRE(co_t == 32 || co_t == 64, "sc_pack_weights_bx3: co_t must be 32 or 64 (got %d)", co_t);
  SC_REQUIRE(((uintptr_t)wpk & 15) == 0, "sc_pack_weights_bx3: destination must be 16-byte aligned");
  const int M = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  const int nchunk = (K + 15) / 16;
  const size_t total = (size_t)((M + co_t - 1) / co_t) * nchunk * 9 * 2 * co_t * 8;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(k_pack_weights_bx3, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<unsigned short*>(wpk), Cout, Cin, co_t, transpose_flip, nchunk,
                     terms == SC_TERMS_F16X2 ? 2 : terms, terms == SC_TERMS_F16X2 ? 1 : 0, total);
  SC_LAUNCH_OK("sc_pack_weights_bx3");
  return SC_OK;
}

extern "C" int sc_conv3x3_bx3(const sc_conv_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_conv3x3_bx3: null args");
  SC_REQUIRE(a->ks == 3, "sc_conv3x3_bx3: ks must be 3 (got %d)", a->ks);
  SC_REQUIRE(a->co_t == 32 || a->co_t == 64, "sc_conv3x3_bx3: co_t must be 32 or 64 (got %d)", a->co_t);
  SC_REQUIRE(a->nsrc == 1 || a->nsrc == 2, "sc_conv3x3_bx3: nsrc must be 1 or 2");
  const int C0 = a->src[0].C, C1 = a->nsrc == 2 ? a->src[1].C : 0;
  SC_REQUIRE(C0 > 0 && C1 >= 0 && (a->nsrc == 1 || (C0 % 16 == 0 && C1 > 0)),
             "sc_conv3x3_bx3: a concat needs the first source's channels to be a multiple of 16 (got %d,%d)", C0, C1);
  SC_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cout > 0, "sc_conv3x3_bx3: bad shape");
  SC_REQUIRE(a->csplit > 0 && a->csplit <= a->Cout, "sc_conv3x3_bx3: bad csplit");
  SC_REQUIRE(a->csplit == a->Cout || (a->add0 == nullptr && a->add1 == nullptr), "sc_conv3x3_bx3: add tensors need a single output");
  SC_REQUIRE(((uintptr_t)a->wpk & 15) == 0, "sc_conv3x3_bx3: packed filters must be 16-byte aligned");
  for (int s = 0; s < a->nsrc; ++s) {
    SC_REQUIRE(a->src[s].up == 0 || (a->H % 2 == 0 && a->W % 2 == 0), "sc_conv3x3_bx3: upsampled source needs even H,W");
    SC_REQUIRE(a->src[s].up == 0 || a->src[s].up == 1, "sc_conv3x3_bx3: up must be 0 or 1");
    SC_REQUIRE(a->src[s].mode == SC_SRC_RAW || a->src[s].cst != nullptr, "sc_conv3x3_bx3: source %d needs constants", s);
    SC_REQUIRE(a->src[s].mode != SC_SRC_NORM, "sc_conv3x3_bx3: NORM sources are the stem's");
    SC_REQUIRE(a->src[s].mode != SC_SRC_BNBWD || a->src[s].aux != nullptr, "sc_conv3x3_bx3: BNBWD source needs aux");
    SC_REQUIRE(a->src[s].mode != SC_SRC_BNBWD || a->nsrc == 1, "sc_conv3x3_bx3: a BNBWD source cannot be part of a concat");
  }
  ConvXP p;
  p.s0 = to_srcd(a->src[0]);
  p.s1 = a->nsrc == 2 ? to_srcd(a->src[1]) : empty_srcd();
  p.wpk = reinterpret_cast<const uintx4*>(a->wpk); p.N = a->N; p.H = a->H; p.W = a->W; p.Cout = a->Cout;
  p.out0 = a->out0; p.out1 = a->out1; p.csplit = a->csplit; p.accum0 = a->accum0; p.accum1 = a->accum1;
  p.add0 = a->add0; p.add1 = a->add1; p.stats = a->stats;
  p.down0 = a->down0 ? 1 : 0;
  SC_REQUIRE(!p.down0 || (a->H % 2 == 0 && a->W % 2 == 0 && a->stats == nullptr && a->add0 == nullptr && a->add1 == nullptr),
             "sc_conv3x3_bx3: down0 (2x2-summed half-resolution out0) needs even H, W and no stats/add tensors");
  const int co_tiles = (a->Cout + a->co_t - 1) / a->co_t;
  dim3 grid(((a->W + 31) / 32) * ((a->H + 7) / 8), co_tiles, a->N);
  constexpr int xcdmap_env = 2;   // XCD-aware 1-D numbering (0 = plain 3-D grid: 2.7x instead of 1.24x the algorithmic bytes)
  p.xcdmap = xcdmap_env >= 2 ? 2 : ((xcdmap_env && co_tiles > 1) ? 1 : 0);
  {
    // cout-major numbering (3) for launches whose packed filters exceed this many bytes.  Off by default: measured on
    // decoder.blocks.0 (12.7 MB of filters) it is level on the forward (292 vs 291 us) and slower on the backward-data launch
    // (392 vs 368 us) -- the Infinity Cache absorbs the filter re-reads, the patches are what an XCD should share
    constexpr long big_env = 0L;      // cout-major numbering above this filter size: measured +6 % slower backward-data -> off
    const long nkc = (C0 + C1 + 15) / 16;
    const long wbytes = (long)co_tiles * nkc * 3 * 6 * (a->terms == SC_TERMS_F16X2 || a->terms == 2 ? 2 : (a->terms == 1 ? 1 : 3)) * a->co_t * 16;
    if (xcdmap_env >= 2 && big_env > 0 && wbytes > big_env && (co_tiles >= 8 || co_tiles == 1 || co_tiles == 2 || co_tiles == 4)) p.xcdmap = 3;
  }
  if (p.xcdmap == 3) {
    const long total = (long)grid.x * a->N;
    const long slots = co_tiles >= 8 ? (long)((co_tiles + 7) / 8) * total : (total + (8 / co_tiles) - 1) / (8 / co_tiles);
    SC_REQUIRE(slots * 8 < (1L << 31), "sc_conv3x3_bx3: grid too large");
    grid = dim3((unsigned)(slots * 8));
  } else if (p.xcdmap) {
    const long pt8 = ((long)grid.x * a->N + 7) / 8 * 8;
    SC_REQUIRE(pt8 * co_tiles < (1L << 31), "sc_conv3x3_bx3: grid too large");
    grid = dim3((unsigned)(pt8 * co_tiles));
  }
  hipStream_t st = (hipStream_t)stream;
  const bool bnb = a->src[0].mode == SC_SRC_BNBWD;
  SC_REQUIRE(a->terms >= 0 && a->terms <= 4, "sc_conv3x3_bx3: terms must be 0 (= 3), 1, 2, 3 or SC_TERMS_F16X2 (got %d)", a->terms);
  p.absmax = a->absmax; p.xb0 = a->xbound[0]; p.xb1 = a->xbound[1];
  set_bnr(p, a->bnr);
  if (a->bnr) {
    SC_REQUIRE(bnb && a->bnr->y && a->bnr->cst && a->bnr->rows, "sc_conv3x3_bx3: bnr needs a BNBWD source and y / cst / rows");
    SC_REQUIRE(!a->accum0 && !a->add0 && !a->add1 && !a->stats, "sc_conv3x3_bx3: bnr: out0 must receive the complete gradient (no accum0 / add / stats)");
    SC_REQUIRE(a->csplit == a->Cout || a->csplit % a->co_t == 0, "sc_conv3x3_bx3: bnr: csplit must fall on a cout-tile boundary");
  }
#define SC_LAUNCH_BX3(NT, HF)                                                                                  \
  do {                                                                                                         \
    if (a->co_t == 64 && bnb) hipLaunchKernelGGL((k_conv3_bx3<2, true, NT, HF>), grid, dim3(256), 0, st, p);   \
    else if (a->co_t == 64) hipLaunchKernelGGL((k_conv3_bx3<2, false, NT, HF>), grid, dim3(256), 0, st, p);    \
    else if (bnb) hipLaunchKernelGGL((k_conv3_bx3<1, true, NT, HF>), grid, dim3(256), 0, st, p);               \
    else hipLaunchKernelGGL((k_conv3_bx3<1, false, NT, HF>), grid, dim3(256), 0, st, p);                       \
  } while (0)
  constexpr int ws_env = 1;
  static const int ws_env_min = [] { const char* e = getenv("STARCOP_BX3_WS_MINCHUNKS"); return e ? atoi(e) : 16; }();
  if (a->terms == 1) SC_LAUNCH_BX3(1, false); else if (a->terms == 2) SC_LAUNCH_BX3(2, false);
  // the wave-specialised kernel pays for its 8-wave work-groups (prologue / epilogue of only two per CU) on short K loops:
  // measured faster from 16 chunks of 16 input channels up (decoder.blocks.0), level at 8-10, slower below
  else if (a->terms == SC_TERMS_F16X2 && ws_env && (C0 + C1 + 15) / 16 >= ws_env_min && (!bnb || a->src[0].C <= 256) && !a->bnr) {
    if (a->co_t == 64 && bnb) hipLaunchKernelGGL((k_conv3_ws<2, true>), grid, dim3(512), 0, st, p);
    else if (a->co_t == 64) hipLaunchKernelGGL((k_conv3_ws<2, false>), grid, dim3(512), 0, st, p);
    else if (bnb) hipLaunchKernelGGL((k_conv3_ws<1, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((k_conv3_ws<1, false>), grid, dim3(512), 0, st, p);
  }
  else if (a->terms == SC_TERMS_F16X2) SC_LAUNCH_BX3(2, true); else SC_LAUNCH_BX3(3, false);
#undef SC_LAUNCH_BX3
  SC_LAUNCH_OK("sc_conv3x3_bx3");
  return SC_OK;
}

// does the pipelined kernel of this plan sum its K parts itself? (mirrors the kernel's RED: 8-slot ring of two fp16 terms as scratch)
static bool wgrad3_pipe_reduces(const WgradXPlan& pl) {
  const int npair = pl.nci * pl.wm, cit = 32 * pl.nci, xcp = 8 * 40 + 8;
  return pl.kp == 1 || 3 * npair * 3072 <= 2 * cit * xcp / 2;
}

static size_t wgrad_bx3_workspace_of(const WgradXPlan& pl) {
  const size_t E = (size_t)9 * pl.CoP * pl.CiP;
  const int nparts = pl.nsl * pl.kp;
  return (size_t)nparts * E + sc_reduce_scratch_floats(nparts, E);
}
extern "C" size_t sc_wgrad_bx3_workspace_floats(int N, int H, int W, int Cout, int Cin) {      // (either tiling: `terms` picks it at launch)
  const size_t a = wgrad_bx3_workspace_of(plan_wgrad_bx3(N, H, W, Cout, Cin, false));
  const size_t b = wgrad_bx3_workspace_of(plan_wgrad_bx3(N, H, W, Cout, Cin, true));
  return a > b ? a : b;
}

extern "C" int sc_conv3x3_wgrad_bx3(const sc_wgrad_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_conv3x3_wgrad_bx3: null args");
  SC_REQUIRE(a->ks == 3, "sc_conv3x3_wgrad_bx3: ks must be 3");
  SC_REQUIRE(a->nsrc == 1 || a->nsrc == 2, "sc_conv3x3_wgrad_bx3: nsrc must be 1 or 2");
  const int C0 = a->src[0].C, C1 = a->nsrc == 2 ? a->src[1].C : 0;
  SC_REQUIRE(C0 + C1 == a->Cin, "sc_conv3x3_wgrad_bx3: source channels (%d+%d) != Cin %d", C0, C1, a->Cin);
  SC_REQUIRE(a->dy.C == a->Cout, "sc_conv3x3_wgrad_bx3: dy channels %d != Cout %d", a->dy.C, a->Cout);
  SC_REQUIRE(a->dy.up == 0 && a->dy.mode != SC_SRC_NORM, "sc_conv3x3_wgrad_bx3: unsupported dy source");
  SC_REQUIRE(a->dy.mode != SC_SRC_BNBWD || a->dy.aux != nullptr, "sc_conv3x3_wgrad_bx3: BNBWD dy needs aux");
  SC_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cout > 0 && a->Cin > 0, "sc_conv3x3_wgrad_bx3: bad shape");
  for (int s = 0; s < a->nsrc; ++s) {
    SC_REQUIRE(a->src[s].mode == SC_SRC_RAW || a->src[s].mode == SC_SRC_AFFINE, "sc_conv3x3_wgrad_bx3: input sources must be RAW or AFFINE");
    SC_REQUIRE(a->src[s].up == 0 || (a->src[s].up == 1 && a->H % 2 == 0 && a->W % 2 == 0), "sc_conv3x3_wgrad_bx3: upsampled source needs even H,W");
  }
  static const bool no96 = [] { const char* e = getenv("STARCOP_WGRAD96"); return e && atoi(e) == 0; }();      // (same-box A/B)
  constexpr int pipe_env = 1;      // one-barrier refill pipeline (0 = the two-barrier stages it replaced: 1.89 vs 1.63 ms per step)
  // the pipelined variant: BatchNorm-backward gradients, 8-byte gradient loads (even W), 32-bit byte offsets within a tensor
  bool pipe = pipe_env && a->terms == SC_TERMS_F16X2 && a->dy.mode == SC_SRC_BNBWD && a->W % 2 == 0 && (((uintptr_t)a->dy.x | (uintptr_t)a->dy.aux) & 7) == 0 &&
              (size_t)a->Cout * a->H * a->W * 4 < (1ull << 32);
  for (int s = 0; s < a->nsrc; ++s) pipe = pipe && (size_t)a->N * a->src[s].C * (a->H >> a->src[s].up) * (a->W >> a->src[s].up) * 4 < (1ull << 32);
  const WgradXPlan pl = plan_wgrad_bx3(a->N, a->H, a->W, a->Cout, a->Cin, pipe && !no96);      // (the 96-wide tile: pipelined kernel only)
  const size_t need = sc_wgrad_bx3_workspace_floats(a->N, a->H, a->W, a->Cout, a->Cin);
  SC_REQUIRE(a->part_floats >= need, "sc_conv3x3_wgrad_bx3: workspace too small (%zu < %zu floats)", a->part_floats, need);
  WgradXP p;
  p.dy = to_srcd(a->dy); p.s0 = to_srcd(a->src[0]); p.s1 = a->nsrc == 2 ? to_srcd(a->src[1]) : empty_srcd();
  p.N = a->N; p.H = a->H; p.W = a->W; p.Cout = a->Cout; p.Cin = a->Cin; p.part = a->part;
  p.nsl = pl.nsl; p.CoP = pl.CoP; p.CiP = pl.CiP;
  dim3 grid(pl.nsl, pl.ci_tiles, pl.co_tiles);
  hipStream_t st = (hipStream_t)stream;
  SC_REQUIRE(a->terms >= 0 && a->terms <= 4, "sc_conv3x3_wgrad_bx3: terms must be 0 (= 3), 1, 2, 3 or SC_TERMS_F16X2 (got %d)", a->terms);
  p.absmax = a->absmax; p.xb0 = a->xbound[0]; p.xb1 = a->xbound[1];
  int nparts = pl.nsl * pl.kp;          // (the pipelined variant sums its K parts in the kernel: pl.nsl)
#define SC_WGX(WM_, NT_, NCI_, HF_, PIPE_) hipLaunchKernelGGL((k_wgrad3_bx3<WM_, NT_, NCI_, HF_, PIPE_>), grid, dim3(768), 0, st, p)
#define SC_WGX_NT(NT_, HF_, PIPE_)                                   \
  do {                                                               \
    if constexpr (HF_ && PIPE_) { if (pl.nci == 3) { SC_WGX(1, NT_, 3, HF_, PIPE_); break; } }      \
    if (pl.wm == 2 && pl.nci == 2) SC_WGX(2, NT_, 2, HF_, PIPE_);    \
    else if (pl.wm == 2) SC_WGX(2, NT_, 1, HF_, PIPE_);              \
    else if (pl.nci == 2) SC_WGX(1, NT_, 2, HF_, PIPE_);             \
    else SC_WGX(1, NT_, 1, HF_, PIPE_);                              \
  } while (0)
  if (a->terms == 1) SC_WGX_NT(1, false, false); else if (a->terms == 2) SC_WGX_NT(2, false, false);
  else if (a->terms == SC_TERMS_F16X2) {
    if (pipe) { SC_WGX_NT(2, true, true); if (wgrad3_pipe_reduces(pl)) nparts = pl.nsl; } else SC_WGX_NT(2, true, false);
  }
  else SC_WGX_NT(3, false, false);
#undef SC_WGX_NT
#undef SC_WGX
  SC_LAUNCH_OK("sc_conv3x3_wgrad_bx3");
  return sc_wgrad_finish(a->part, nparts, 9, a->Cout, a->Cin, pl.CoP, pl.CiP, a->dw, st);
}

static int wgrad_thin16_slices(int N, int H, int W, int Cin) {
  const long T = (long)N * ((W + 31) / 32) * ((H + 3) / 4 + 1);
  const long full = Cin > 16 ? 512 : 768;     // ONE full round: three work-groups per CU are resident with 16 input channels, two with 32
  return (int)(T < full ? T : full);
}

extern "C" size_t sc_wgrad_thin16_workspace_floats(int N, int H, int W, int Cout, int Cin) {
  (void)Cout;
  const size_t E = (size_t)9 * 16 * Cin;
  const int nparts = wgrad_thin16_slices(N, H, W, Cin);
  return (size_t)nparts * E + sc_reduce_scratch_floats(nparts, E);
}

extern "C" int sc_conv3x3_wgrad_thin16(const sc_wgrad_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr && a->ks == 3 && a->nsrc == 1, "sc_conv3x3_wgrad_thin16: one source, ks = 3");
  SC_REQUIRE(a->Cout >= 1 && a->Cout <= 16 && (a->Cin == 16 || a->Cin == 32) && a->src[0].C == a->Cin && a->dy.C == a->Cout,
             "sc_conv3x3_wgrad_thin16: needs <= 16 output and 16 or 32 input channels (got %d, %d)", a->Cout, a->Cin);
  SC_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0, "sc_conv3x3_wgrad_thin16: bad shape");
  SC_REQUIRE(a->terms == SC_TERMS_F16X2, "sc_conv3x3_wgrad_thin16: terms must be SC_TERMS_F16X2");
  SC_REQUIRE(a->dy.up == 0 && a->dy.mode != SC_SRC_NORM && (a->dy.mode != SC_SRC_BNBWD || a->dy.aux != nullptr), "sc_conv3x3_wgrad_thin16: unsupported dy source");
  SC_REQUIRE(a->dy.mode == SC_SRC_RAW || a->dy.cst != nullptr, "sc_conv3x3_wgrad_thin16: dy source needs constants");
  const sc_src& s = a->src[0];
  SC_REQUIRE((s.mode == SC_SRC_RAW || (s.mode == SC_SRC_AFFINE && s.cst != nullptr)), "sc_conv3x3_wgrad_thin16: input source must be RAW or AFFINE");
  SC_REQUIRE(s.up == 0 || (s.up == 1 && a->H % 2 == 0 && a->W % 2 == 0), "sc_conv3x3_wgrad_thin16: upsampled source needs even H, W");
  SC_REQUIRE((size_t)a->Cout * a->H * a->W * 4 < (1ull << 32) && (size_t)a->N * a->Cin * a->H * a->W * 4 < (1ull << 32),
             "sc_conv3x3_wgrad_thin16: tensors too large for 32-bit byte offsets");
  SC_REQUIRE(a->W % 2 == 0 && (((uintptr_t)a->dy.x | (uintptr_t)a->dy.aux) & 7) == 0, "sc_conv3x3_wgrad_thin16: needs an even width and 8-byte aligned gradient tensors");
  const size_t need = sc_wgrad_thin16_workspace_floats(a->N, a->H, a->W, a->Cout, a->Cin);
  SC_REQUIRE(a->part_floats >= need, "sc_conv3x3_wgrad_thin16: workspace too small (%zu < %zu floats)", a->part_floats, need);
  WgradXP p;
  p.absmax = a->absmax; p.xb0 = a->xbound[0]; p.xb1 = a->xbound[1];
  p.dy = to_srcd(a->dy); p.s0 = to_srcd(s); p.s1 = empty_srcd();
  p.N = a->N; p.H = a->H; p.W = a->W; p.Cout = a->Cout; p.Cin = a->Cin; p.part = a->part;
  p.nsl = wgrad_thin16_slices(a->N, a->H, a->W, a->Cin); p.CoP = 16; p.CiP = a->Cin;
  hipStream_t st = (hipStream_t)stream;
  const bool bnb = a->dy.mode == SC_SRC_BNBWD;
  dim3 grid(p.nsl);
  if (a->Cin == 16) { if (bnb) hipLaunchKernelGGL((k_wgrad_thin_h<16, true>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((k_wgrad_thin_h<16, false>), grid, dim3(256), 0, st, p); }
  else              { if (bnb) hipLaunchKernelGGL((k_wgrad_thin_h<32, true>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((k_wgrad_thin_h<32, false>), grid, dim3(256), 0, st, p); }
  SC_LAUNCH_OK("sc_conv3x3_wgrad_thin16");
  return sc_wgrad_finish(a->part, p.nsl, 9, a->Cout, a->Cin, 16, a->Cin, a->dw, st);
}

// K steps of the pack: the 3 x 3 filter's, and for the forward filter of a 32-channel layer the 16 phase steps of k_conv3_thin_sp behind them
static int thin_steps(int Cout, int Cin, int transpose_flip) {
  const int K = transpose_flip ? Cout : Cin;
  if (transpose_flip && Cin == 32) return 16;            // k_conv3_thin_spd: 8 K steps x 2 row blocks
  return (9 * (K / 8) + 3) / 4 + ((!transpose_flip && K == 32) ? 16 : 0);
}

extern "C" size_t sc_packed_weight_floats_thin16(int Cout, int Cin, int transpose_flip) {
  return (size_t)thin_steps(Cout, Cin, transpose_flip) * 2 * 64 * 4;       // 16-byte entries -> floats
}

static bool thin16_shape_ok(int Cout, int Cin, int transpose_flip) {
  const int M = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  if (transpose_flip && Cin == 32 && Cout == 16) return true;      // the half-resolution data gradient (k_conv3_thin_spd)
  return M >= 1 && M <= 16 && (K == 16 || K == 32);
}

extern "C" int sc_pack_weights_thin16(const float* w, float* wpk, int Cout, int Cin, int transpose_flip, sc_stream stream) {
  SC_REQUIRE(w && wpk && ((uintptr_t)wpk & 15) == 0, "sc_pack_weights_thin16: bad pointer");
  SC_REQUIRE(thin16_shape_ok(Cout, Cin, transpose_flip), "sc_pack_weights_thin16: needs <= 16 output and 16 or 32 input channels (got %d, %d, flip %d)",
             Cout, Cin, transpose_flip);
  const size_t total = (size_t)thin_steps(Cout, Cin, transpose_flip) * 512;
  hipLaunchKernelGGL(k_pack_weights_thin, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<unsigned short*>(wpk), Cout, Cin, transpose_flip, total);
  SC_LAUNCH_OK("sc_pack_weights_thin16");
  return SC_OK;
}

extern "C" int sc_conv3x3_thin16(const sc_conv_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr && a->ks == 3 && a->nsrc == 1, "sc_conv3x3_thin16: one source, ks = 3");
  const int Cin = a->src[0].C;
  if (a->down0) {
    // the data gradient of a 32 -> 16 channel layer whose source was up-sampled: 32 rows from 16 gradient channels, stored 2x2-summed
    // at half resolution (k_conv3_thin_spd; filters: sc_pack_weights_thin16(Cout = 16, Cin = 32, transpose_flip = 1))
    const sc_src& s = a->src[0];
    SC_REQUIRE(Cin == 16 && a->Cout == 32 && a->csplit == 32 && s.mode == SC_SRC_BNBWD && s.aux && s.cst && s.up == 0,
               "sc_conv3x3_thin16(down0): needs a 16-channel BatchNorm-backward source and 32 output channels (got %d, %d)", Cin, a->Cout);
    SC_REQUIRE(a->H % 2 == 0 && a->W % 2 == 0 && a->N > 0 && a->N <= 65535 && a->out0 && !a->add0 && !a->add1 && !a->stats && !a->bnr && !a->out1,
               "sc_conv3x3_thin16(down0): even H, W; one output, no add / statistics / bnr epilogue");
    SC_REQUIRE(a->terms == SC_TERMS_F16X2 && ((uintptr_t)a->wpk & 15) == 0, "sc_conv3x3_thin16(down0): terms must be SC_TERMS_F16X2, filters 16-byte aligned");
    SC_REQUIRE((size_t)16 * a->H * a->W < ((size_t)1 << 30), "sc_conv3x3_thin16(down0): one image of the gradient must stay below 2^30 elements");
    ConvXP p{};
    p.s0 = to_srcd(s); p.s1 = empty_srcd();
    p.wpk = reinterpret_cast<const uintx4*>(a->wpk); p.N = a->N; p.H = a->H; p.W = a->W; p.Cout = a->Cout;
    p.out0 = a->out0; p.csplit = a->Cout; p.accum0 = a->accum0; p.absmax = a->absmax;
    set_bnr(p, nullptr);
    const long total = (long)((a->W + 63) / 64) * ((a->H + 7) / 8) * a->N, per_xcd = (total + 7) / 8;
    p.xcdmap = 2;
    hipLaunchKernelGGL(k_conv3_thin_spd, dim3((unsigned)(per_xcd * 8)), dim3(256), 0, (hipStream_t)stream, p);
    SC_LAUNCH_OK("sc_conv3x3_thin16(down0)");
    return SC_OK;
  }
  SC_REQUIRE(a->Cout >= 1 && a->Cout <= 16 && (Cin == 16 || Cin == 32), "sc_conv3x3_thin16: needs <= 16 output and 16 or 32 input channels (got %d, %d)",
             a->Cout, Cin);
  SC_REQUIRE(a->N > 0 && a->N <= 65535 && a->H > 0 && a->W > 0, "sc_conv3x3_thin16: bad shape");
  SC_REQUIRE(a->csplit == a->Cout && !a->accum0 && !a->add0 && !a->add1 && !a->down0 && a->out0,
             "sc_conv3x3_thin16: a single plain output only (no split / add / accumulate / down-sum epilogue)");
  SC_REQUIRE(a->terms == SC_TERMS_F16X2, "sc_conv3x3_thin16: terms must be SC_TERMS_F16X2");
  SC_REQUIRE(((uintptr_t)a->wpk & 15) == 0, "sc_conv3x3_thin16: packed filters must be 16-byte aligned");
  const sc_src& s = a->src[0];
  SC_REQUIRE(s.up == 0 || (s.up == 1 && a->H % 2 == 0 && a->W % 2 == 0), "sc_conv3x3_thin16: upsampled source needs even H, W");
  SC_REQUIRE(s.mode == SC_SRC_RAW || s.cst != nullptr, "sc_conv3x3_thin16: source needs constants");
  SC_REQUIRE(s.mode != SC_SRC_NORM && (s.mode != SC_SRC_BNBWD || s.aux != nullptr), "sc_conv3x3_thin16: unsupported source");
  SC_REQUIRE((size_t)a->Cout * a->H * a->W < ((size_t)1 << 30), "sc_conv3x3_thin16: one image of the output must stay below 2^30 elements");
  ConvXP p{};
  p.s0 = to_srcd(s); p.s1 = empty_srcd();
  p.wpk = reinterpret_cast<const uintx4*>(a->wpk); p.N = a->N; p.H = a->H; p.W = a->W; p.Cout = a->Cout;
  p.out0 = a->out0; p.csplit = a->Cout; p.stats = a->stats; p.absmax = a->absmax; p.xb0 = a->xbound[0]; p.xb1 = a->xbound[1];
  set_bnr(p, a->bnr);
  SC_REQUIRE(!a->bnr || (s.mode == SC_SRC_BNBWD && a->bnr->y && a->bnr->cst && a->bnr->rows && !a->stats),
             "sc_conv3x3_thin16: bnr needs a BNBWD source, y / cst / rows and no stats");
  static const bool sp_off = [] { const char* e = getenv("STARCOP_THIN_SP"); return e && atoi(e) == 0; }();       // (same-box A/B)
  const bool thin_sp = Cin == 32 && s.up == 1 && s.mode != SC_SRC_BNBWD && !sp_off;      // the sub-pixel form (k_conv3_thin_sp)
  const int tw = (Cin == 16 || thin_sp) ? 64 : 32;        // (k_conv3_thin_h: TW)
  dim3 grid(((a->W + tw - 1) / tw) * ((a->H + 7) / 8), 1, a->N);
  constexpr int xcdmap_env = 2;
  p.xcdmap = xcdmap_env ? 2 : 0;
  if (p.xcdmap) {
    const long total = (long)grid.x * a->N, per_xcd = (total + 7) / 8;
    grid = dim3((unsigned)(per_xcd * 8));
  }
  hipStream_t st = (hipStream_t)stream;
  const bool bnb = s.mode == SC_SRC_BNBWD;
  static const bool ups_off = [] { const char* e = getenv("STARCOP_THIN_UPS"); return e && atoi(e) == 0; }();      // (same-box A/B)
  if (thin_sp) hipLaunchKernelGGL(k_conv3_thin_sp, grid, dim3(256), 0, st, p);
  else if (Cin == 16) { if (bnb) hipLaunchKernelGGL((k_conv3_thin_h<16, true>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((k_conv3_thin_h<16, false>), grid, dim3(256), 0, st, p); }
  else if (s.up == 1 && !bnb && !ups_off) hipLaunchKernelGGL((k_conv3_thin_h<32, false, true>), grid, dim3(256), 0, st, p);
  else           { if (bnb) hipLaunchKernelGGL((k_conv3_thin_h<32, true>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((k_conv3_thin_h<32, false>), grid, dim3(256), 0, st, p); }
  SC_LAUNCH_OK("sc_conv3x3_thin16");
  return SC_OK;
}

extern "C" size_t sc_pack_work_items(int Cout, int Cin, int ks, int co_t, int transpose_flip, int bx3) {
  if (bx3 == SC_PACK_THIN16) return (size_t)thin_steps(Cout, Cin, transpose_flip) * 512;
  if (bx3 == SC_PACK_SPD) return spd_pack_items(Cout, co_t, (transpose_flip & 3) == 3 ? Cin - co_t : 0);
  if (bx3 == SC_PACK_SP) return sp_pack_items(Cout, co_t, Cin - co_t);      // co_t = up-sampled channels (the leading ones of Cin)
  if (bx3 == SC_PACK_PW3) {
    const int M = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
    return (size_t)(((M + 31) / 32 + 3) / 4 * 4) * ((K + 15) / 16) * 512;
  }
  if (bx3) {
    const int M = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
    return (size_t)((M + co_t - 1) / co_t) * ((K + 15) / 16) * 9 * 2 * co_t * 8;
  }
  return sc_packed_weight_floats(Cout, Cin, ks, co_t, transpose_flip);
}

extern "C" int sc_pack_weights_batch(const sc_pack_desc* descs_dev, const uint32_t* block_starts_dev, int n, uint32_t total_blocks,
                                     sc_stream stream) {
  SC_REQUIRE(descs_dev && block_starts_dev && n > 0 && total_blocks > 0, "sc_pack_weights_batch: bad argument");
  hipLaunchKernelGGL(k_pack_batch, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const PackDesc*>(descs_dev), block_starts_dev, n);
  SC_LAUNCH_OK("sc_pack_weights_batch");
  return SC_OK;
}
