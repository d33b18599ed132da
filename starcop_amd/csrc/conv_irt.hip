// Fused TRAINING execution of a MobileNetV2 inverted-residual block's expansion + depthwise pair: the 6x-expanded tensor
//     e = conv1x1(x, W_e)            (torchvision InvertedResidual.conv[0], Cin -> hidden = 6 Cin channels)
// is never stored -- not in the forward pass, not its gradient in the backward pass.  (smp.Unet('mobilenet_v2') encoder blocks,
// /root/reference/starcop/models/model_module.py:244-251)  In features.2 - features.4 (256^2 / 128^2)
// e and dL/de are 150-400 MB each at batch 16 and the block's eight passes over them ARE its run time; recomputing e costs
// K = Cin <= 32 multiply-adds per element on the matrix cores against a 4-byte load.  Four sweeps replace
//     expand conv (+stats) | depthwise fwd | depthwise bwd (dx, dW, BN sums) | expand dgrad | expand wgrad :
//
//   k_irt_stats   x -> per-channel sum / sum of squares of e (for the expansion's train-mode BatchNorm); e stays in registers
//   k_irt_fwd     x -> e (MFMA) -> BN + ReLU6 -> LDS -> 3x3 depthwise stencil (stride 1 | 2) -> RAW d + its statistics rows
//   k_irt_bsums   (dy_d, x) -> recomputed e, g_e = depthwise-dgrad(dy_d) in registers -> BatchNorm-backward sums of e,
//                 depthwise filter gradient, and G[h][ci] = sum_px g'_e[h][px] x[ci][px]  (+ the second moments of x)
//   k_irt_bdata   (dy_d, x) -> recomputed e, g_e, dy_e = A g'_e + B e + D -> dx = W_e^T dy_e  (chained MFMA, no LDS transpose)
//   k_irt_dwe     dW_e = A (.) G + B (.) (W_e M) + D (x) s      -- exact, because dy_e is affine in (g'_e, e) per channel and
//                 e = W_e x:  sum_px dy_e x^T = A sum g'_e x^T + B W_e sum x x^T + D sum x^T;  M = sum x x^T, s = sum x.
//
// Arithmetic: every fp32 MFMA operand is split exactly into three bf16 terms, six products, fp32 accumulation (the "fp32-x3"
// arithmetic of conv_pw3.hip: one fp32 rounding per product, fp32's exponent range, no scaling or range assumptions).
// MFMA v_mfma_f32_32x32x16_bf16:  A (32 x 16): lane l -> row l&31, k = 8*(l>>5)..+7;  B (16 x 32): lane l -> column l&31, same k;
// D: register i of lane l = D[row 8*(i/4) + 4*(l>>5) + (i%4)][column l&31].  Two orientations of the same e = W_e x block:
//   (1) rows = pixels, columns = hidden channels: a lane owns ONE channel and 16 pixels -> per-channel sums are in-lane, and with
//       a 4 x 8 pixel block the 16 pixels are a 4 x 4 patch (stencils from 36 LDS reads); the registers are the A operand of a
//       contraction over PIXELS (G);
//   (2) rows = hidden channels, columns = pixels: a lane owns ONE pixel and 16 channels -> the registers are the B operand of a
//       contraction over HIDDEN channels (dx) with the filter rows permuted to the accumulator's channel order.
#include "sc_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) unsigned int uintx4;

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
__device__ __forceinline__ void split8(const float (&v)[8], uintx4 (&t)[3]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned t0, t1, t2;
    split3x2(v[2 * q], v[2 * q + 1], t0, t1, t2);
    t[0][q] = t0; t[1][q] = t1; t[2][q] = t2;
  }
}
__device__ __forceinline__ floatx16 mfma_bf16(const uintx4& a, const uintx4& b, const floatx16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six products of weight >= 2^-24, smallest first
__device__ __forceinline__ floatx16 mfma6(const uintx4 (&a)[3], const uintx4 (&b)[3], floatx16 c) {
  c = mfma_bf16(a[1], b[1], c);
  c = mfma_bf16(a[2], b[0], c);
  c = mfma_bf16(a[0], b[2], c);
  c = mfma_bf16(a[1], b[0], c);
  c = mfma_bf16(a[0], b[1], c);
  c = mfma_bf16(a[0], b[0], c);
  return c;
}
__device__ __forceinline__ floatx16 zero16() {
  floatx16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
__device__ __forceinline__ float wave_total(float v) {      // sum over the 64 lanes, uniform result (DPP row sums + two readlanes)
  v = half_sum32(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
}

constexpr int IRT_MAXCH = 6;        // hidden <= 192

struct IrtP {
  SrcD x;                  // block input: RAW (residual sum) or AFFINE (BatchNorm'd projection of the previous block); cst never NULL
  const float* we;         // expansion filter [Hd][Cin]
  const float* wd;         // depthwise filter [Hd][9]
  const float* cst_e;      // [Hd][SC_CST] forward constants of the expansion's BatchNorm (scale, shift, mean, invstd)
  const float* cstb_e;     // [Hd][SC_CST] its backward constants (.., .., A, B, D)                      (k_irt_bdata, k_irt_dwe)
  SrcD dy;                 // gradient of the depthwise conv's RAW output: BNBWD source (g_d, d, constants) or RAW     (backward)
  float* stats;            // [rows][Hd][2] partial sums                                                 (k_irt_stats, k_irt_fwd)
  float* dout;             // raw depthwise output [N][Hd][Ho][Wo]                                       (k_irt_fwd)
  double* esums;           // [rows][Hd][2] BatchNorm-backward sums of e                                 (k_irt_bsums)
  double* dwacc;           // [Hd][9] depthwise filter gradient, fp64 atomics                            (k_irt_bsums)
  float* gpart;            // [rows][nch*32][32]                                                         (k_irt_bsums)
  float* mpart;            // [rows][33][32]: rows 0..31 = M[ci][ci'], row 32 = s[ci']                    (k_irt_bsums)
  float* dx; const float* add0; int accum;                                                            // (k_irt_bdata)
  int N, Cin, Hd, H, W, S, Ho, Wo, nch;
  int tiles_x, tiles_y, ntiles, tiles_per_wg, npb;
};

// ---- operand builders ---------------------------------------------------------------------------------------------------------
// block input as an MFMA operand: lane -> ONE pixel (xb points at channel 0 of it), k-slots = channels ks*16 + lhi*8 + j
template <int NKS>
__device__ __forceinline__ void irt_xop(const IrtP& p, const float* __restrict__ s_xc, const float* __restrict__ xb, bool ok, int lhi,
                                        float lo, float hi, size_t HW, uintx4 (&op)[NKS][3]) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ks * 16 + lhi * 8 + j;
      const int cc = c < p.Cin ? c : p.Cin - 1;
      const float raw = xb[(size_t)cc * HW];
      const float t = sc_pro_affine(raw, s_xc[cc * 2], s_xc[cc * 2 + 1], lo, hi);
      v[j] = (ok && c < p.Cin) ? t : 0.f;
    }
    split8(v, op[ks]);
  }
}
// expansion filter as an MFMA operand: lane -> hidden channel h, k-slots = input channels ks*16 + lhi*8 + j
template <int NKS>
__device__ __forceinline__ void irt_wop(const IrtP& p, int h, int lhi, uintx4 (&op)[NKS][3]) {
  const bool hok = h < p.Hd;
  const float* wr = p.we + (size_t)(hok ? h : 0) * p.Cin;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ks * 16 + lhi * 8 + j;
      const float t = wr[c < p.Cin ? c : p.Cin - 1];
      v[j] = (hok && c < p.Cin) ? t : 0.f;
    }
    split8(v, op[ks]);
  }
}
__device__ __forceinline__ void irt_xconsts(const IrtP& p, float* s_xc) {
  if (threadIdx.x < 32) {
    const int c = (int)threadIdx.x < p.Cin ? (int)threadIdx.x : p.Cin - 1;
    s_xc[threadIdx.x * 2] = p.x.cst[(size_t)c * SC_CST];
    s_xc[threadIdx.x * 2 + 1] = p.x.cst[(size_t)c * SC_CST + 1];
  }
}

// =================================================================================================================================
// (A) statistics of e: flat 32-pixel blocks of the whole batch, grid-stride; the filter operands of every chunk sit in LDS
template <int NKS>
__global__ __launch_bounds__(256) void k_irt_stats(const IrtP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uintx4* s_w = reinterpret_cast<uintx4*>(smem);                                   // [nch][NKS][3][64]
  float* s_xc = reinterpret_cast<float*>(s_w + (size_t)p.nch * NKS * 3 * 64);       // [32][2]
  float* s_red = s_xc + 64;                                                        // [4][nch*32][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nch = p.nch;
  for (int ch = wave; ch < nch; ch += 4) {
    uintx4 op[NKS][3];
    irt_wop<NKS>(p, ch * 32 + l31, lhi, op);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int t = 0; t < 3; ++t) s_w[((size_t)(ch * NKS + ks) * 3 + t) * 64 + lane] = op[ks][t];
  }
  irt_xconsts(p, s_xc);
  __syncthreads();
  const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
  const size_t HW = (size_t)p.H * p.W;
  const long NP = (long)p.N * (long)HW;
  float s1[IRT_MAXCH], s2[IRT_MAXCH];
#pragma unroll
  for (int ch = 0; ch < IRT_MAXCH; ++ch) { s1[ch] = 0.f; s2[ch] = 0.f; }
  for (int pb = blockIdx.x * 4 + wave; pb < p.npb; pb += gridDim.x * 4) {
    const long gp = (long)pb * 32 + l31;
    const bool ok = gp < NP;
    const long gpc = ok ? gp : 0;
    const int n = (int)(gpc / (long)HW);
    const size_t px = (size_t)(gpc - (long)n * (long)HW);
    uintx4 xop[NKS][3];
    irt_xop<NKS>(p, s_xc, p.x.x + (size_t)n * p.Cin * HW + px, ok, lhi, lo, hi, HW, xop);
#pragma unroll
    for (int ch = 0; ch < IRT_MAXCH; ++ch) {
      if (ch < nch) {
        floatx16 acc = zero16();
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          uintx4 w[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) w[t] = s_w[((size_t)(ch * NKS + ks) * 3 + t) * 64 + lane];
          acc = mfma6(xop[ks], w, acc);            // rows = pixels, columns = hidden channels
        }
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a += acc[r]; b = fmaf(acc[r], acc[r], b); }
        s1[ch] += a; s2[ch] += b;
      }
    }
  }
#pragma unroll
  for (int ch = 0; ch < IRT_MAXCH; ++ch) {
    if (ch < nch) {
      const float a = s1[ch] + __shfl_xor(s1[ch], 32, 64), b = s2[ch] + __shfl_xor(s2[ch], 32, 64);
      if (lhi == 0) {
        s_red[((size_t)(wave * nch + ch) * 32 + l31) * 2] = a;
        s_red[((size_t)(wave * nch + ch) * 32 + l31) * 2 + 1] = b;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < nch * 32; i += 256) {
    if (i < p.Hd) {
      const size_t q = (size_t)nch * 32 * 2;
      const float a = ((s_red[i * 2] + s_red[q + i * 2]) + s_red[2 * q + i * 2]) + s_red[3 * q + i * 2];
      const float b = ((s_red[i * 2 + 1] + s_red[q + i * 2 + 1]) + s_red[2 * q + i * 2 + 1]) + s_red[3 * q + i * 2 + 1];
      *reinterpret_cast<float2*>(p.stats + ((size_t)blockIdx.x * p.Hd + i) * 2) = make_float2(a, b);
    }
  }
}

// =================================================================================================================================
// (B) forward: one work-group = one output tile of one image (stride 1: 8 x 32, stride 2: 4 x 16 outputs); its input tile of e
// (+ the 3x3 halo) is recomputed chunk by chunk (32 hidden channels) into LDS and the stencil runs on it.
//   phase 1  wave w: pixel blocks w, w+4, ... of the flattened e tile: e = x * W_e (MFMA, the x operands stay in registers over
//            all chunks), BN + ReLU6, ZERO outside the image (the depthwise conv pads its activated input) -> s_e[channel][pixel]
//   phase 2  wave w: channels 8w .. 8w+7 of the chunk, lanes = output pixels: 3x3 stencil, raw d -> HBM, the tile's per-channel
//            sum / sum of squares of d -> one statistics row per tile
template <int S> struct IrtFwdGeo {
  static constexpr int TH = S == 1 ? 8 : 4, TW = S == 1 ? 32 : 16;
  static constexpr int EH = (TH - 1) * S + 3, EW = (TW - 1) * S + 3, EPX = EH * EW;
  static constexpr int NBLK = (EPX + 31) / 32, BPW = (NBLK + 3) / 4, EPAD = NBLK * 32 + 4;      // EPAD % 32 == 4: conflict-free b128 stores
  static constexpr int RPL = TH * TW / 64;                                                        // output rows per lane (4 | 1)
};

template <int NKS, int S>
__global__ __launch_bounds__(256) void k_irt_fwd(const IrtP p) {
  using G = IrtFwdGeo<S>;
  constexpr int TH = G::TH, TW = G::TW, EW = G::EW, EPX = G::EPX, NBLK = G::NBLK, BPW = G::BPW, EPAD = G::EPAD, RPL = G::RPL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_e = reinterpret_cast<float*>(smem);              // [32][EPAD]
  float* s_xc = s_e + 32 * EPAD;                            // [32][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int tile = blockIdx.x;
  const int n = tile / (p.tiles_x * p.tiles_y), tt = tile - n * p.tiles_x * p.tiles_y;
  const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
  const int ey0 = oy0 * S - 1, ex0 = ox0 * S - 1;
  const int H = p.H, W = p.W, Hd = p.Hd;
  const size_t HW = (size_t)H * W, HWo = (size_t)p.Ho * p.Wo;
  irt_xconsts(p, s_xc);
  __syncthreads();
  const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
  // the x operands of this wave's pixel blocks and the inside-the-image bits of its accumulator pixels
  uintx4 xop[BPW][NKS][3];
  unsigned inside[BPW];
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    const int blk = wave + 4 * b;
    const int f = blk * 32 + l31;
    const int ey = f / EW, ex = f - ey * EW;
    const int y = ey0 + ey, x = ex0 + ex;
    const bool ok = blk < NBLK && f < EPX && y >= 0 && y < H && x >= 0 && x < W;
    irt_xop<NKS>(p, s_xc, p.x.x + (size_t)n * p.Cin * HW + (ok ? (size_t)y * W + x : 0), ok, lhi, lo, hi, HW, xop[b]);
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int fi = blk * 32 + 8 * (i >> 2) + 4 * lhi + (i & 3);
      const int eyi = fi / EW, exi = fi - eyi * EW;
      const int yi = ey0 + eyi, xi = ex0 + exi;
      if (fi < EPX && yi >= 0 && yi < H && xi >= 0 && xi < W) bits |= 1u << i;
    }
    inside[b] = bits;
  }
  // output pixels of this lane (phase 2)
  const int lox = lane % TW, lrg = lane / TW;                 // column, row group
  const int oyl = lrg * RPL;                                  // first output row of the lane within the tile
  const bool colok = ox0 + lox < p.Wo;

  for (int chunk = 0; chunk < p.nch; ++chunk) {
    // ---------------- phase 1
    {
      const int h = chunk * 32 + l31;
      uintx4 wop[NKS][3];
      irt_wop<NKS>(p, h, lhi, wop);
      const int hc = h < Hd ? h : Hd - 1;
      const float sc = p.cst_e[(size_t)hc * SC_CST], sh = p.cst_e[(size_t)hc * SC_CST + 1];
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        const int blk = wave + 4 * b;
        if (blk < NBLK) {
          floatx16 acc = zero16();
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) acc = mfma6(xop[b][ks], wop[ks], acc);        // rows = pixels, columns = hidden channels
          float* dst = s_e + l31 * EPAD + blk * 32 + 4 * lhi;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 v;
            float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float t = fminf(fmaxf(fmaf(acc[4 * j + q], sc, sh), 0.f), 6.f);
              vv[q] = ((inside[b] >> (4 * j + q)) & 1u) ? t : 0.f;
            }
            *reinterpret_cast<float4*>(dst + 8 * j) = v;
          }
        }
      }
    }
    __syncthreads();
    // ---------------- phase 2
#pragma unroll 2
    for (int cc = 0; cc < 8; ++cc) {
      const int cl = wave * 8 + cc;
      const int h = chunk * 32 + cl;                          // wave-uniform
      if (h < Hd) {
        float wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = p.wd[(size_t)h * 9 + k];
        const float* e = s_e + cl * EPAD + (oyl * S) * EW + lox * S;
        float s1 = 0.f, s2 = 0.f;
        if (S == 1) {
          // sliding window down RPL rows of the lane's column
          float r0[3], r1[3], r2[3];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) { r0[kx] = e[kx]; r1[kx] = e[EW + kx]; }
#pragma unroll
          for (int rr = 0; rr < RPL; ++rr) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) r2[kx] = e[(rr + 2) * EW + kx];
            float a = 0.f;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) { a = fmaf(wk[kx], r0[kx], a); a = fmaf(wk[3 + kx], r1[kx], a); a = fmaf(wk[6 + kx], r2[kx], a); }
            const int oy = oy0 + oyl + rr;
            const bool ok = colok && oy < p.Ho;
            if (ok) p.dout[((size_t)n * Hd + h) * HWo + (size_t)oy * p.Wo + ox0 + lox] = a;
            const float am = ok ? a : 0.f;
            s1 += am; s2 = fmaf(am, am, s2);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) { r0[kx] = r1[kx]; r1[kx] = r2[kx]; }
          }
        } else {
          float a = 0.f;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a = fmaf(wk[ky * 3 + kx], e[ky * EW + kx], a);
          const int oy = oy0 + oyl;
          const bool ok = colok && oy < p.Ho;
          if (ok) p.dout[((size_t)n * Hd + h) * HWo + (size_t)oy * p.Wo + ox0 + lox] = a;
          const float am = ok ? a : 0.f;
          s1 = am; s2 = am * am;
        }
        const float t1 = wave_total(s1), t2 = wave_total(s2);
        if (lane == 0) *reinterpret_cast<float2*>(p.stats + ((size_t)tile * Hd + h) * 2) = make_float2(t1, t2);
      }
    }
    __syncthreads();
  }
}

// =================================================================================================================================
// staging of the depthwise output's gradient for a tile of 8 x 32 e pixels: region of d that the tile's e pixels feed
//   stride 1: rows r0-1 .. r0+8, columns c0-1 .. c0+32 (10 x 34);  stride 2: rows r0/2 .. r0/2+4, columns c0/2 .. c0/2+16 (5 x 17)
// s_dy[channel][PITCH]; zeros outside the image and for channels past Hd.  Wave w stages channels w, w+4, ...
template <int S> struct IrtBwdGeo {
  static constexpr int RH = S == 1 ? 10 : 5, RW = S == 1 ? 34 : 17, RSZ = RH * RW;
  static constexpr int PITCH = (RSZ | 1);                    // odd: lanes = channels read conflict-free
};
template <int S>
__device__ __forceinline__ void irt_stage_dy(const IrtP& p, float* __restrict__ s_dy, const float* __restrict__ s_dc, int n, int chunk,
                                             int r0, int c0, int lane, int wave) {
  using B = IrtBwdGeo<S>;
  const int ry0 = S == 1 ? r0 - 1 : r0 / 2, rx0 = S == 1 ? c0 - 1 : c0 / 2;
  const size_t HWo = (size_t)p.Ho * p.Wo;
  const float lo = sc_act_lo(p.dy.act), hi = sc_act_hi(p.dy.act);
  const bool bnb = p.dy.mode == SC_SRC_BNBWD;
  for (int c = wave; c < 32; c += 4) {
    const int h = chunk * 32 + c;
    const bool hok = h < p.Hd;
    const size_t cb = ((size_t)n * p.Hd + (hok ? h : 0)) * HWo;
    const float k0 = s_dc[c * 8], k1 = s_dc[c * 8 + 1], kA = s_dc[c * 8 + 2], kB = s_dc[c * 8 + 3], kD = s_dc[c * 8 + 4];
    for (int r = lane; r < B::RSZ; r += 64) {
      const int ry = r / B::RW, rx = r - ry * B::RW;
      const int oy = ry0 + ry, ox = rx0 + rx;
      const bool inb = hok && oy >= 0 && oy < p.Ho && ox >= 0 && ox < p.Wo;
      const size_t o = cb + (inb ? (size_t)oy * p.Wo + ox : 0);
      const float g = p.dy.x[o];
      float v = g;
      if (bnb) v = sc_pro_bnbwd(g, p.dy.aux[o], k0, k1, kA, kB, kD, lo, hi);
      s_dy[c * B::PITCH + r] = inb ? v : 0.f;
    }
  }
}
// constants of the gradient source for the 32 channels of a chunk -> s_dc[32][8] (identity for a RAW source)
__device__ __forceinline__ void irt_dy_consts(const IrtP& p, float* s_dc, int chunk) {
  if (threadIdx.x < 32) {
    const int h = chunk * 32 + threadIdx.x;
    const int hc = h < p.Hd ? h : p.Hd - 1;
    float4 c0 = make_float4(1.f, 0.f, 1.f, 0.f); float c4 = 0.f;
    if (p.dy.mode == SC_SRC_BNBWD) { c0 = *reinterpret_cast<const float4*>(p.dy.cst + (size_t)hc * SC_CST); c4 = p.dy.cst[(size_t)hc * SC_CST + 4]; }
    *reinterpret_cast<float4*>(s_dc + threadIdx.x * 8) = c0;
    *reinterpret_cast<float4*>(s_dc + threadIdx.x * 8 + 4) = make_float4(c4, 0.f, 0.f, 0.f);
  }
}

// =================================================================================================================================
// (Bi) backward sums.  grid = (tile sets, chunks): a work-group owns ONE chunk of 32 hidden channels and walks its tiles of
// 8 x 32 e pixels; a wave owns two 4 x 8 pixel blocks per tile (orientation 1: lane = channel, registers = a 4 x 4 pixel patch).
template <int NKS, int S>
__global__ __launch_bounds__(256) void k_irt_bsums(const IrtP p) {
  using B = IrtBwdGeo<S>;
  constexpr int RW = B::RW, PITCH = B::PITCH;
  constexpr int LP = S == 1 ? 6 : 3;                         // rows / columns of dy_d a 4 x 4 patch touches
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_dy = reinterpret_cast<float*>(smem);              // [32][PITCH]
  float* s_dc = s_dy + 32 * PITCH;                           // [32][8]
  float* s_xc = s_dc + 256;                                  // [32][2]
  float* s_red = s_xc + 64;                                  // reduction scratch: [3][16][64] floats (>= [4][32][12])
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int chunk = blockIdx.y;
  const int H = p.H, W = p.W, Hd = p.Hd, Cin = p.Cin;
  const size_t HW = (size_t)H * W;
  irt_xconsts(p, s_xc);
  irt_dy_consts(p, s_dc, chunk);
  // this lane's hidden channel
  const int h = chunk * 32 + l31;
  const bool hok = h < Hd;
  const int hc = hok ? h : Hd - 1;
  const float4 ce = *reinterpret_cast<const float4*>(p.cst_e + (size_t)hc * SC_CST);      // scale, shift, mean, invstd
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = hok ? p.wd[(size_t)hc * 9 + k] : 0.f;
  uintx4 wop[NKS][3];
  irt_wop<NKS>(p, h, lhi, wop);
  const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
  // x^T operand constants: this lane's input channel l31
  const bool ciok = l31 < Cin;
  const int cic = ciok ? l31 : Cin - 1;
  const float xsc = p.x.cst[(size_t)cic * SC_CST], xsh = p.x.cst[(size_t)cic * SC_CST + 1];

  float s1 = 0.f, s2 = 0.f, sx = 0.f, dwd[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) dwd[k] = 0.f;
  floatx16 accG = zero16(), accM = zero16();
  const int t_begin = blockIdx.x * p.tiles_per_wg;
  const int t_end = min(t_begin + p.tiles_per_wg, p.ntiles);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int n = tile / (p.tiles_x * p.tiles_y), tt = tile - n * p.tiles_x * p.tiles_y;
    const int r0 = (tt / p.tiles_x) * 8, c0 = (tt % p.tiles_x) * 32;
    __syncthreads();                                           // the previous tile's stencil reads are done (and the constants are in LDS)
    irt_stage_dy<S>(p, s_dy, s_dc, n, chunk, r0, c0, lane, wave);
    // operands of this wave's two pixel blocks (global loads: in flight across the barrier)
    uintx4 xop[2][NKS][3], xT[2][2][3];
    bool bok[2];
    float sxl = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int b = wave * 2 + t;
      const int pr = r0 + 4 * (b >> 2), pc0 = c0 + 8 * (b & 3);
      bok[t] = pr < H && pc0 < W;                              // H % 4 == 0, W % 8 == 0: a block is inside or outside as a whole
      const int y = pr + (l31 >> 3), x = pc0 + (l31 & 7);
      irt_xop<NKS>(p, s_xc, p.x.x + (size_t)n * Cin * HW + (bok[t] ? (size_t)y * W + x : 0), bok[t], lhi, lo, hi, HW, xop[t]);
      // x^T: lane -> input channel l31, k-slots (step s) = pixels (row 2s + j/4, column 4*lhi + j%4) of the block
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const size_t o = ((size_t)n * Cin + cic) * HW + (bok[t] ? (size_t)(pr + 2 * s + rr) * W + pc0 + 4 * lhi : 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float tv = sc_pro_affine(p.x.x[o + q], xsc, xsh, lo, hi);
            v[4 * rr + q] = (bok[t] && ciok) ? tv : 0.f;
          }
        }
        if (chunk == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) sxl += v[j];
        }
        split8(v, xT[t][s]);
      }
    }
    sx += sxl;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!bok[t]) continue;                                   // wave-uniform
      const int b = wave * 2 + t;
      floatx16 acc = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) acc = mfma6(xop[t][ks], wop[ks], acc);        // e[pixel (i/4, 4*lhi + i%4) of the block][channel l31]
      // dy_d values the patch touches
      float L[LP][LP];
      {
        const float* base = s_dy + l31 * PITCH + (S == 1 ? (4 * (b >> 2)) * RW + 8 * (b & 3) + 4 * lhi
                                                          : (2 * (b >> 2)) * RW + 4 * (b & 3) + 2 * lhi);
#pragma unroll
        for (int a = 0; a < LP; ++a)
#pragma unroll
          for (int q = 0; q < LP; ++q) L[a][q] = base[a * RW + q];
      }
      float gq[16];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float e = acc[4 * u + v];
          const float yh = fmaf(e, ce.x, ce.y);
          const bool pass = yh > 0.f && yh < 6.f;
          const float eh = fminf(fmaxf(yh, 0.f), 6.f);
          const float xn = (e - ce.z) * ce.w;
          float g = 0.f;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              // output pixel o = (i + 1 - k) / S  (stride 2: only taps with i + 1 - k even); patch-local index into L
              if (S == 1) {
                const float d = L[u + 2 - ky][v + 2 - kx];
                g = fmaf(wk[ky * 3 + kx], d, g);
                dwd[ky * 3 + kx] = fmaf(eh, d, dwd[ky * 3 + kx]);
              } else if (((u + 1 - ky) & 1) == 0 && ((v + 1 - kx) & 1) == 0) {
                const float d = L[(u + 1 - ky) / 2][(v + 1 - kx) / 2];
                g = fmaf(wk[ky * 3 + kx], d, g);
                dwd[ky * 3 + kx] = fmaf(eh, d, dwd[ky * 3 + kx]);
              }
            }
          const float gm = pass ? g : 0.f;
          s1 += gm;
          s2 = fmaf(gm, xn, s2);
          gq[4 * u + v] = gm;
        }
      // G[h][ci] += sum over the block's 32 pixels of g'_e[h][px] x[ci][px]   (A = g'_e: rows = channels, k = pixels)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = gq[8 * s + j];
        uintx4 ga[3];
        split8(v8, ga);
        accG = mfma6(ga, xT[t][s], accG);
        if (chunk == 0) accM = mfma6(xT[t][s], xT[t][s], accM);
      }
    }
  }
  // ---------------- epilogue: the work-group's partial row
  __syncthreads();
  {
    // per-channel scalars: halves, then waves (fixed order)
    float v[12];
    v[0] = s1; v[1] = s2;
#pragma unroll
    for (int k = 0; k < 9; ++k) v[2 + k] = dwd[k];
    v[11] = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) v[k] += __shfl_xor(v[k], 32, 64);
    if (lhi == 0) {
#pragma unroll
      for (int k = 0; k < 11; ++k) s_red[(wave * 32 + l31) * 12 + k] = v[k];
    }
  }
  __syncthreads();
  if (tid < 32 && chunk * 32 + tid < Hd) {
    float v[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) v[k] = ((s_red[tid * 12 + k] + s_red[(32 + tid) * 12 + k]) + s_red[(64 + tid) * 12 + k]) + s_red[(96 + tid) * 12 + k];
    const int hh = chunk * 32 + tid;
    p.esums[((size_t)blockIdx.x * Hd + hh) * 2] = (double)v[0];
    p.esums[((size_t)blockIdx.x * Hd + hh) * 2 + 1] = (double)v[1];
#pragma unroll
    for (int k = 0; k < 9; ++k) atomicAdd(p.dwacc + (size_t)hh * 9 + k, (double)v[2 + k]);
  }
  __syncthreads();
  // G: waves 1..3 -> LDS, wave 0 adds them in order
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[((wave - 1) * 16 + r) * 64 + lane] = accG[r];
  }
  __syncthreads();
  if (wave == 0) {
    float* part = p.gpart + ((size_t)blockIdx.x * p.nch + chunk) * 32 * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = ((accG[r] + s_red[r * 64 + lane]) + s_red[(16 + r) * 64 + lane]) + s_red[(32 + r) * 64 + lane];
      part[(8 * (r >> 2) + 4 * lhi + (r & 3)) * 32 + l31] = v;
    }
  }
  if (chunk == 0) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_red[((wave - 1) * 16 + r) * 64 + lane] = accM[r];
    }
    __syncthreads();
    float* mp = p.mpart + (size_t)blockIdx.x * 33 * 32;
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = ((accM[r] + s_red[r * 64 + lane]) + s_red[(16 + r) * 64 + lane]) + s_red[(32 + r) * 64 + lane];
        mp[(8 * (r >> 2) + 4 * lhi + (r & 3)) * 32 + l31] = v;
      }
    }
    __syncthreads();
    const float sh2 = sx + __shfl_xor(sx, 32, 64);
    if (lhi == 0) s_red[wave * 32 + l31] = sh2;
    __syncthreads();
    if (tid < 32) mp[32 * 32 + tid] = ((s_red[tid] + s_red[32 + tid]) + s_red[64 + tid]) + s_red[96 + tid];
  }
}

// =================================================================================================================================
// (Bii) backward data.  One work-group = one tile of 8 x 32 e pixels; a wave owns two image rows of 32 pixels (orientation 2:
// lane = pixel, registers = 16 hidden channels); dx accumulates over the chunks in registers.
template <int NKS, int S>
__global__ __launch_bounds__(256) void k_irt_bdata(const IrtP p) {
  using B = IrtBwdGeo<S>;
  constexpr int RW = B::RW, PITCH = B::PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_dy = reinterpret_cast<float*>(smem);              // [32][PITCH]
  float* s_dc = s_dy + 32 * PITCH;                           // [32][8]
  float* s_xc = s_dc + 256;                                  // [32][2]
  float* s_k = s_xc + 64;                                    // [2][5][16]: scale, shift, A, B, D by (lhi, accumulator register)
  float* s_wd = s_k + 160;                                   // [2][9][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, W = p.W, Hd = p.Hd, Cin = p.Cin;
  const size_t HW = (size_t)H * W;
  const int tile = blockIdx.x;
  const int n = tile / (p.tiles_x * p.tiles_y), tt = tile - n * p.tiles_x * p.tiles_y;
  const int r0 = (tt / p.tiles_x) * 8, c0 = (tt % p.tiles_x) * 32;
  irt_xconsts(p, s_xc);
  __syncthreads();
  const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
  const int ix = c0 + l31;
  uintx4 xop[2][NKS][3];
  bool pok[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int iy = r0 + 2 * wave + t;
    pok[t] = iy < H && ix < W;
    irt_xop<NKS>(p, s_xc, p.x.x + (size_t)n * Cin * HW + (pok[t] ? (size_t)iy * W + ix : 0), pok[t], lhi, lo, hi, HW, xop[t]);
  }
  floatx16 accdx[2] = {zero16(), zero16()};
  const bool ciok = l31 < Cin;

  for (int chunk = 0; chunk < p.nch; ++chunk) {
    __syncthreads();                                           // the previous chunk's reads of s_dy / s_k / s_wd are done
    irt_dy_consts(p, s_dc, chunk);
    if (tid < 32) {
      // constants / depthwise taps by (lhi, register): hidden = chunk*32 + 8*(i/4) + 4*lhi + (i%4)
      const int lh = tid >> 4, i = tid & 15;
      const int hh = chunk * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
      const bool ok = hh < Hd;
      const int hc = ok ? hh : Hd - 1;
      s_k[(lh * 5 + 0) * 16 + i] = ok ? p.cst_e[(size_t)hc * SC_CST] : 0.f;
      s_k[(lh * 5 + 1) * 16 + i] = ok ? p.cst_e[(size_t)hc * SC_CST + 1] : 0.f;
      s_k[(lh * 5 + 2) * 16 + i] = ok ? p.cstb_e[(size_t)hc * SC_CST + 2] : 0.f;
      s_k[(lh * 5 + 3) * 16 + i] = ok ? p.cstb_e[(size_t)hc * SC_CST + 3] : 0.f;
      s_k[(lh * 5 + 4) * 16 + i] = ok ? p.cstb_e[(size_t)hc * SC_CST + 4] : 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) s_wd[(lh * 9 + k) * 16 + i] = ok ? p.wd[(size_t)hc * 9 + k] : 0.f;
    }
    __syncthreads();                                           // s_dc ready for the staging
    irt_stage_dy<S>(p, s_dy, s_dc, n, chunk, r0, c0, lane, wave);
    // filter operands of this chunk: rows of W_e (for e) and the permuted rows of W_e^T (for dx)
    uintx4 wop[NKS][3], wT[2][3];
    irt_wop<NKS>(p, chunk * 32 + l31, lhi, wop);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int hh = chunk * 32 + 16 * s + 8 * (j >> 2) + 4 * lhi + (j & 3);
        const float t = p.we[(size_t)(hh < Hd ? hh : Hd - 1) * Cin + (ciok ? l31 : 0)];
        v[j] = (hh < Hd && ciok) ? t : 0.f;
      }
      split8(v, wT[s]);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int iyl = 2 * wave + t;                            // row of the tile (wave-uniform)
      floatx16 acc = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) acc = mfma6(wop[ks], xop[t][ks], acc);        // e[channel 8*(i/4) + 4*lhi + (i%4)][pixel l31]
      float g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        int rowl;                                              // row of the staged region
        if (S == 1) rowl = iyl + 2 - ky;
        else {
          if (((iyl + 1 - ky) & 1) != 0) continue;             // wave-uniform
          rowl = (iyl + 1 - ky) >> 1;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          int coll; bool cv = true;
          if (S == 1) coll = l31 + 2 - kx;
          else { const int q = l31 + 1 - kx; cv = (q & 1) == 0 && q >= 0; coll = cv ? (q >> 1) : 0; }
          const float* src = s_dy + rowl * RW + coll;
          const float4* wq = reinterpret_cast<const float4*>(s_wd + (lhi * 9 + ky * 3 + kx) * 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 w4 = wq[j];
            const float* ww = reinterpret_cast<const float*>(&w4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int i = 4 * j + q;
              const float d = src[(8 * (i >> 2) + 4 * lhi + (i & 3)) * PITCH];
              g[i] = fmaf(ww[q], (S == 1 || cv) ? d : 0.f, g[i]);
            }
          }
        }
      }
      // dy_e = A g' + B e + D, straight into the B operand of dx = W_e^T dy_e
      float dye[16];
      {
        const float4* kq = reinterpret_cast<const float4*>(s_k + lhi * 5 * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 ksc = kq[j], ksh = kq[4 + j], kA = kq[8 + j], kB = kq[12 + j], kD = kq[16 + j];
          const float* a0 = reinterpret_cast<const float*>(&ksc); const float* a1 = reinterpret_cast<const float*>(&ksh);
          const float* a2 = reinterpret_cast<const float*>(&kA); const float* a3 = reinterpret_cast<const float*>(&kB);
          const float* a4 = reinterpret_cast<const float*>(&kD);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 4 * j + q;
            const float e = acc[i];
            const float yh = fmaf(e, a0[q], a1[q]);
            const float gm = (yh > 0.f && yh < 6.f) ? g[i] : 0.f;
            dye[i] = fmaf(gm, a2[q], fmaf(e, a3[q], a4[q]));
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = dye[8 * s + j];
        uintx4 db[3];
        split8(v8, db);
        accdx[t] = mfma6(wT[s], db, accdx[t]);                  // dx[ci 8*(i/4) + 4*lhi + (i%4)][pixel l31]
      }
    }
  }
  // ---------------- store
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (!pok[t]) continue;
    const int iy = r0 + 2 * wave + t;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ci = 8 * (i >> 2) + 4 * lhi + (i & 3);
      if (ci < Cin) {
        const size_t idx = ((size_t)n * Cin + ci) * HW + (size_t)iy * W + ix;
        float o = accdx[t][i];
        if (p.add0) o += p.add0[idx];
        if (p.accum) o += p.dx[idx];
        p.dx[idx] = o;
      }
    }
  }
}

// =================================================================================================================================
// finalize: M = sum of the rows' [33][32] blocks (fp64), then dW_e[h][ci] = A_h G[h][ci] + B_h sum_k W_e[h][k] M[k][ci] + D_h s[ci]
__global__ __launch_bounds__(256) void k_irt_msum(const float* __restrict__ mpart, int rows, double* __restrict__ mfin) {
  __shared__ double s_t[256];
  const int r = blockIdx.x, ci = threadIdx.x & 31, sl = threadIdx.x >> 5;
  double v = 0.0;
  for (int k = sl; k < rows; k += 8) v += (double)mpart[((size_t)k * 33 + r) * 32 + ci];
  s_t[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x < 32) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += s_t[k * 32 + threadIdx.x];
    mfin[r * 32 + threadIdx.x] = t;
  }
}
__global__ __launch_bounds__(256) void k_irt_dwe(const float* __restrict__ gpart, int rows, int HdP, const double* __restrict__ mfin,
                                                 const float* __restrict__ we, const float* __restrict__ cstb, float* __restrict__ dw,
                                                 int Hd, int Cin) {
  __shared__ double s_t[256];
  const int h = blockIdx.x, ci = threadIdx.x & 31, sl = threadIdx.x >> 5;
  double v = 0.0;
  for (int k = sl; k < rows; k += 8) v += (double)gpart[((size_t)k * HdP + h) * 32 + ci];
  s_t[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x < 32 && ci < Cin) {
    double G = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) G += s_t[k * 32 + ci];
    double wm = 0.0;
    for (int k = 0; k < Cin; ++k) wm += (double)we[(size_t)h * Cin + k] * mfin[k * 32 + ci];
    const double A = cstb[(size_t)h * SC_CST + 2], Bc = cstb[(size_t)h * SC_CST + 3], D = cstb[(size_t)h * SC_CST + 4];
    dw[(size_t)h * Cin + ci] = (float)(A * G + Bc * wm + D * mfin[32 * 32 + ci]);
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
bool irt_ok(int Cin, int Hd, int H, int W, int S) {
  return (S == 1 || S == 2) && Cin >= 8 && Cin <= 32 && Cin % 8 == 0 && Hd >= 8 && Hd <= 32 * IRT_MAXCH && H >= 4 && W >= 8 && H % 4 == 0 &&
         W % 8 == 0 && (S == 1 || (H % 2 == 0 && W % 2 == 0));
}
int irt_stat_wgs(long npb) { const long w = (npb + 3) / 4; return (int)(w < 512 ? w : 512); }
int irt_fwd_tiles(int N, int H, int W, int S, int* tx, int* ty) {
  const int Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
  const int TH = S == 1 ? 8 : 4, TW = S == 1 ? 32 : 16;
  *tx = (Wo + TW - 1) / TW; *ty = (Ho + TH - 1) / TH;
  return N * *tx * *ty;
}
int irt_bwd_tiles(int N, int H, int W, int* tx, int* ty) {
  *tx = (W + 31) / 32; *ty = (H + 7) / 8;
  return N * *tx * *ty;
}
int irt_bsum_per_wg(int ntiles, int nch) {
  // about two work-groups per CU over both grid dimensions, at least 2 tiles each
  int want = (512 + nch - 1) / nch;
  int per = (ntiles + want - 1) / want;
  return per < 2 ? 2 : per;
}

int irt_fill(IrtP& p, const sc_irt_args* a, const char* who) {
  SC_REQUIRE(a && a->x.x && a->w_expand && a->w_dw && a->cst_expand, "%s: null argument", who);
  SC_REQUIRE(irt_ok(a->Cin, a->hidden, a->H, a->W, a->stride), "%s: unsupported block (Cin %d, hidden %d, %dx%d, stride %d)", who, a->Cin,
             a->hidden, a->H, a->W, a->stride);
  SC_REQUIRE(a->N > 0 && a->x.C == a->Cin && a->x.up == 0 && (a->x.mode == SC_SRC_RAW || (a->x.mode == SC_SRC_AFFINE && a->x.cst)),
             "%s: the block input must be a RAW or AFFINE source of Cin channels", who);
  p.x = to_srcd(a->x);
  if (p.x.mode == SC_SRC_RAW) {
    p.x.cst = sc_identity_cst_table(p.x.C);
    p.x.act = SC_ACT_NONE;
    SC_REQUIRE(p.x.cst != nullptr, "%s: identity constants unavailable", who);
  }
  p.we = a->w_expand; p.wd = a->w_dw; p.cst_e = a->cst_expand; p.cstb_e = nullptr;
  p.dy = empty_srcd();
  p.stats = nullptr; p.dout = nullptr; p.esums = nullptr; p.dwacc = nullptr; p.gpart = nullptr; p.mpart = nullptr;
  p.dx = nullptr; p.add0 = nullptr; p.accum = 0;
  p.N = a->N; p.Cin = a->Cin; p.Hd = a->hidden; p.H = a->H; p.W = a->W; p.S = a->stride;
  p.Ho = (a->H - 1) / a->stride + 1; p.Wo = (a->W - 1) / a->stride + 1;
  p.nch = (a->hidden + 31) / 32;
  p.tiles_x = p.tiles_y = p.ntiles = p.tiles_per_wg = 0;
  const long NP = (long)a->N * a->H * a->W;
  SC_REQUIRE(NP * a->hidden < (1L << 31), "%s: tensor too large for 32-bit element counts", who);
  p.npb = (int)((NP + 31) / 32);
  return SC_OK;
}
int irt_dy(IrtP& p, const sc_src* dy, const char* who) {
  SC_REQUIRE(dy && dy->x && dy->C == p.Hd && dy->up == 0, "%s: dy must be a source of `hidden` channels at the depthwise output's size", who);
  SC_REQUIRE(dy->mode == SC_SRC_RAW || (dy->mode == SC_SRC_BNBWD && dy->aux && dy->cst), "%s: dy must be a RAW or BNBWD source", who);
  p.dy = to_srcd(*dy);
  return SC_OK;
}
template <typename K>
void irt_lds_attr(K kern, size_t lds) {
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace

// =================================================================================================================================
extern "C" int sc_irt_supported(int Cin, int hidden, int H, int W, int stride) { return irt_ok(Cin, hidden, H, W, stride) ? 1 : 0; }

extern "C" int sc_irt_rows(int stage, int N, int H, int W, int stride) {
  int tx, ty;
  if (stage == 0) return irt_stat_wgs(((long)N * H * W + 31) / 32);
  if (stage == 1) return irt_fwd_tiles(N, H, W, stride, &tx, &ty);
  return -1;
}
// rows of stage 2 depend on the chunk count (the grid is tile sets x chunks)
static int irt_bsum_rows(int N, int H, int W, int hidden, int* per_out) {
  int tx, ty;
  const int nt = irt_bwd_tiles(N, H, W, &tx, &ty);
  const int per = irt_bsum_per_wg(nt, (hidden + 31) / 32);
  if (per_out) *per_out = per;
  return (nt + per - 1) / per;
}
extern "C" int sc_irt_bwd_rows(int N, int hidden, int H, int W) { return irt_bsum_rows(N, H, W, hidden, nullptr); }

extern "C" size_t sc_irt_bwd_workspace_floats(int N, int hidden, int H, int W) {
  const size_t rows = (size_t)irt_bsum_rows(N, H, W, hidden, nullptr);
  const size_t HdP = (size_t)((hidden + 31) / 32) * 32;
  return rows * HdP * 32 + rows * 33 * 32 + 2 * 33 * 32 + 64;
}

extern "C" int sc_irt_expand_stats(const sc_irt_args* a, float* stats, sc_stream stream) {
  IrtP p;
  if (int rc = irt_fill(p, a, "sc_irt_expand_stats")) return rc;
  SC_REQUIRE(stats, "sc_irt_expand_stats: null stats");
  p.stats = stats;
  const int nks = (p.Cin + 15) / 16;
  const int wgs = irt_stat_wgs(p.npb);
  const size_t lds = (size_t)p.nch * nks * 3 * 64 * 16 + 64 * 4 + (size_t)4 * p.nch * 32 * 2 * 4;
  hipStream_t st = (hipStream_t)stream;
  if (nks == 1) { irt_lds_attr(&k_irt_stats<1>, lds); hipLaunchKernelGGL((k_irt_stats<1>), dim3(wgs), dim3(256), lds, st, p); }
  else { irt_lds_attr(&k_irt_stats<2>, lds); hipLaunchKernelGGL((k_irt_stats<2>), dim3(wgs), dim3(256), lds, st, p); }
  SC_LAUNCH_OK("sc_irt_expand_stats");
  return SC_OK;
}

extern "C" int sc_irt_fwd(const sc_irt_args* a, float* d_out, float* stats_d, sc_stream stream) {
  IrtP p;
  if (int rc = irt_fill(p, a, "sc_irt_fwd")) return rc;
  SC_REQUIRE(d_out && stats_d, "sc_irt_fwd: null output");
  p.dout = d_out; p.stats = stats_d;
  p.ntiles = irt_fwd_tiles(p.N, p.H, p.W, p.S, &p.tiles_x, &p.tiles_y);
  const int nks = (p.Cin + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(32 * (p.S == 1 ? IrtFwdGeo<1>::EPAD : IrtFwdGeo<2>::EPAD) + 64) * 4;
#define SC_IRT_F(NK, SS) { irt_lds_attr(&k_irt_fwd<NK, SS>, lds); hipLaunchKernelGGL((k_irt_fwd<NK, SS>), dim3(p.ntiles), dim3(256), lds, st, p); }
  if (nks == 1) { if (p.S == 1) SC_IRT_F(1, 1) else SC_IRT_F(1, 2) }
  else { if (p.S == 1) SC_IRT_F(2, 1) else SC_IRT_F(2, 2) }
#undef SC_IRT_F
  SC_LAUNCH_OK("sc_irt_fwd");
  return SC_OK;
}

extern "C" int sc_irt_bwd_sums(const sc_irt_args* a, const sc_src* dy_d, double* e_sums, double* dw_acc, float* work, sc_stream stream) {
  IrtP p;
  if (int rc = irt_fill(p, a, "sc_irt_bwd_sums")) return rc;
  if (int rc = irt_dy(p, dy_d, "sc_irt_bwd_sums")) return rc;
  SC_REQUIRE(e_sums && dw_acc && work, "sc_irt_bwd_sums: null output");
  p.ntiles = irt_bwd_tiles(p.N, p.H, p.W, &p.tiles_x, &p.tiles_y);
  const int rows = irt_bsum_rows(p.N, p.H, p.W, p.Hd, &p.tiles_per_wg);
  p.esums = e_sums; p.dwacc = dw_acc;
  p.gpart = work;
  p.mpart = work + (size_t)rows * p.nch * 32 * 32;
  const int nks = (p.Cin + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(32 * (p.S == 1 ? IrtBwdGeo<1>::PITCH : IrtBwdGeo<2>::PITCH) + 256 + 64 + 3 * 16 * 64) * 4;
  const dim3 grid(rows, p.nch);
#define SC_IRT_S(NK, SS) { irt_lds_attr(&k_irt_bsums<NK, SS>, lds); hipLaunchKernelGGL((k_irt_bsums<NK, SS>), grid, dim3(256), lds, st, p); }
  if (nks == 1) { if (p.S == 1) SC_IRT_S(1, 1) else SC_IRT_S(1, 2) }
  else { if (p.S == 1) SC_IRT_S(2, 1) else SC_IRT_S(2, 2) }
#undef SC_IRT_S
  SC_LAUNCH_OK("sc_irt_bwd_sums");
  return SC_OK;
}

extern "C" int sc_irt_bwd_data(const sc_irt_args* a, const sc_src* dy_d, const float* cst_bwd_expand, float* dx, const float* add0, int accum,
                               sc_stream stream) {
  IrtP p;
  if (int rc = irt_fill(p, a, "sc_irt_bwd_data")) return rc;
  if (int rc = irt_dy(p, dy_d, "sc_irt_bwd_data")) return rc;
  SC_REQUIRE(cst_bwd_expand && dx, "sc_irt_bwd_data: null argument");
  p.cstb_e = cst_bwd_expand; p.dx = dx; p.add0 = add0; p.accum = accum;
  p.ntiles = irt_bwd_tiles(p.N, p.H, p.W, &p.tiles_x, &p.tiles_y);
  const int nks = (p.Cin + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(32 * (p.S == 1 ? IrtBwdGeo<1>::PITCH : IrtBwdGeo<2>::PITCH) + 256 + 64 + 160 + 288) * 4;
#define SC_IRT_D(NK, SS) { irt_lds_attr(&k_irt_bdata<NK, SS>, lds); hipLaunchKernelGGL((k_irt_bdata<NK, SS>), dim3(p.ntiles), dim3(256), lds, st, p); }
  if (nks == 1) { if (p.S == 1) SC_IRT_D(1, 1) else SC_IRT_D(1, 2) }
  else { if (p.S == 1) SC_IRT_D(2, 1) else SC_IRT_D(2, 2) }
#undef SC_IRT_D
  SC_LAUNCH_OK("sc_irt_bwd_data");
  return SC_OK;
}

extern "C" int sc_irt_wgrad_finalize(const sc_irt_args* a, const float* cst_bwd_expand, float* work, float* dw_expand, sc_stream stream) {
  SC_REQUIRE(a && cst_bwd_expand && work && dw_expand && a->w_expand, "sc_irt_wgrad_finalize: null argument");
  SC_REQUIRE(irt_ok(a->Cin, a->hidden, a->H, a->W, a->stride), "sc_irt_wgrad_finalize: unsupported block");
  const int rows = irt_bsum_rows(a->N, a->H, a->W, a->hidden, nullptr);
  const int nch = (a->hidden + 31) / 32;
  const float* gpart = work;
  const float* mpart = work + (size_t)rows * nch * 32 * 32;
  // 8-byte aligned fp64 scratch behind the partial rows
  uintptr_t mf = reinterpret_cast<uintptr_t>(mpart + (size_t)rows * 33 * 32);
  mf = (mf + 7) & ~(uintptr_t)7;
  double* mfin = reinterpret_cast<double*>(mf);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_irt_msum, dim3(33), dim3(256), 0, st, mpart, rows, mfin);
  hipLaunchKernelGGL(k_irt_dwe, dim3(a->hidden), dim3(256), 0, st, gpart, rows, nch * 32, mfin, a->w_expand, cst_bwd_expand, dw_expand, a->hidden,
                     a->Cin);
  SC_LAUNCH_OK("sc_irt_wgrad_finalize");
  return SC_OK;
}
