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
#include <stdlib.h>

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
// two independent six-product blocks issued alternately: a chain of dependent MFMAs on ONE accumulator runs at the instruction's
// latency (16 passes), two interleaved chains at its issue rate
__device__ __forceinline__ void mfma6x2(const uintx4 (&a)[3], const uintx4 (&b)[3], floatx16& c, const uintx4 (&d)[3], const uintx4 (&e)[3],
                                        floatx16& f) {
  c = mfma_bf16(a[1], b[1], c); f = mfma_bf16(d[1], e[1], f);
  c = mfma_bf16(a[2], b[0], c); f = mfma_bf16(d[2], e[0], f);
  c = mfma_bf16(a[0], b[2], c); f = mfma_bf16(d[0], e[2], f);
  c = mfma_bf16(a[1], b[0], c); f = mfma_bf16(d[1], e[0], f);
  c = mfma_bf16(a[0], b[1], c); f = mfma_bf16(d[0], e[1], f);
  c = mfma_bf16(a[0], b[0], c); f = mfma_bf16(d[0], e[0], f);
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
  float* dx; const float* add0; int accum;                                                            // (k_irt_bwd: partials; k_irt_fix)
  const uintx4* wpk;       // packed split filter operands (k_irt_pack)                                  (k_irt_bwd)
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
  float* s_kc = s_xc + 64;                                  // [nch*32][12]: BN_e scale, shift, depthwise taps [9], pad
  uintx4* s_w = reinterpret_cast<uintx4*>(s_kc + (size_t)p.nch * 32 * 12);      // [nch][NKS][3][64] split filter operands
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, W = p.W, Hd = p.Hd;
  const size_t HW = (size_t)H * W, HWo = (size_t)p.Ho * p.Wo;
  // ---------------- once per (persistent) work-group: the tables of every chunk.  (The first version did this per output tile:
  // 4096 tiles x 3 chunks of filter loads + splits and scalar loads of the taps in the stencil loop: 140 us on features.2.)
  for (int ch = wave; ch < p.nch; ch += 4) {
    uintx4 op[NKS][3];
    irt_wop<NKS>(p, ch * 32 + l31, lhi, op);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int t = 0; t < 3; ++t) s_w[((size_t)(ch * NKS + ks) * 3 + t) * 64 + lane] = op[ks][t];
  }
  for (int i = tid; i < p.nch * 32; i += 256) {
    const bool ok = i < Hd;
    const int hc = ok ? i : Hd - 1;
    float* k = s_kc + i * 12;
    k[0] = p.cst_e[(size_t)hc * SC_CST]; k[1] = p.cst_e[(size_t)hc * SC_CST + 1];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[2 + t] = ok ? p.wd[(size_t)hc * 9 + t] : 0.f;
    k[11] = 0.f;
  }
  irt_xconsts(p, s_xc);
  __syncthreads();
  const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
  const int lox = lane % TW, lrg = lane / TW;                 // phase 2: column, row group of this lane's output pixels
  const int oyl = lrg * RPL;

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int n = tile / (p.tiles_x * p.tiles_y), tt = tile - n * p.tiles_x * p.tiles_y;
    const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
    const int ey0 = oy0 * S - 1, ex0 = ox0 * S - 1;
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
      for (int j = 0; j < 4; ++j) {
        // accumulator registers 4 j .. 4 j + 3 = four consecutive flattened pixels: one division, then a walk along the row
        const int f0 = blk * 32 + 8 * j + 4 * lhi;
        int eyi = f0 / EW, exi = f0 - eyi * EW;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int yi = ey0 + eyi, xi = ex0 + exi;
          if (f0 + q < EPX && yi >= 0 && yi < H && xi >= 0 && xi < W) bits |= 1u << (4 * j + q);
          if (++exi == EW) { exi = 0; ++eyi; }
        }
      }
      inside[b] = bits;
    }
    const bool colok = ox0 + lox < p.Wo;
    int zt = 0;                                               // opaque zero: keeps the tile-invariant LDS table reads inside the loop
    asm volatile("" : "+v"(zt));                              // (hoisted, they pin a register set per chunk: 219 registers)

    for (int chunk = 0; chunk < p.nch; ++chunk) {
      // ---------------- phase 1
      {
        const float2 cs = *reinterpret_cast<const float2*>(s_kc + (chunk * 32 + l31) * 12 + zt);
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
          const int blk = wave + 4 * b;
          if (blk < NBLK) {
            floatx16 acc = zero16();
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
              uintx4 w[3];
#pragma unroll
              for (int t = 0; t < 3; ++t) w[t] = s_w[((chunk * NKS + ks) * 3 + t) * 64 + lane + zt];
              acc = mfma6(xop[b][ks], w, acc);               // rows = pixels, columns = hidden channels
            }
            float* dst = s_e + l31 * EPAD + blk * 32 + 4 * lhi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float4 v;
              float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float t = fminf(fmaxf(fmaf(acc[4 * j + q], cs.x, cs.y), 0.f), 6.f);
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
          {
            const float* kq = s_kc + (chunk * 32 + cl) * 12 + 2 + zt;      // broadcast reads
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = kq[k];
          }
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
          if (p.stats) {
            const float t1 = wave_total(s1), t2 = wave_total(s2);
            if (lane == 0) *reinterpret_cast<float2*>(p.stats + ((size_t)tile * Hd + h) * 2) = make_float2(t1, t2);
          }
        }
      }
      __syncthreads();
    }
  }
}

// =================================================================================================================================
// (C) backward, stride 2, ONE heavy sweep.  dy_e = scale (g' - c1 - c2 xn) is affine in (g', e) per channel, and its coefficient
// of g' -- the BatchNorm scale -- is known BEFORE the batch sums c1, c2 are, so
//     dx = W_e^T dy_e = W_e^T (scale (.) g')  +  Q x + r,      Q = W_e^T diag(B) W_e  (Cin x Cin),  r = W_e^T D
// and the sweep that forms g' can emit the first term at once; the remainder is a Cin -> Cin pointwise fix-up of the small tensor
// (k_irt_fix) once sc_bn_bwd_finalize has produced B, D.  e, g_e and dy_e never exist in memory, and (dy_d, d) are read ONCE.
//
// Work-group = 512 threads = 8 waves on a tile of 8 x 32 e pixels; wave w owns the 4 x 8 pixel block (w / 4, w % 4) -- orientation
// 1: lane = hidden channel, registers = a 4 x 4 pixel patch, so BN / ReLU6 switch / the stride-2 stencil (9 LDS reads per patch, tap
// sets fixed by pixel parity at compile time) / all per-channel sums are in-lane.  A work-group walks its tiles and up to 3 chunks
// of 32 hidden channels (grid.y = chunk groups; a second group writes its own dx partial, the fix-up adds them):
//   per tile:   stage (dy_d of all its chunks: 5 x 17 per channel, every load issued before the first is used) | barrier
//   per chunk:  e = x W_e (MFMA) -> g' -> sums, depthwise filter gradient, G += g' x^T (MFMA, K = pixels)
//               scale (.) g' -> wave-private LDS transpose -> B operand (K = hidden) -> dx += W_e^T (.) (MFMA)
constexpr int IRT_CG = 3;                  // chunks per work-group
constexpr int IRT_RW = 17, IRT_RSZ = 51;   // staged region of d per channel for a tile of 4 x 32 e pixels: 3 x 17 (odd pitch: lanes = channels read conflict-free)

struct IrtBwdLds {                         // byte offsets into dynamic LDS
  int k, dc, xc, dy, T, acc, total;
};
__host__ __device__ inline IrtBwdLds irt_bwd_lds() {
  // 65.7 KB: two work-groups per CU.  The split filter operands (27-36 KB per group) are read from a packed copy in global memory
  // instead (k_irt_pack: 16 bytes per lane and operand, L1 / L2 resident).  Measured on features.2: every table in LDS, one
  // work-group per CU: 412-452 us; this layout: 347 us; the dx operands back in LDS + e operands requested a chunk ahead (79.7 KB,
  // 20 spilled registers): 432 us.
  IrtBwdLds l;
  l.k = 0;                                        // [CG][32][16] float: scale, shift, mean, invstd, wd[9], pad
  l.dc = l.k + IRT_CG * 32 * 16 * 4;              // [CG][32][8] float
  l.xc = l.dc + IRT_CG * 32 * 8 * 4;              // [32][2] float
  l.dy = l.xc + 64 * 4;                           // [CG][32][51] float
  l.T = l.dy + IRT_CG * 32 * IRT_RSZ * 4;         // [4][32][36] float (also the epilogue's reduction scratch: [3][16][64])
  l.acc = l.T + 4 * 32 * 36 * 4;                  // [4][CG*32][12] float: per-wave accumulators of sum g', sum g' xn, dW_dw[9]
  l.total = l.acc + 4 * IRT_CG * 32 * 12 * 4;
  return l;
}

template <int NKS>
__global__ __launch_bounds__(256, 2) void k_irt_bwd(const IrtP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IrtBwdLds L_ = irt_bwd_lds();
  float* s_k = reinterpret_cast<float*>(smem + L_.k);
  float* s_dc = reinterpret_cast<float*>(smem + L_.dc);
  float* s_xc = reinterpret_cast<float*>(smem + L_.xc);
  float* s_dy = reinterpret_cast<float*>(smem + L_.dy);
  float* s_T = reinterpret_cast<float*>(smem + L_.T);
  float* s_acc = reinterpret_cast<float*>(smem + L_.acc);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, W = p.W, Hd = p.Hd, Cin = p.Cin;
  const size_t HW = (size_t)H * W, HWo = (size_t)p.Ho * p.Wo;
  const int ch0 = blockIdx.y * IRT_CG;
  const int ncg = min(IRT_CG, p.nch - ch0);
  // ---------------- once per work-group: filter operands and per-channel constants of its chunks
  const uintx4* g_w = p.wpk + (size_t)ch0 * NKS * 3 * 64 + lane;                      // [nch][NKS][3][64]
  const uintx4* g_wT = p.wpk + (size_t)p.nch * NKS * 3 * 64 + (size_t)ch0 * 2 * 3 * 64 + lane;   // [nch][2][3][64]
  for (int i = tid; i < IRT_CG * 32; i += 256) {
    const int hh = ch0 * 32 + i;
    const bool ok = hh < Hd && i < ncg * 32;
    const int hc = hh < Hd ? hh : Hd - 1;
    float* k = s_k + i * 16;
    const float4 ce = *reinterpret_cast<const float4*>(p.cst_e + (size_t)hc * SC_CST);
    k[0] = ce.x; k[1] = ce.y; k[2] = ce.z; k[3] = ce.w;
#pragma unroll
    for (int t = 0; t < 9; ++t) k[4 + t] = ok ? p.wd[(size_t)hc * 9 + t] : 0.f;
    k[13] = 0.f; k[14] = 0.f; k[15] = 0.f;
    float4 c0 = make_float4(1.f, 0.f, 1.f, 0.f); float c4 = 0.f;
    if (p.dy.mode == SC_SRC_BNBWD) { c0 = *reinterpret_cast<const float4*>(p.dy.cst + (size_t)hc * SC_CST); c4 = p.dy.cst[(size_t)hc * SC_CST + 4]; }
    *reinterpret_cast<float4*>(s_dc + i * 8) = c0;
    *reinterpret_cast<float4*>(s_dc + i * 8 + 4) = make_float4(c4, 0.f, 0.f, 0.f);
  }
  for (int i = tid; i < 4 * IRT_CG * 32 * 12; i += 256) s_acc[i] = 0.f;
  irt_xconsts(p, s_xc);
  const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
  const float dlo = sc_act_lo(p.dy.act), dhi = sc_act_hi(p.dy.act);
  const bool bnb = p.dy.mode == SC_SRC_BNBWD;
  const bool ciok = l31 < Cin;
  const int cic = ciok ? l31 : Cin - 1;
  const float xsc = p.x.cst[(size_t)cic * SC_CST], xsh = p.x.cst[(size_t)cic * SC_CST + 1];
  const int bc = wave;                                       // this wave's 4 x 8 pixel block of the 4 x 32 tile
  float* myT = s_T + wave * (32 * 36);
  float* myacc = s_acc + wave * (IRT_CG * 32 * 12);
  // staging: a lane owns ONE position (ry, rx) of the 3 x 17 region, a wave walks channels wave, wave + 4, ... of the group
  const int sry = lane / IRT_RW, srx = lane - sry * IRT_RW;
  const bool slane = lane < IRT_RSZ;
  const int nst = ncg * 8;                                   // channels per wave

  floatx16 accG[IRT_CG];
#pragma unroll
  for (int cq = 0; cq < IRT_CG; ++cq) accG[cq] = zero16();
  const int t_begin = blockIdx.x * p.tiles_per_wg;
  const int t_end = min(t_begin + p.tiles_per_wg, p.ntiles);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int n = tile / (p.tiles_x * p.tiles_y), tt = tile - n * p.tiles_x * p.tiles_y;
    const int r0 = (tt / p.tiles_x) * 4, c0 = (tt % p.tiles_x) * 32;
    __syncthreads();                                          // the previous tile's reads of s_dy are done (first pass: the tables are written)
    // ---------------- stage dy_d of every chunk of the group (8 loads per batch in flight, then the prologue and the LDS stores)
    {
      const int oy = (r0 >> 1) + sry, ox = (c0 >> 1) + srx;
      const bool inb = slane && oy < p.Ho && ox < p.Wo;
      const size_t pos = inb ? (size_t)oy * p.Wo + ox : 0;
      const float* gsrc = p.dy.x + ((size_t)n * Hd + ch0 * 32) * HWo + pos;
      const float* ysrc = bnb ? p.dy.aux + ((size_t)n * Hd + ch0 * 32) * HWo + pos : gsrc;
      const int hrem = Hd - ch0 * 32;                         // channels of the group that exist
      // every load of the tile in flight before the first is used (one memory round trip; 3 batches of 8 measured 14 us per tile)
      float gv[IRT_CG * 8], yv[IRT_CG * 8];
#pragma unroll
      for (int k = 0; k < IRT_CG * 8; ++k) {
        const int cl = wave + 4 * k;
        const size_t o = (size_t)((k < nst && cl < hrem) ? cl : 0) * HWo;
        gv[k] = gsrc[o];
        yv[k] = ysrc[o];
      }
#pragma unroll
      for (int k = 0; k < IRT_CG * 8; ++k) {
        const int cl = wave + 4 * k;
        float v = gv[k];
        if (bnb) {
          const float4 dc = *reinterpret_cast<const float4*>(s_dc + cl * 8);
          v = sc_pro_bnbwd(gv[k], yv[k], dc.x, dc.y, dc.z, dc.w, s_dc[cl * 8 + 4], dlo, dhi);
        }
        if (slane && k < nst) s_dy[cl * IRT_RSZ + lane] = (inb && cl < hrem) ? v : 0.f;
      }
    }
    // ---------------- operands of this wave's pixel block
    const int pr = r0, pc0 = c0 + 8 * bc;
    const bool bok = pr < H && pc0 < W;                       // H % 4 == 0, W % 8 == 0: inside or outside as a whole
    uintx4 xop[NKS][3], xT[2][3];
    {
      const int y = pr + (l31 >> 3), x = pc0 + (l31 & 7);
      irt_xop<NKS>(p, s_xc, p.x.x + (size_t)n * Cin * HW + (bok ? (size_t)y * W + x : 0), bok, lhi, lo, hi, HW, xop);
#pragma unroll
      for (int sstep = 0; sstep < 2; ++sstep) {
        float v[8];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const float4 q4 = *reinterpret_cast<const float4*>(p.x.x + ((size_t)n * Cin + cic) * HW +
                                                             (bok ? (size_t)(pr + 2 * sstep + rr) * W + pc0 + 4 * lhi : 0));
          const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float tv = sc_pro_affine(qv[q], xsc, xsh, lo, hi);
            v[4 * rr + q] = (bok && ciok) ? tv : 0.f;
          }
        }
        split8(v, xT[sstep]);
      }
    }
    __syncthreads();
    floatx16 accdx = zero16();
    if (bok) {
#pragma unroll
      for (int cq = 0; cq < IRT_CG; ++cq) {
        if (cq < ncg) {
          // an opaque zero in every table index: these LDS reads are invariant over the tile loop, and hoisted out of it they pin
          // ~50 registers per chunk for the whole kernel -- they must be re-read where they are used
          int z = 0;
          asm volatile("" : "+v"(z));
          floatx16 acc = zero16();
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            uintx4 w[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) w[t] = g_w[((cq * NKS + ks) * 3 + t) * 64 + z];
            acc = mfma6(xop[ks], w, acc);                     // e[pixel (i/4, 4 lhi + i%4) of the block][channel l31]
          }
          const float4* kq = reinterpret_cast<const float4*>(s_k + (cq * 32 + l31) * 16 + z);
          const float4 ce = kq[0];                            // scale, shift, mean, invstd
          float wk[9];
          {
            const float4 a = kq[1], b = kq[2]; const float c = s_k[(cq * 32 + l31) * 16 + 12 + z];
            wk[0] = a.x; wk[1] = a.y; wk[2] = a.z; wk[3] = a.w; wk[4] = b.x; wk[5] = b.y; wk[6] = b.z; wk[7] = b.w; wk[8] = c;
          }
          float Lr[3][3];
          {
            const float* base = s_dy + (cq * 32 + l31) * IRT_RSZ + 4 * bc + 2 * lhi + z;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int q = 0; q < 3; ++q) Lr[a][q] = base[a * IRT_RW + q];
          }
          float gq[16], dwd[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) dwd[k] = 0.f;
          float t1 = 0.f, t2 = 0.f;
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
                for (int kx = 0; kx < 3; ++kx)
                  if (((u + 1 - ky) & 1) == 0 && ((v + 1 - kx) & 1) == 0) {      // output pixel (i + 1 - k) / 2 exists for this tap
                    const float d = Lr[(u + 1 - ky) / 2][(v + 1 - kx) / 2];
                    g = fmaf(wk[ky * 3 + kx], d, g);
                    dwd[ky * 3 + kx] = fmaf(eh, d, dwd[ky * 3 + kx]);
                  }
              const float gm = pass ? g : 0.f;
              t1 += gm;
              t2 = fmaf(gm, xn, t2);
              gq[4 * u + v] = gm;
              myT[(8 * u + 4 * lhi + v) * 36 + l31] = gm * ce.x;          // scale (.) g'  ->  T[pixel][channel]
            }
          // per-channel scalars of this block: both half-waves hold the same channel -> add the halves, lane < 32 accumulates into
          // the wave's own LDS row (no other wave touches it: deterministic order)
          {
            float* arow = myacc + (cq * 32 + l31) * 12;
            t1 += __shfl_xor(t1, 32, 64);
            t2 += __shfl_xor(t2, 32, 64);
#pragma unroll
            for (int k = 0; k < 9; ++k) dwd[k] += __shfl_xor(dwd[k], 32, 64);
            if (lhi == 0) {
              float4 a0 = *reinterpret_cast<float4*>(arow), a1 = *reinterpret_cast<float4*>(arow + 4), a2 = *reinterpret_cast<float4*>(arow + 8);
              a0.x += t1; a0.y += t2; a0.z += dwd[0]; a0.w += dwd[1];
              a1.x += dwd[2]; a1.y += dwd[3]; a1.z += dwd[4]; a1.w += dwd[5];
              a2.x += dwd[6]; a2.y += dwd[7]; a2.z += dwd[8];
              *reinterpret_cast<float4*>(arow) = a0; *reinterpret_cast<float4*>(arow + 4) = a1; *reinterpret_cast<float4*>(arow + 8) = a2;
            }
          }
          // G[h][ci] += sum over the block's pixels of g'[h][px] x[ci][px]   (A = g': rows = channels, k = pixels), and
          // dx[ci][px] += sum_h W_e[h][ci] (scale g')[h][px]: the transposed tile as the B operand (lane = pixel, k = channels);
          // the two contractions accumulate into different registers and are issued alternately
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int sstep = 0; sstep < 2; ++sstep) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = gq[8 * sstep + j];
            uintx4 ga[3];
            split8(v8, ga);
            const float4 a = *reinterpret_cast<const float4*>(myT + l31 * 36 + 16 * sstep + 8 * lhi);
            const float4 b = *reinterpret_cast<const float4*>(myT + l31 * 36 + 16 * sstep + 8 * lhi + 4);
            const float t8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uintx4 gb[3], wt[3];
            split8(t8, gb);
#pragma unroll
            for (int t = 0; t < 3; ++t) wt[t] = g_wT[((cq * 2 + sstep) * 3 + t) * 64 + z];
            mfma6x2(ga, xT[sstep], accG[cq], wt, gb, accdx);   // dx[ci 8 (i/4) + 4 lhi + i%4][pixel l31 of the block]
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
      // ---------------- this group's dx partial of the block: lane = pixel (row l31 / 8, column l31 % 8)
      float* dst = p.dx + (size_t)blockIdx.y * p.N * Cin * HW + (size_t)n * Cin * HW + (size_t)(pr + (l31 >> 3)) * W + pc0 + (l31 & 7);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = 8 * (i >> 2) + 4 * lhi + (i & 3);
        if (ci < Cin) dst[(size_t)ci * HW] = accdx[i];
      }
    }
  }
  // ---------------- epilogue: the work-group's partial rows
  __syncthreads();
  if (tid < ncg * 32 && ch0 * 32 + tid < Hd) {
    float v[11];
#pragma unroll
    for (int k = 0; k < 11; ++k)
      v[k] = ((s_acc[tid * 12 + k] + s_acc[(IRT_CG * 32 + tid) * 12 + k]) + s_acc[(2 * IRT_CG * 32 + tid) * 12 + k]) + s_acc[(3 * IRT_CG * 32 + tid) * 12 + k];
    const int hh = ch0 * 32 + tid;
    p.esums[((size_t)blockIdx.x * Hd + hh) * 2] = (double)v[0];
    p.esums[((size_t)blockIdx.x * Hd + hh) * 2 + 1] = (double)v[1];
#pragma unroll
    for (int k = 0; k < 9; ++k) atomicAdd(p.dwacc + (size_t)hh * 9 + k, (double)v[2 + k]);
  }
  float* s_red = s_T;
#pragma unroll
  for (int cq = 0; cq < IRT_CG; ++cq) {
    if (cq < ncg) {
      __syncthreads();
      if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_red[((wave - 1) * 16 + r) * 64 + lane] = accG[cq][r];
      }
      __syncthreads();
      if (wave == 0) {
        float* part = p.gpart + ((size_t)blockIdx.x * p.nch + ch0 + cq) * 32 * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = ((accG[cq][r] + s_red[r * 64 + lane]) + s_red[(16 + r) * 64 + lane]) + s_red[(32 + r) * 64 + lane];
          part[(8 * (r >> 2) + 4 * lhi + (r & 3)) * 32 + l31] = v;
        }
      }
    }
  }
}

// =================================================================================================================================
// the split filter operands of k_irt_bwd, once per launch: block = chunk; waves 0..NKS-1: e operands (lane -> hidden channel, k =
// input channels), waves NKS..NKS+1: dx operands (lane -> input channel, k-slots of step s = hidden channels 16 s + 8 lhi + j)
template <int NKS>
__global__ __launch_bounds__(256) void k_irt_pack(const IrtP p, uintx4* __restrict__ wpk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int chunk = blockIdx.x;
  if (wave == 0) {
    uintx4 op[NKS][3];
    irt_wop<NKS>(p, chunk * 32 + l31, lhi, op);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int t = 0; t < 3; ++t) wpk[((size_t)(chunk * NKS + ks) * 3 + t) * 64 + lane] = op[ks][t];
  } else if (wave <= 2) {
    const int sstep = wave - 1;
    const bool ciok = l31 < p.Cin;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int hh = chunk * 32 + 16 * sstep + 8 * lhi + j;
      const float t = p.we[(size_t)(hh < p.Hd ? hh : p.Hd - 1) * p.Cin + (ciok ? l31 : 0)];
      v[j] = (hh < p.Hd && ciok) ? t : 0.f;
    }
    uintx4 op[3];
    split8(v, op);
#pragma unroll
    for (int t = 0; t < 3; ++t) wpk[(size_t)p.nch * NKS * 3 * 64 + ((size_t)(chunk * 2 + sstep) * 3 + t) * 64 + lane] = op[t];
  }
}

// =================================================================================================================================
// second moments of the block input: M[ci][ci'] = sum_px x[ci][px] x[ci'][px], s[ci] = sum_px x[ci][px]  (for dW_e, off the critical
// path).  K = pixels: lane -> channel l31, k-slots = 8 consecutive pixels (H*W % 8 == 0); wave-strided 16-pixel steps.
__global__ __launch_bounds__(256) void k_irt_xmom(const IrtP p) {
  extern __shared__ __attribute__((aligned(16))) float s_mr[];      // [3][16][64] + [4][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int Cin = p.Cin;
  const size_t HW = (size_t)p.H * p.W;
  const long NP = (long)p.N * (long)HW;
  const long ksteps = (NP + 15) / 16;
  const bool ciok = l31 < Cin;
  const int cic = ciok ? l31 : Cin - 1;
  const float xsc = p.x.cst[(size_t)cic * SC_CST], xsh = p.x.cst[(size_t)cic * SC_CST + 1];
  const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
  floatx16 accM = zero16();
  float sx = 0.f;
  const long per = (ksteps + gridDim.x - 1) / gridDim.x;
  const long k_begin = (long)blockIdx.x * per, k_end = min(k_begin + per, ksteps);
  // four 16-pixel steps per trip: eight 16-byte loads in flight per lane (one step per trip ran at 1.4 TB/s)
  for (long kq0 = k_begin + wave; kq0 < k_end; kq0 += 16) {
    float4 ra[4], rb[4];
    bool okv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long kq = kq0 + 4 * u;
      const long g0 = (kq * 2 + lhi) * 8;
      okv[u] = kq < k_end && g0 < NP;
      const long gc = okv[u] ? g0 : 0;
      const int n = (int)(gc / (long)HW);
      const float* src = p.x.x + ((size_t)n * Cin + cic) * HW + (size_t)(gc - (long)n * (long)HW);
      ra[u] = *reinterpret_cast<const float4*>(src);
      rb[u] = *reinterpret_cast<const float4*>(src + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float x8[8] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w, rb[u].x, rb[u].y, rb[u].z, rb[u].w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float t = sc_pro_affine(x8[j], xsc, xsh, lo, hi); v[j] = (okv[u] && ciok) ? t : 0.f; sx += v[j]; }
      uintx4 op[3];
      split8(v, op);
      accM = mfma6(op, op, accM);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_mr[((wave - 1) * 16 + r) * 64 + lane] = accM[r];
  }
  const float sh2 = sx + __shfl_xor(sx, 32, 64);
  if (lhi == 0) s_mr[3 * 16 * 64 + wave * 32 + l31] = sh2;
  __syncthreads();
  float* mp = p.mpart + (size_t)blockIdx.x * 33 * 32;
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = ((accM[r] + s_mr[r * 64 + lane]) + s_mr[(16 + r) * 64 + lane]) + s_mr[(32 + r) * 64 + lane];
      mp[(8 * (r >> 2) + 4 * lhi + (r & 3)) * 32 + l31] = v;
    }
  }
  if (tid < 32) {
    const float* q = s_mr + 3 * 16 * 64;
    mp[32 * 32 + tid] = ((q[tid] + q[32 + tid]) + q[64 + tid]) + q[96 + tid];
  }
}

// Q = W_e^T diag(B) W_e (row-major [ci'][ci], 32 x Cin), r = W_e^T D: block = ci', thread = (ci, slice of the hidden channels)
__global__ __launch_bounds__(256) void k_irt_qr(const float* __restrict__ we, const float* __restrict__ cstb, int Hd, int Cin, float* __restrict__ qr) {
  __shared__ double s_q[256], s_r[8];
  const int a = blockIdx.x, b = threadIdx.x & 31, sl = threadIdx.x >> 5;
  double q = 0.0, r = 0.0;
  if (b < Cin) {
    for (int h = sl; h < Hd; h += 8) {
      const double wa = we[(size_t)h * Cin + a];
      q += wa * (double)cstb[(size_t)h * SC_CST + 3] * (double)we[(size_t)h * Cin + b];
      if (b == 0) r += wa * (double)cstb[(size_t)h * SC_CST + 4];
    }
  }
  s_q[threadIdx.x] = q;
  if (b == 0) s_r[sl] = r;
  __syncthreads();
  if (threadIdx.x < 32 && b < Cin) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += s_q[k * 32 + b];
    qr[a * Cin + b] = (float)t;                                // laid out as a [Cin][Cin] filter for irt_wop
    if (b == 0) {
      double u = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) u += s_r[k];
      qr[32 * 32 + a] = (float)u;
    }
  }
}

// fix-up: dx = sum of the chunk groups' partials + Q x + r (+ add0) (+ dx): flat 32-pixel blocks, pixels on the accumulator rows
template <int NKS>
__global__ __launch_bounds__(256) void k_irt_fix(const IrtP p, const float* __restrict__ dxp, int ngroups, const float* __restrict__ qr) {
  __shared__ float s_xc[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int Cin = p.Cin;
  const size_t HW = (size_t)p.H * p.W;
  const long NP = (long)p.N * (long)HW;
  irt_xconsts(p, s_xc);
  __syncthreads();
  const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
  IrtP q = p;
  q.we = qr; q.Hd = Cin;                                     // the Q matrix as a Cin -> Cin pointwise filter
  uintx4 qop[NKS][3];
  irt_wop<NKS>(q, l31, lhi, qop);
  const float rr = l31 < Cin ? qr[32 * 32 + l31] : 0.f;
  const size_t gstride = (size_t)p.N * Cin * HW;
  for (int pb = blockIdx.x * 4 + wave; pb < p.npb; pb += gridDim.x * 4) {
    const long gp = (long)pb * 32 + l31;
    const bool ok = gp < NP;
    const long gpc = ok ? gp : 0;
    const int n = (int)(gpc / (long)HW);
    uintx4 xop[NKS][3];
    irt_xop<NKS>(p, s_xc, p.x.x + (size_t)n * Cin * HW + (size_t)(gpc - (long)n * (long)HW), ok, lhi, lo, hi, HW, xop);
    floatx16 acc = zero16();
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) acc = mfma6(xop[ks], qop[ks], acc);      // (Q x)[pixel 8 (i/4) + 4 lhi + i%4][ci' = l31]
    if (l31 < Cin) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long g = (long)pb * 32 + 8 * j + 4 * lhi;       // H*W % 4 == 0: four consecutive pixels of one image
        if (g >= NP) continue;
        const int n_ = (int)(g / (long)HW);
        const size_t idx = ((size_t)n_ * Cin + l31) * HW + (size_t)(g - (long)n_ * (long)HW);
        float4 o = make_float4(acc[4 * j] + rr, acc[4 * j + 1] + rr, acc[4 * j + 2] + rr, acc[4 * j + 3] + rr);
        for (int gr = 0; gr < ngroups; ++gr) {
          const float4 t = *reinterpret_cast<const float4*>(dxp + (size_t)gr * gstride + idx);
          o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
        }
        if (p.add0) { const float4 t = *reinterpret_cast<const float4*>(p.add0 + idx); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
        if (p.accum) { const float4 t = *reinterpret_cast<const float4*>(p.dx + idx); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
        *reinterpret_cast<float4*>(p.dx + idx) = o;
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
  // stride 2 only: there the expanded tensor is four times the depthwise output and its eight passes dominate the block; at stride 1
  // the recomputation (tile halos, staging of an equally large d) measured slower than the separate kernels
  return S == 2 && Cin >= 8 && Cin <= 32 && Cin % 8 == 0 && Hd >= 8 && Hd <= 32 * IRT_MAXCH && H >= 4 && W >= 8 && H % 4 == 0 && W % 8 == 0;
}
int irt_stat_wgs(long npb) { const long w = (npb + 3) / 4; return (int)(w < 512 ? w : 512); }
int irt_fwd_tiles(int N, int H, int W, int S, int* tx, int* ty) {
  const int Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
  const int TH = S == 1 ? 8 : 4, TW = S == 1 ? 32 : 16;
  *tx = (Wo + TW - 1) / TW; *ty = (Ho + TH - 1) / TH;
  return N * *tx * *ty;
}
int irt_bwd_tiles(int N, int H, int W, int* tx, int* ty) {        // tiles of 4 x 32 e pixels
  *tx = (W + 31) / 32; *ty = (H + 3) / 4;
  return N * *tx * *ty;
}
// rows (= work-groups along x) of the backward sweep: two 256-thread work-groups per CU over both grid dimensions
int irt_bwd_rows(int N, int hidden, int H, int W, int* per_out) {
  int tx, ty;
  const int nt = irt_bwd_tiles(N, H, W, &tx, &ty);
  const int ngroups = ((hidden + 31) / 32 + IRT_CG - 1) / IRT_CG;
  int want = 512 / ngroups;
  if (want < 1) want = 1;
  int per = (nt + want - 1) / want;
  if (per < 1) per = 1;
  if (per_out) *per_out = per;
  return (nt + per - 1) / per;
}
int irt_mom_wgs(long NP) { const long k = (NP + 15) / 16; const long w = (k + 15) / 16; return (int)(w < 256 ? (w < 1 ? 1 : w) : 256); }

struct IrtWork { float* gpart; float* mpart; double* mfin; float* qr; float* wpk; float* dxp; size_t total; };
IrtWork irt_work(float* work, int N, int Cin, int hidden, int H, int W) {
  IrtWork w;
  const size_t rows = (size_t)irt_bwd_rows(N, hidden, H, W, nullptr), nch = (size_t)(hidden + 31) / 32;
  const size_t ngroups = (nch + IRT_CG - 1) / IRT_CG;
  const size_t mrows = (size_t)irt_mom_wgs((long)N * H * W);
  size_t o = 0;
  w.gpart = work + o; o += rows * nch * 32 * 32;
  w.mpart = work + o; o += mrows * 33 * 32;
  o = (o + 1) & ~(size_t)1;                                    // fp64 scratch: 8-byte aligned (work itself is 16-byte aligned)
  w.mfin = reinterpret_cast<double*>(work + o); o += 2 * 33 * 32;
  w.qr = work + o; o += 33 * 32;
  o = (o + 3) & ~(size_t)3;
  w.wpk = work + o; o += nch * 4 * 3 * 64 * 4;                 // [nch][NKS <= 2][3][64] + [nch][2][3][64] 16-byte entries
  w.dxp = work + o; o += ngroups * (size_t)N * Cin * H * W;
  w.total = o;
  return w;
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
  p.dx = nullptr; p.add0 = nullptr; p.accum = 0; p.wpk = nullptr;
  p.N = a->N; p.Cin = a->Cin; p.Hd = a->hidden; p.H = a->H; p.W = a->W; p.S = a->stride;
  p.Ho = (a->H - 1) / a->stride + 1; p.Wo = (a->W - 1) / a->stride + 1;
  p.nch = (a->hidden + 31) / 32;
  p.tiles_x = p.tiles_y = p.ntiles = p.tiles_per_wg = 0;
  const long NP = (long)a->N * a->H * a->W;
  SC_REQUIRE(NP * a->hidden < (1L << 31), "%s: tensor too large for 32-bit element counts", who);
  p.npb = (int)((NP + 31) / 32);
  return SC_OK;
}
template <typename K>
void irt_lds_attr(K kern, size_t lds) {
  // once per kernel (the pointer identifies the instantiation); not a stream operation, so it is safe under graph capture too
  static const void* done[16];
  static int ndone = 0;
  if (lds <= 64 * 1024) return;
  const void* f = reinterpret_cast<const void*>(kern);
  for (int i = 0; i < ndone; ++i) if (done[i] == f) return;
  (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (ndone < 16) done[ndone++] = f;
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
extern "C" int sc_irt_bwd_rows(int N, int hidden, int H, int W) { return irt_bwd_rows(N, hidden, H, W, nullptr); }

extern "C" size_t sc_irt_bwd_workspace_floats(int N, int Cin, int hidden, int H, int W) {
  return irt_work(nullptr, N, Cin, hidden, H, W).total + 16;
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
  SC_REQUIRE(d_out, "sc_irt_fwd: null output");            // stats_d == NULL: inference (eval-mode constants in cst_expand)
  p.dout = d_out; p.stats = stats_d;
  p.ntiles = irt_fwd_tiles(p.N, p.H, p.W, p.S, &p.tiles_x, &p.tiles_y);
  const int nks = (p.Cin + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(32 * IrtFwdGeo<2>::EPAD + 64 + p.nch * 32 * 12) * 4 + (size_t)p.nch * nks * 3 * 64 * 16;
  // persistent work-groups: as many as fit the chip at this LDS size (160 KB per CU), at most one per tile
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  const int wgs = p.ntiles < 256 * per_cu ? p.ntiles : 256 * per_cu;
  if (nks == 1) { irt_lds_attr(&k_irt_fwd<1, 2>, lds); hipLaunchKernelGGL((k_irt_fwd<1, 2>), dim3(wgs), dim3(256), lds, st, p); }
  else { irt_lds_attr(&k_irt_fwd<2, 2>, lds); hipLaunchKernelGGL((k_irt_fwd<2, 2>), dim3(wgs), dim3(256), lds, st, p); }
  SC_LAUNCH_OK("sc_irt_fwd");
  return SC_OK;
}

extern "C" int sc_irt_bwd(const sc_irt_args* a, const sc_src* dy_d, double* e_sums, double* dw_acc, float* work, sc_stream stream) {
  IrtP p;
  if (int rc = irt_fill(p, a, "sc_irt_bwd")) return rc;
  SC_REQUIRE(dy_d && dy_d->x && dy_d->C == p.Hd && dy_d->up == 0, "sc_irt_bwd: dy must be a source of `hidden` channels at the depthwise output's size");
  SC_REQUIRE(dy_d->mode == SC_SRC_RAW || (dy_d->mode == SC_SRC_BNBWD && dy_d->aux && dy_d->cst), "sc_irt_bwd: dy must be a RAW or BNBWD source");
  SC_REQUIRE(e_sums && dw_acc && work && ((uintptr_t)work & 15) == 0, "sc_irt_bwd: null / misaligned output");
  p.dy = to_srcd(*dy_d);
  p.ntiles = irt_bwd_tiles(p.N, p.H, p.W, &p.tiles_x, &p.tiles_y);
  const int rows = irt_bwd_rows(p.N, p.Hd, p.H, p.W, &p.tiles_per_wg);
  const IrtWork w = irt_work(work, p.N, p.Cin, p.Hd, p.H, p.W);
  p.esums = e_sums; p.dwacc = dw_acc; p.gpart = w.gpart; p.dx = w.dxp;
  const int nks = (p.Cin + 15) / 16;
  const int ngroups = (p.nch + IRT_CG - 1) / IRT_CG;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)irt_bwd_lds().total;      // (measured with a padded request: the second work-group per CU is lost between 80.0 and 82 KB)
  const dim3 grid(rows, ngroups);
  uintx4* wpk = reinterpret_cast<uintx4*>(w.wpk);
  p.wpk = wpk;
  if (nks == 1) {
    hipLaunchKernelGGL((k_irt_pack<1>), dim3(p.nch), dim3(256), 0, st, p, wpk);
    irt_lds_attr(&k_irt_bwd<1>, lds); hipLaunchKernelGGL((k_irt_bwd<1>), grid, dim3(256), lds, st, p);
  } else {
    hipLaunchKernelGGL((k_irt_pack<2>), dim3(p.nch), dim3(256), 0, st, p, wpk);
    irt_lds_attr(&k_irt_bwd<2>, lds); hipLaunchKernelGGL((k_irt_bwd<2>), grid, dim3(256), lds, st, p);
  }
  SC_LAUNCH_OK("sc_irt_bwd");
  return SC_OK;
}

extern "C" int sc_irt_xmoments(const sc_irt_args* a, float* work, sc_stream stream) {
  IrtP p;
  if (int rc = irt_fill(p, a, "sc_irt_xmoments")) return rc;
  SC_REQUIRE(work && ((uintptr_t)work & 15) == 0, "sc_irt_xmoments: null / misaligned workspace");
  const IrtWork w = irt_work(work, p.N, p.Cin, p.Hd, p.H, p.W);
  p.mpart = w.mpart;
  const int wgs = irt_mom_wgs((long)p.N * p.H * p.W);
  hipLaunchKernelGGL(k_irt_xmom, dim3(wgs), dim3(256), (3 * 16 * 64 + 4 * 32) * sizeof(float), (hipStream_t)stream, p);
  SC_LAUNCH_OK("sc_irt_xmoments");
  return SC_OK;
}

extern "C" int sc_irt_bwd_fix(const sc_irt_args* a, const float* cst_bwd_expand, float* work, float* dx, const float* add0, int accum,
                              sc_stream stream) {
  IrtP p;
  if (int rc = irt_fill(p, a, "sc_irt_bwd_fix")) return rc;
  SC_REQUIRE(cst_bwd_expand && work && dx && ((uintptr_t)work & 15) == 0, "sc_irt_bwd_fix: null / misaligned argument");
  const IrtWork w = irt_work(work, p.N, p.Cin, p.Hd, p.H, p.W);
  p.dx = dx; p.add0 = add0; p.accum = accum;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_irt_qr, dim3(p.Cin), dim3(256), 0, st, p.we, cst_bwd_expand, p.Hd, p.Cin, w.qr);
  const int nks = (p.Cin + 15) / 16;
  const int ngroups = (p.nch + IRT_CG - 1) / IRT_CG;
  const int wgs = irt_stat_wgs(p.npb) * 2 > (p.npb + 3) / 4 ? (p.npb + 3) / 4 : irt_stat_wgs(p.npb) * 2;
  if (nks == 1) hipLaunchKernelGGL((k_irt_fix<1>), dim3(wgs), dim3(256), 0, st, p, w.dxp, ngroups, w.qr);
  else hipLaunchKernelGGL((k_irt_fix<2>), dim3(wgs), dim3(256), 0, st, p, w.dxp, ngroups, w.qr);
  SC_LAUNCH_OK("sc_irt_bwd_fix");
  return SC_OK;
}

extern "C" int sc_irt_wgrad_finalize(const sc_irt_args* a, const float* cst_bwd_expand, float* work, float* dw_expand, sc_stream stream) {
  SC_REQUIRE(a && cst_bwd_expand && work && dw_expand && a->w_expand, "sc_irt_wgrad_finalize: null argument");
  SC_REQUIRE(irt_ok(a->Cin, a->hidden, a->H, a->W, a->stride), "sc_irt_wgrad_finalize: unsupported block");
  const IrtWork w = irt_work(work, a->N, a->Cin, a->hidden, a->H, a->W);
  const int rows = irt_bwd_rows(a->N, a->hidden, a->H, a->W, nullptr);
  const int mrows = irt_mom_wgs((long)a->N * a->H * a->W);
  const int nch = (a->hidden + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_irt_msum, dim3(33), dim3(256), 0, st, w.mpart, mrows, w.mfin);
  hipLaunchKernelGGL(k_irt_dwe, dim3(a->hidden), dim3(256), 0, st, w.gpart, rows, nch * 32, w.mfin, a->w_expand, cst_bwd_expand, dw_expand, a->hidden,
                     a->Cin);
  SC_LAUNCH_OK("sc_irt_wgrad_finalize");
  return SC_OK;
}
