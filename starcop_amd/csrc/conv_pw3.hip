// Pointwise (1x1) convolutions of the MobileNetV2 encoder on the 16-bit matrix cores with fp32 accuracy: forward,
// backward-data and backward-weight.  (torch.nn.functional.conv2d with a 1x1 filter + the producers' batch_norm / relu6 and
// their autograd backward, as smp.Unet('mobilenet_v2') dispatches them: /root/reference/starcop/models/model_module.py:244-251.)
//
// Why a second family beside k_conv_mfma<1> (conv_mfma.hip, fp32 MFMA + LDS staging): 34 of the network's 63 convolutions are
// pointwise, and two thirds of their launches run at 32x32 or 16x16 pixels, where a launch has 1-3 rounds of work-groups and a
// work-group's life is one serial chain (constants -> global loads -> LDS -> barrier -> 8..16 MFMAs -> a 750-instruction
// statistics epilogue): 22-58 us per launch for 3-8 us of traffic and matrix work.  This family has
//   * NO LDS staging and NO barrier in the K loop.  GEMM view  D[pixel][cout] = sum_k X[pixel][k] W[k][cout]  on
//     v_mfma_f32_32x32x16_bf16:  A (32 x 16): lane l -> pixel l&31, k = 8*(l>>5) .. +7;  B (16 x 32): lane l -> cout l&31, same k.
//     In NCHW the A operand IS a coalesced read (32 consecutive pixels of channel c and of channel c+8 per load instruction), B
//     comes pre-split and pre-permuted from the packed filter (16 bytes per lane).  A wave owns 32 pixels x (32*NCB) couts and is
//     independent of every other wave; per-channel constants sit in LDS (read on the LDS counter, so the global-load ring keeps
//     its depth), global loads run PD K-steps ahead in a register ring that is refilled in place (straight-line, exact vmcnt).
//   * fp32 operands split EXACTLY into three bf16 terms (a = a0 + a1 + a2), the six leading products accumulated in fp32:
//     one fp32 rounding per product, fp32's exponent range, no scaling and no range assumptions (the "fp32-x3" arithmetic of
//     conv_bx3.hip).  Pointwise layers are nowhere near MFMA-bound, so the cheaper two-term split would buy nothing here.
//   * pixels on the accumulator ROWS, channels on its COLUMNS: a channel's sum / sum of squares over the wave's 32 pixels is
//     the sum of the lane's 16 accumulator registers (+ one cross-half add) -- 34 VALU per 32x32 block instead of 160 DPP steps;
//     the store is four 16-byte segments per lane (4 consecutive pixels of the lane's channel).
//   * the weight gradient  dW[cout][cin] = sum_pixels dy[cout][px] x[cin][px]  with K = pixels: both operands are 32-byte
//     contiguous reads per lane (8 pixels of the lane's channel), split in registers; the four waves of a work-group take
//     interleaved K steps and sum their accumulators through LDS in a fixed order before the partial is written.
#include "sc_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) unsigned int uintx4;

// exact three-term bf16 split of two floats; packed pairs (low half = first value)
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

struct PwP {
  SrcD s;                 // the single source: AFFINE / RAW (forward) or BNBWD (backward-data)
  const uintx4* wpk;      // [co block][k step][term][lane] 16-byte entries (k_pack item layout below)
  int NP, HW, K, M, nks, npb;
  float* out; const float* add0; int accum; float* stats;
  // sc_bnr_args (data-gradient launches that write a tensor's COMPLETE gradient): the BatchNorm-backward sums of that tensor from the
  // epilogue -- rows [pixel block][M][2] = {sum g', sum g' x_hat} -- instead of a sc_bn_bwd_small launch over (gradient, y) behind it
  const float* bnr_y; const float* bnr_cst; float* bnr_rows; int bnr_act;
};

// NCB: 32-cout blocks per wave; PD: K steps of global loads in flight (ring depth); BNB: BatchNorm-backward source
template <int NCB, int PD, bool BNB>
__global__ __launch_bounds__(256) void k_pw3(const PwP p) {
  constexpr int CW = BNB ? 8 : 2;
  extern __shared__ __attribute__((aligned(16))) float s_cst[];              // [nks*16][CW]: forward (scale, shift) | backward (scale, shift, A, B, D, -, -, -)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HW = p.HW, K = p.K, M = p.M, nks = p.nks;
  // ---- per-channel constants -> LDS (channels past K: zeros, so a padded operand is an exact 0)
  for (int c = tid; c < nks * 16; c += 256) {
    if (BNB) {
      float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f); float c4 = 0.f;
      if (c < K) { c0 = *reinterpret_cast<const float4*>(p.s.cst + (size_t)c * SC_CST); c4 = p.s.cst[(size_t)c * SC_CST + 4]; }
      *reinterpret_cast<float4*>(s_cst + c * 8) = c0;
      *reinterpret_cast<float4*>(s_cst + c * 8 + 4) = make_float4(c4, 0.f, 0.f, 0.f);
    } else {
      float2 c0 = make_float2(0.f, 0.f);
      if (c < K) c0 = *reinterpret_cast<const float2*>(p.s.cst + (size_t)c * SC_CST);
      *reinterpret_cast<float2*>(s_cst + c * 2) = c0;
    }
  }
  __syncthreads();
  const int pb = blockIdx.x * 4 + wave;
  if (pb >= p.npb) return;
  const int cbase = blockIdx.y * NCB;
  const float lo = sc_act_lo(p.s.act), hi = sc_act_hi(p.s.act);
  const int gp = pb * 32 + l31;
  const bool pok = gp < p.NP;
  const int gpc = pok ? gp : 0;
  const int n = gpc / HW, px = gpc - n * HW;
  const float* xb = p.s.x + ((size_t)n * K) * HW + px;
  const float* ab = BNB ? p.s.aux + ((size_t)n * K) * HW + px : nullptr;
  const uintx4* wb = p.wpk + ((size_t)cbase * nks * 3) * 64 + lane;

  float xr[PD][8];
  float yr[BNB ? PD : 1][8];
  uintx4 br[PD][NCB][3];
  floatx16 acc[NCB];
#pragma unroll
  for (int m = 0; m < NCB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

#define PW3_ISSUE(slot, ks_)                                                                   \
  {                                                                                            \
    const int kk_ = (ks_) < nks ? (ks_) : nks - 1;                                              \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
      const int c_ = kk_ * 16 + lhi * 8 + j;                                                   \
      const unsigned o_ = (unsigned)(c_ < K ? c_ : K - 1) * (unsigned)HW;                      \
      xr[slot][j] = xb[o_];                                                                    \
      if (BNB) yr[BNB ? slot : 0][j] = ab[o_];                                                 \
    }                                                                                          \
    _Pragma("unroll") for (int m = 0; m < NCB; ++m)                                            \
      _Pragma("unroll") for (int t = 0; t < 3; ++t) br[slot][m][t] = wb[(((size_t)m * nks + kk_) * 3 + t) * 64]; \
  }

#pragma unroll
  for (int s = 0; s < PD; ++s) PW3_ISSUE(s, s)

  for (int ks0 = 0; ks0 < nks; ks0 += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      const int ks = ks0 + s;
      const int kc = ks < nks ? ks : nks - 1;
      const bool live = pok && ks < nks;
      float v[8];
      if (BNB) {
        const float4* q = reinterpret_cast<const float4*>(s_cst + (kc * 16 + lhi * 8) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 c0 = q[2 * j]; const float c4 = q[2 * j + 1].x;
          const float t = sc_pro_bnbwd(xr[s][j], yr[BNB ? s : 0][j], c0.x, c0.y, c0.z, c0.w, c4, lo, hi);
          v[j] = live ? t : 0.f;
        }
      } else {
        const float4* q = reinterpret_cast<const float4*>(s_cst + (kc * 16 + lhi * 8) * 2);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float4 c = q[jj];
          const float t0 = sc_pro_affine(xr[s][2 * jj], c.x, c.y, lo, hi), t1 = sc_pro_affine(xr[s][2 * jj + 1], c.z, c.w, lo, hi);
          v[2 * jj] = live ? t0 : 0.f;
          v[2 * jj + 1] = live ? t1 : 0.f;
        }
      }
      uintx4 a[3];
      split8(v, a);
#pragma unroll
      for (int m = 0; m < NCB; ++m) acc[m] = mfma6(a, br[s][m], acc[m]);
      PW3_ISSUE(s, ks + PD)
    }
  }
#undef PW3_ISSUE

  // ---- epilogue: acc[m][i] = D[pixel 8*(i/4) + 4*lhi + (i%4)][cout l31 of block cbase+m]
  const bool vec4 = (HW & 3) == 0;
  unsigned ooff[4]; bool ook[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = pb * 32 + 8 * j + 4 * lhi;
    ook[j] = g < p.NP;
    const int gc = ook[j] ? g : 0;
    const int n_ = gc / HW;
    ooff[j] = (unsigned)(n_ * M) * (unsigned)HW + (unsigned)(gc - n_ * HW);
  }
#pragma unroll
  for (int m = 0; m < NCB; ++m) {
    const int co = (cbase + m) * 32 + l31;
    const bool cok = co < M;
    if (p.stats) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s1 += acc[m][r]; s2 = fmaf(acc[m][r], acc[m][r], s2); }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lhi == 0 && cok) *reinterpret_cast<float2*>(p.stats + ((size_t)pb * M + co) * 2) = make_float2(s1, s2);
    }
    if (!cok) continue;
    if (BNB && p.bnr_y != nullptr) {
      // the lane's channel `co` of the output tensor: g' = g act'(BN(y)), sums over the wave's 32 pixels (the lane holds 16, its partner
      // lane + 32 the other 16); the raw values y of the four 16-byte segments are requested together, ahead of the stores
      const float4 cb = *reinterpret_cast<const float4*>(p.bnr_cst + (size_t)co * SC_CST);      // scale, shift, mean, invstd
      const float blo = sc_act_lo(p.bnr_act), bhi = sc_act_hi(p.bnr_act);
      float yv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) yv[r] = 0.f;
      if (vec4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!ook[j]) continue;
          const float4 t = *reinterpret_cast<const float4*>(p.bnr_y + ooff[j] + (unsigned)co * (unsigned)HW);
          yv[4 * j] = t.x; yv[4 * j + 1] = t.y; yv[4 * j + 2] = t.z; yv[4 * j + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int g = pb * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
          if (g < p.NP) { const int n_ = g / HW; yv[r] = p.bnr_y[((size_t)n_ * M + co) * HW + (g - n_ * HW)]; }
        }
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int g = pb * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
        const float yh = fmaf(yv[r], cb.x, cb.y);
        const float gq = (g < p.NP && yh > blo && yh < bhi) ? acc[m][r] : 0.f;
        s1 += gq;
        s2 = fmaf(gq, (yv[r] - cb.z) * cb.w, s2);
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lhi == 0) *reinterpret_cast<float2*>(p.bnr_rows + ((size_t)pb * M + co) * 2) = make_float2(s1, s2);
    }
    if (vec4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!ook[j]) continue;
        const unsigned idx = ooff[j] + (unsigned)co * (unsigned)HW;
        float4 o = make_float4(acc[m][4 * j], acc[m][4 * j + 1], acc[m][4 * j + 2], acc[m][4 * j + 3]);
        if (p.add0) { const float4 t = *reinterpret_cast<const float4*>(p.add0 + idx); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
        if (p.accum) { const float4 t = *reinterpret_cast<const float4*>(p.out + idx); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
        *reinterpret_cast<float4*>(p.out + idx) = o;      // (non-temporal stores measured 1.5-2.4x slower: the 16-byte segments no longer merge in L2)
      }
    } else {      // tiny planes (H*W not a multiple of 4): one pixel at a time
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int g = pb * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
        if (g >= p.NP) continue;
        const int n_ = g / HW;
        const size_t idx = ((size_t)n_ * M + co) * HW + (g - n_ * HW);
        float o = acc[m][r];
        if (p.add0) o += p.add0[idx];
        if (p.accum) o += p.out[idx];
        p.out[idx] = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient: D[co][ci] = sum over flat pixels of dy[co][px] * x[ci][px]
//   A (32 x 16): lane l -> dy[co = 32*mb + l&31][pixels 16*kstep + 8*(l>>5) .. +7]     (two float4 loads per tensor)
//   B (16 x 32): lane l -> x [ci = 32*nb + l&31][same pixels]
// A work-group = 4 waves on the same (TM x TN)-block tile and K slice; wave w takes K steps w, w+4, ... of the slice; the four
// accumulators are summed through LDS (fixed order: bit-reproducible) and written as ONE partial [slice][CoP][CiP].
struct PwWP {
  SrcD dy, s;
  int NP, HW, Cout, Cin;
  int ksteps;            // ceil(NP / 16)
  int per_wg;            // K steps per work-group (a multiple of 4)
  int CoP, CiP;
  float* part;
};

template <int TM, int TN, int PD>
__global__ __launch_bounds__(256) void k_pw3_wgrad(const PwWP p) {
  extern __shared__ __attribute__((aligned(16))) float s_red[];           // [3][TM*TN*16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HW = p.HW;
  const int mb0 = blockIdx.y * TM, nb0 = blockIdx.z * TN;
  // per-lane channels and their constants (fixed for the whole kernel)
  float d_sc[TM], d_sh[TM], d_A[TM], d_B[TM], d_D[TM]; bool d_ok[TM]; unsigned d_c[TM];
  float x_sc[TN], x_sh[TN]; bool x_ok[TN]; unsigned x_c[TN];
  const bool dy_bnb = p.dy.mode == SC_SRC_BNBWD;
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    const int co = (mb0 + m) * 32 + l31;
    d_ok[m] = co < p.Cout;
    d_c[m] = d_ok[m] ? co : 0;
    const float4 c0 = *reinterpret_cast<const float4*>(p.dy.cst + (size_t)d_c[m] * SC_CST);
    d_sc[m] = c0.x; d_sh[m] = c0.y; d_A[m] = c0.z; d_B[m] = c0.w; d_D[m] = p.dy.cst[(size_t)d_c[m] * SC_CST + 4];
  }
#pragma unroll
  for (int q = 0; q < TN; ++q) {
    const int ci = (nb0 + q) * 32 + l31;
    x_ok[q] = ci < p.Cin;
    x_c[q] = x_ok[q] ? ci : 0;
    const float2 c0 = *reinterpret_cast<const float2*>(p.s.cst + (size_t)x_c[q] * SC_CST);
    x_sc[q] = c0.x; x_sh[q] = c0.y;
  }
  const float dlo = sc_act_lo(p.dy.act), dhi = sc_act_hi(p.dy.act);
  const float xlo = sc_act_lo(p.s.act), xhi = sc_act_hi(p.s.act);

  floatx16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int q = 0; q < TN; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

  const int k_begin = blockIdx.x * p.per_wg;
  const int k_end = min(k_begin + p.per_wg, p.ksteps);
  // this wave's K steps: k_begin + wave, + 4, ...
  float4 gr[PD][TM][2], yr[PD][TM][2], xr[PD][TN][2];
  bool okr[PD];

#define PW3W_ISSUE(slot, kst_)                                                                          \
  {                                                                                                     \
    const int kq_ = (kst_);                                                                             \
    const int g0_ = (kq_ * 2 + lhi) * 8;                 /* first of this lane's 8 flat pixels */         \
    okr[slot] = kq_ < k_end && g0_ < p.NP;                                                              \
    const int gc_ = okr[slot] ? g0_ : 0;                                                                \
    const int n_ = gc_ / HW;                                                                            \
    const unsigned px_ = (unsigned)(gc_ - n_ * HW);                                                     \
    _Pragma("unroll") for (int m = 0; m < TM; ++m) {                                                    \
      const size_t o_ = ((size_t)n_ * p.Cout + d_c[m]) * HW + px_;                                      \
      gr[slot][m][0] = *reinterpret_cast<const float4*>(p.dy.x + o_);                                   \
      gr[slot][m][1] = *reinterpret_cast<const float4*>(p.dy.x + o_ + 4);                               \
      if (dy_bnb) {                                                                                     \
        yr[slot][m][0] = *reinterpret_cast<const float4*>(p.dy.aux + o_);                               \
        yr[slot][m][1] = *reinterpret_cast<const float4*>(p.dy.aux + o_ + 4);                           \
      }                                                                                                 \
    }                                                                                                   \
    _Pragma("unroll") for (int q = 0; q < TN; ++q) {                                                    \
      const size_t o_ = ((size_t)n_ * p.Cin + x_c[q]) * HW + px_;                                       \
      xr[slot][q][0] = *reinterpret_cast<const float4*>(p.s.x + o_);                                    \
      xr[slot][q][1] = *reinterpret_cast<const float4*>(p.s.x + o_ + 4);                                \
    }                                                                                                   \
  }

#pragma unroll
  for (int s = 0; s < PD; ++s) PW3W_ISSUE(s, k_begin + wave + 4 * s)

  for (int k0 = k_begin + wave; k0 < k_end; k0 += 4 * PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      const bool live = okr[s];
      uintx4 a[TM][3], b[TN][3];
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const float g8[8] = {gr[s][m][0].x, gr[s][m][0].y, gr[s][m][0].z, gr[s][m][0].w, gr[s][m][1].x, gr[s][m][1].y, gr[s][m][1].z, gr[s][m][1].w};
        const float y8[8] = {yr[s][m][0].x, yr[s][m][0].y, yr[s][m][0].z, yr[s][m][0].w, yr[s][m][1].x, yr[s][m][1].y, yr[s][m][1].z, yr[s][m][1].w};
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = dy_bnb ? sc_pro_bnbwd(g8[j], y8[j], d_sc[m], d_sh[m], d_A[m], d_B[m], d_D[m], dlo, dhi)
                                 : sc_pro_affine(g8[j], d_sc[m], d_sh[m], dlo, dhi);
          v[j] = (live && d_ok[m]) ? t : 0.f;
        }
        split8(v, a[m]);
      }
#pragma unroll
      for (int q = 0; q < TN; ++q) {
        const float x8[8] = {xr[s][q][0].x, xr[s][q][0].y, xr[s][q][0].z, xr[s][q][0].w, xr[s][q][1].x, xr[s][q][1].y, xr[s][q][1].z, xr[s][q][1].w};
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = sc_pro_affine(x8[j], x_sc[q], x_sh[q], xlo, xhi);
          v[j] = (live && x_ok[q]) ? t : 0.f;
        }
        split8(v, b[q]);
      }
      PW3W_ISSUE(s, k0 + 4 * (s + PD))
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int q = 0; q < TN; ++q) acc[m][q] = mfma6(a[m], b[q], acc[m][q]);
    }
  }
#undef PW3W_ISSUE

  // ---- sum the four waves' accumulators (waves 1..3 -> LDS, wave 0 adds them in order), write the partial
  constexpr int NR = TM * TN * 16;
  if (wave > 0) {
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int q = 0; q < TN; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_red[((size_t)(wave - 1) * NR + (m * TN + q) * 16 + r) * 64 + lane] = acc[m][q][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* part = p.part + (size_t)blockIdx.x * p.CoP * p.CiP;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int q = 0; q < TN; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = ((m * TN + q) * 16 + r) * 64 + lane;
          const float v = ((acc[m][q][r] + s_red[o]) + s_red[NR * 64 + o]) + s_red[2 * NR * 64 + o];
          const int co = (mb0 + m) * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
          const int ci = (nb0 + q) * 32 + l31;
          part[(size_t)co * p.CiP + ci] = v;
        }
  }
}


struct PwWPlan { int tm, tn, pd, per_wg, nparts, CoP, CiP, ksteps; };

PwWPlan plan_pw3_wgrad(int N, int H, int W, int Cout, int Cin) {
  PwWPlan pl;
  const long NP = (long)N * H * W;
  pl.ksteps = (int)((NP + 15) / 16);
  const int MB = (Cout + 31) / 32, NB = (Cin + 31) / 32;
  pl.tm = MB >= 2 ? 2 : 1;
  pl.tn = NB >= 2 ? 2 : 1;
  pl.pd = (pl.tm * pl.tn == 4) ? 2 : (pl.tm * pl.tn == 2 ? 3 : 4);
  pl.CoP = (MB + pl.tm - 1) / pl.tm * pl.tm * 32;
  pl.CiP = (NB + pl.tn - 1) / pl.tn * pl.tn * 32;
  const int tiles = (pl.CoP / (32 * pl.tm)) * (pl.CiP / (32 * pl.tn));
  // about two work-groups (eight waves) per CU overall, every wave at least 4 K steps
  int want = (512 + tiles - 1) / tiles;
  if (want < 1) want = 1;
  int per_wg = (pl.ksteps + want - 1) / want;
  per_wg = (per_wg + 3) / 4 * 4;
  if (per_wg < 16) per_wg = 16;
  pl.per_wg = per_wg;
  pl.nparts = (pl.ksteps + per_wg - 1) / per_wg;
  return pl;
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" size_t sc_packed_weight_floats_pw3(int Cout, int Cin, int transpose_flip) {
  const int M = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  const size_t MBE = (size_t)((M + 31) / 32 + 3) / 4 * 4, nks = (K + 15) / 16;
  return MBE * nks * 3 * 64 * 4;
}

extern "C" int sc_conv1x1_pw3(const sc_conv_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_conv1x1_pw3: null args");
  SC_REQUIRE(a->ks == 1 && a->nsrc == 1, "sc_conv1x1_pw3: ks must be 1 with a single source");
  SC_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cout > 0 && a->src[0].C > 0, "sc_conv1x1_pw3: bad shape");
  SC_REQUIRE(a->csplit == a->Cout && a->out0 && !a->out1 && !a->add1 && !a->accum1 && !a->down0,
             "sc_conv1x1_pw3: a single plain output (optional add0 / accum0)");
  SC_REQUIRE(a->src[0].up == 0 && a->src[0].x, "sc_conv1x1_pw3: source cannot be upsampled");
  SC_REQUIRE(a->src[0].mode != SC_SRC_NORM, "sc_conv1x1_pw3: NORM sources belong to the stem");
  SC_REQUIRE(a->src[0].mode == SC_SRC_RAW || a->src[0].cst, "sc_conv1x1_pw3: source constants missing");
  SC_REQUIRE(a->src[0].mode != SC_SRC_BNBWD || a->src[0].aux, "sc_conv1x1_pw3: BNBWD source needs aux");
  SC_REQUIRE(((uintptr_t)a->wpk & 15) == 0, "sc_conv1x1_pw3: packed filters must be 16-byte aligned");
  const long NP = (long)a->N * a->H * a->W;
  SC_REQUIRE(NP * (long)(a->Cout > a->src[0].C ? a->Cout : a->src[0].C) < (1L << 32), "sc_conv1x1_pw3: tensor too large for 32-bit element offsets");
  PwP p;
  p.s = to_srcd(a->src[0]);
  if (p.s.mode == SC_SRC_RAW) {
    p.s.cst = sc_identity_cst_table(p.s.C);
    SC_REQUIRE(p.s.cst != nullptr, "sc_conv1x1_pw3: identity constants unavailable (C = %d)", p.s.C);
    p.s.act = SC_ACT_NONE;
  }
  p.wpk = reinterpret_cast<const uintx4*>(a->wpk);
  p.NP = (int)NP; p.HW = a->H * a->W; p.K = a->src[0].C; p.M = a->Cout;
  p.nks = (p.K + 15) / 16;
  p.npb = (int)((NP + 31) / 32);
  p.out = a->out0; p.add0 = a->add0; p.accum = a->accum0; p.stats = a->stats;
  p.bnr_y = nullptr; p.bnr_cst = nullptr; p.bnr_rows = nullptr; p.bnr_act = SC_ACT_NONE;
  if (a->bnr) {
    SC_REQUIRE(p.s.mode == SC_SRC_BNBWD && !a->add0 && !a->accum0 && !a->stats, "sc_conv1x1_pw3: sc_bnr_args go with a data-gradient launch that writes the complete gradient");
    SC_REQUIRE(a->bnr->y && a->bnr->cst && a->bnr->rows && !a->bnr->absmax, "sc_conv1x1_pw3: sc_bnr_args need y, cst and rows (no range hint on this path)");
    p.bnr_y = a->bnr->y; p.bnr_cst = a->bnr->cst; p.bnr_rows = a->bnr->rows; p.bnr_act = a->bnr->act;
  }
  const int MB = (p.M + 31) / 32;
  int ncb = 1;
  for (int c = 4; c >= 2; c >>= 1)
    if (MB >= c - (c == 4 ? 1 : 0) && (long)p.npb * ((MB + c - 1) / c) >= 1536) { ncb = c; break; }
  const bool bnb = p.s.mode == SC_SRC_BNBWD;
  dim3 grid((p.npb + 3) / 4, (MB + ncb - 1) / ncb);
  const size_t lds = (size_t)p.nks * 16 * (bnb ? 8 : 2) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define SC_PW(NCB, PD, B) hipLaunchKernelGGL((k_pw3<NCB, PD, B>), grid, dim3(256), lds, st, p)
  if (ncb == 4) { if (bnb) SC_PW(4, 2, true); else SC_PW(4, 2, false); }
  else if (ncb == 2) { if (bnb) SC_PW(2, 4, true); else SC_PW(2, 4, false); }
  else { if (bnb) SC_PW(1, 4, true); else SC_PW(1, 4, false); }
#undef SC_PW
  SC_LAUNCH_OK("sc_conv1x1_pw3");
  return SC_OK;
}

extern "C" size_t sc_wgrad_pw3_workspace_floats(int N, int H, int W, int Cout, int Cin) {
  const PwWPlan pl = plan_pw3_wgrad(N, H, W, Cout, Cin);
  const size_t E = (size_t)pl.CoP * pl.CiP;
  return (size_t)pl.nparts * E + sc_reduce_scratch_floats(pl.nparts, E);
}

extern "C" int sc_conv1x1_wgrad_pw3(const sc_wgrad_args* a, sc_wgrad_pending* pending, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_conv1x1_wgrad_pw3: null args");
  SC_REQUIRE(a->ks == 1 && a->nsrc == 1, "sc_conv1x1_wgrad_pw3: ks must be 1 with a single source");
  SC_REQUIRE(a->src[0].C == a->Cin && a->dy.C == a->Cout, "sc_conv1x1_wgrad_pw3: channel mismatch");
  SC_REQUIRE(a->dy.up == 0 && a->src[0].up == 0, "sc_conv1x1_wgrad_pw3: sources cannot be upsampled");
  SC_REQUIRE(a->src[0].mode == SC_SRC_RAW || a->src[0].mode == SC_SRC_AFFINE, "sc_conv1x1_wgrad_pw3: input must be a RAW or AFFINE source");
  SC_REQUIRE(a->dy.mode != SC_SRC_NORM && (a->dy.mode != SC_SRC_BNBWD || a->dy.aux), "sc_conv1x1_wgrad_pw3: bad dy source");
  SC_REQUIRE((a->H * a->W) % 8 == 0, "sc_conv1x1_wgrad_pw3: H*W must be a multiple of 8 (got %d)", a->H * a->W);
  const PwWPlan pl = plan_pw3_wgrad(a->N, a->H, a->W, a->Cout, a->Cin);
  const size_t need = sc_wgrad_pw3_workspace_floats(a->N, a->H, a->W, a->Cout, a->Cin);
  SC_REQUIRE(a->part_floats >= need, "sc_conv1x1_wgrad_pw3: workspace too small (%zu < %zu floats)", a->part_floats, need);
  PwWP p;
  p.dy = to_srcd(a->dy); p.s = to_srcd(a->src[0]);
  if (p.dy.mode == SC_SRC_RAW) { p.dy.cst = sc_identity_cst_table(p.dy.C); p.dy.act = SC_ACT_NONE; SC_REQUIRE(p.dy.cst, "sc_conv1x1_wgrad_pw3: identity constants unavailable"); }
  if (p.s.mode == SC_SRC_RAW) { p.s.cst = sc_identity_cst_table(p.s.C); p.s.act = SC_ACT_NONE; SC_REQUIRE(p.s.cst, "sc_conv1x1_wgrad_pw3: identity constants unavailable"); }
  p.NP = a->N * a->H * a->W; p.HW = a->H * a->W; p.Cout = a->Cout; p.Cin = a->Cin;
  p.ksteps = pl.ksteps; p.per_wg = pl.per_wg; p.CoP = pl.CoP; p.CiP = pl.CiP; p.part = a->part;
  dim3 grid(pl.nparts, pl.CoP / (32 * pl.tm), pl.CiP / (32 * pl.tn));
  const size_t lds = (size_t)3 * pl.tm * pl.tn * 16 * 64 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (pl.tm == 2 && pl.tn == 2) hipLaunchKernelGGL((k_pw3_wgrad<2, 2, 2>), grid, dim3(256), lds, st, p);
  else if (pl.tm == 2) hipLaunchKernelGGL((k_pw3_wgrad<2, 1, 3>), grid, dim3(256), lds, st, p);
  else if (pl.tn == 2) hipLaunchKernelGGL((k_pw3_wgrad<1, 2, 3>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((k_pw3_wgrad<1, 1, 4>), grid, dim3(256), lds, st, p);
  SC_LAUNCH_OK("sc_conv1x1_wgrad_pw3");
  if (pending) {
    pending->part = a->part; pending->dw = a->dw; pending->nparts = pl.nparts; pending->taps = 1;
    pending->Cout = a->Cout; pending->Cin = a->Cin; pending->CoP = pl.CoP; pending->CiP = pl.CiP;
    pending->total = (uint64_t)a->Cout * a->Cin;
    return SC_OK;
  }
  return sc_wgrad_finish(a->part, pl.nparts, 1, a->Cout, a->Cin, pl.CoP, pl.CiP, a->dw, st);
}
