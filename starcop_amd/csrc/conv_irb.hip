// One launch per MobileNetV2 inverted-residual block in INFERENCE (eval-mode BatchNorm: the three BatchNorms are per-channel affine
// constants, nothing batch-wide separates the three convolutions):
//     x -> conv1x1 (Cin -> hidden) -> BN_e + ReLU6 -> depthwise 3x3 (stride 1, pad 1) -> BN_d + ReLU6 -> conv1x1 (hidden -> Cout) [-> BN_p, + x]
// torchvision InvertedResidual.conv inside smp.Unet('mobilenet_v2').encoder (starcop/models/model_module.py:244-251, eval mode: the
// notebook / padded_predict path, starcop/models/utils/padding.py:13-50, and the validation loop).  Replaces, per block and forward,
// sc_conv1x1_* (expand) | sc_dwconv3x3_fwd | sc_conv1x1_* (project) | sc_add_srcs -- at 32 x 32 / 16 x 16 planes those are three or four
// dependent 10-35 us launches for 3-8 us of work each; the expanded tensors e and d never leave the CU.
//
// Work-group = 4 waves = one image n, one TH x TW tile of output pixels, ALL hidden channels (in chunks of HC) and all output channels.
//   staging   the (TH + 2) x (TW + 2) input patch, all Cin channels: affine prologue, exact three-bf16-term split -> LDS in MFMA-operand
//             form  s_x[term][cin / 8][pixel] (16 bytes = 8 channels of one pixel), once per work-group
//   per chunk of HC hidden channels:
//     P1  E[pixel][hid] = X W_e^T on v_mfma_f32_32x32x16_bf16 (six products: fp32 accuracy, no range assumptions -- the arithmetic of
//         conv_pw3.hip): A = patch pixels from LDS, B = the PW3 filter pack straight from global memory (requested one chunk ahead);
//         pixels on the accumulator rows, so a lane owns ONE hidden channel: BN_e + ReLU6 are in-lane, zero outside the image (the
//         depthwise convolution pads e, not x) -> s_e[pixel][hid] (fp32)
//     P2  thread = (output pixel, 8 hidden channels): the 3 x 3 stencil from s_e, BN_d + ReLU6, three-term split -> s_d[term][hid / 8][pixel]
//     P3  P[pixel][cout] += D W_p^T: A = s_d, B = the projection's PW3 pack (requested at the top of the chunk); accumulators live
//         across all chunks (a wave owns up to MAXPP (pixel block, cout block) pairs)
//   epilogue  raw p, or -- residual blocks -- z = x + scale_p p + shift_p with the running max |z| (the range record of
//             sc_add_srcs_absmax), 16-byte stores of four neighbouring pixels.
// Two barriers per chunk; one work-group per CU (up to 512 registers per lane: the filter operands of the next MFMA phase are in
// flight in registers while the current phase runs).
#include "sc_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) unsigned int uintx4;

// exact three-term bf16 split of two floats (a = t0 + t1 + t2 up to 2^-24 |a|); packed pairs, low half = first value
__device__ __forceinline__ void irb_split3x2(float a, float b, unsigned& t0, unsigned& t1, unsigned& t2) {
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
__device__ __forceinline__ void irb_split8(const float (&v)[8], uintx4 (&t)[3]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned t0, t1, t2;
    irb_split3x2(v[2 * q], v[2 * q + 1], t0, t1, t2);
    t[0][q] = t0; t[1][q] = t1; t[2][q] = t2;
  }
}
__device__ __forceinline__ floatx16 irb_mfma(const uintx4& a, const uintx4& b, const floatx16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six products of weight >= 2^-24, smallest first.  (Two alternating accumulators per chain -- independent back-to-back MFMAs --
// measured 5-10 % SLOWER: the chains are not what a chunk waits for, tools/bench_irb.py.)
__device__ __forceinline__ floatx16 irb_mfma6(const uintx4 (&a)[3], const uintx4 (&b)[3], floatx16 c) {
  c = irb_mfma(a[1], b[1], c);
  c = irb_mfma(a[2], b[0], c);
  c = irb_mfma(a[0], b[2], c);
  c = irb_mfma(a[1], b[0], c);
  c = irb_mfma(a[0], b[1], c);
  c = irb_mfma(a[0], b[0], c);
  return c;
}

struct IrbP {
  SrcD x;                  // block input [N][Cin][H][W]: RAW or AFFINE
  const uintx4* we;        // PW3 pack of the expansion filter  [hidden / 32][nks_e][3][64] 16-byte entries
  const uintx4* wp;        // PW3 pack of the projection filter [Cout / 32 (padded to 4 blocks)][nks_p][3][64]
  const float* cst_e; const float* wdw; const float* cst_d; const float* cst_p;
  float* out; float* zmax;
  int N, H, W, Ho, Wo, Cin, hid, Cout, nks_e, nks_p, tiles_x, tiles_y, residual;
};

constexpr int IRB_KROWS = 13;      // per-chunk constant rows in LDS: 9 depthwise taps, scale_e, shift_e, scale_d, shift_d

template <int TH, int TW, int HC, int S = 1>
struct IrbCfg {
  static constexpr int PH = (TH - 1) * S + 3, PWD = (TW - 1) * S + 3, NPX = PH * PWD;      // input patch of a TH x TW output tile (stride S, pad 1)
  static constexpr int MBE = (NPX + 31) / 32, NPXP = MBE * 32;      // patch pixels, padded to whole 32-pixel MFMA blocks
  static constexpr int NOUT = TH * TW, MBP = NOUT / 32;
  static constexpr int KGC = HC / 8, NBE = HC / 32;
  static constexpr int NTHR = NOUT * KGC, NW = NTHR / 64;            // one thread per (output pixel, 8 hidden channels) in the stencil phase
  // expansion work of a wave: ONE hidden block nb = wave % NBE (one set of filter operands in registers) x the pixel blocks
  // wave / NBE, + NW / NBE, ...
  static constexpr int WPB = NW / NBE, MAXEP = (MBE + WPB - 1) / WPB;
  static_assert(NW % NBE == 0, "waves per hidden block");
  static constexpr int EPITCH = HC + 4;                                // s_e row pitch in floats (16-byte aligned rows, 4 mod 32 banks)
  static_assert(NOUT % 32 == 0 && HC % 32 == 0 && (NOUT * KGC == 256 || NOUT * KGC == 512), "stencil phase: one thread per (output pixel, 8 hidden channels)");
  static_assert(TW % 4 == 0, "the store takes four neighbouring pixels of a row");
};

template <int TH, int TW, int HC, int S = 1>
static inline size_t irb_smem_bytes(int nks_e) {
  using Cfg = IrbCfg<TH, TW, HC, S>;
  return (size_t)3 * (2 * nks_e) * Cfg::NPXP * 16 + (size_t)Cfg::NPXP * Cfg::EPITCH * 4 + (size_t)3 * Cfg::KGC * Cfg::NOUT * 16 +
         (size_t)2 * IRB_KROWS * HC * 4;
}

// NKE: expansion K steps held in registers (>= ceil(Cin / 16)); MAXPP: (pixel block, cout block) projection pairs per wave.
// Every global load of the chunk loop is UNCONDITIONAL (clamped indices, harmless re-fetches past the end): hipcc counts outstanding
// loads exactly only along straight-line code, a guarded load turns every later wait into vmcnt(0) -- the first version of this
// kernel, with `if (ks < nks_e)` / `if (c + 1 < nchunk)` around its requests, waited for the projection's filter operands (just
// requested) before the expansion's MFMAs and for the next chunk's expansion operands before the projection's: two exposed L2 round
// trips per chunk, 4 us per chunk instead of ~1.2 (features.8 at batch 16: 48.7 us for the block).
// OCC: work-groups per CU the register budget allows (2: <= 256 registers; needs <= 80 KB of LDS, i.e. Cin <= 96 with 4 x 8 tiles).
// One work-group per CU = one wave per SIMD: every LDS / MFMA / memory latency of the serial chunk chain is exposed (elimination
// builds, tools/build_exp_irb.sh: no single phase is more than a quarter of a chunk's 3.5 us) -- a second resident work-group is what
// overlaps them.
template <int TH, int TW, int HC, int NKE, int MAXPP, int OCC, int S = 1>
__global__ __launch_bounds__(TH * TW * (HC / 8), OCC) void k_irb(const IrbP p) {
  using Cfg = IrbCfg<TH, TW, HC, S>;
  constexpr int PWD = Cfg::PWD, NPX = Cfg::NPX, MBE = Cfg::MBE, NPXP = Cfg::NPXP, NOUT = Cfg::NOUT, MBP = Cfg::MBP;
  constexpr int KGC = Cfg::KGC, NBE = Cfg::NBE, WPB = Cfg::WPB, MAXEP = Cfg::MAXEP, EPITCH = Cfg::EPITCH;
  constexpr int KSC = HC / 16;                  // projection K steps per chunk
  constexpr int NTHR = Cfg::NTHR, NW = Cfg::NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nks_e = p.nks_e, KGIN = 2 * nks_e;
  uintx4* const s_x = reinterpret_cast<uintx4*>(smem);                                          // [3][KGIN][NPXP]
  float* const s_e = reinterpret_cast<float*>(smem + (size_t)3 * KGIN * NPXP * 16);            // [NPXP][EPITCH]
  uintx4* const s_d = reinterpret_cast<uintx4*>(reinterpret_cast<unsigned char*>(s_e) + (size_t)NPXP * EPITCH * 4);   // [3][KGC][NOUT]
  float* const s_k = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(s_d) + (size_t)3 * KGC * NOUT * 16);  // [2][13][HC]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;      // (H, W: the INPUT plane; the output plane is Ho x Wo)
  const int Ho = p.Ho, Wo = p.Wo;
  const int per_img = p.tiles_x * p.tiles_y;
  const int n = __builtin_amdgcn_readfirstlane((int)blockIdx.x / per_img);
  const int tile = (int)blockIdx.x - n * per_img;
  const int ty = __builtin_amdgcn_readfirstlane(tile / p.tiles_x), tx = tile - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;                         // output tile origin
  const int iy0 = y0 * S - 1, ix0 = x0 * S - 1;                 // input coordinates of patch pixel (0, 0)
  const size_t HW = (size_t)H * W, HWo = (size_t)Ho * Wo;
  const int nchunk = p.hid / HC;
  const int CB = (Cout + 31) >> 5, PP = MBP * CB;

  // ---- per-chunk constants of chunk 0 -> s_k[0]; request helper for the following chunks (13 * HC floats: <= 4 per thread)
  constexpr int NKV = (IRB_KROWS * HC + NTHR - 1) / NTHR;
  float kv[NKV];
  // (one address per item, computed once: row < 9 -> a depthwise tap, 9 / 10 -> BN_e scale / shift, 11 / 12 -> BN_d; advancing a chunk
  // adds a per-item stride)
  const float* kbase[NKV]; unsigned kstride[NKV];
#pragma unroll
  for (int u = 0; u < NKV; ++u) {
    const int i = min(tid + NTHR * u, IRB_KROWS * HC - 1);
    const int row = i / HC, ch = i - row * HC;
    kbase[u] = row < 9 ? p.wdw + (size_t)ch * 9 + row : (row < 11 ? p.cst_e + (size_t)ch * SC_CST + (row - 9) : p.cst_d + (size_t)ch * SC_CST + (row - 11));
    kstride[u] = (unsigned)HC * (row < 9 ? 9u : (unsigned)SC_CST);
  }
  auto k_request = [&](int c) {
    const int cc = c < nchunk ? c : nchunk - 1;
#pragma unroll
    for (int u = 0; u < NKV; ++u) kv[u] = kbase[u][(size_t)cc * kstride[u]];
  };
  auto k_store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int i = tid + NTHR * u;
      if (i < IRB_KROWS * HC) s_k[buf * IRB_KROWS * HC + i] = kv[u];
    }
  };
  k_request(0);

  // ---- filter operands in flight: expansion (chunk c + 1 during chunk c) and projection (chunk c from its top)
  const int enb = wave % NBE;
  int emb[MAXEP];
#pragma unroll
  for (int q = 0; q < MAXEP; ++q) emb[q] = wave / NBE + WPB * q;      // (>= MBE: no such block for this wave)
  uintx4 be[NKE][3];
  auto e_request = [&](int c) {
    const int cc = c < nchunk ? c : nchunk - 1;
    const uintx4* wb = p.we + ((size_t)(cc * NBE + enb) * nks_e * 3) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < NKE; ++ks) {
      const int kk = ks < nks_e ? ks : nks_e - 1;
#pragma unroll
      for (int t = 0; t < 3; ++t) be[ks][t] = wb[((size_t)kk * 3 + t) * 64];
    }
  };
  uintx4 bp[MAXPP][KSC][3];
  int pmb[MAXPP], pcb[MAXPP];
#pragma unroll
  for (int q = 0; q < MAXPP; ++q) {
    const int pr = wave + NW * q;
    const int prc = pr < PP ? pr : 0;
    pmb[q] = prc % MBP; pcb[q] = prc / MBP;
  }
  auto p_request = [&](int c) {
#pragma unroll
    for (int q = 0; q < MAXPP; ++q) {
      const uintx4* wb = p.wp + ((size_t)pcb[q] * p.nks_p * 3) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
        for (int t = 0; t < 3; ++t) bp[q][ks][t] = wb[((size_t)(c * KSC + ks) * 3 + t) * 64];
    }
  };
  e_request(0);

  // ---- stage the input patch: item = (8-channel group, patch pixel); lanes = consecutive patch pixels
  {
    const bool raw = p.x.mode == SC_SRC_RAW;
    const float lo = raw ? -__builtin_inff() : sc_act_lo(p.x.act), hi = raw ? __builtin_inff() : sc_act_hi(p.x.act);
    const float* const xn = p.x.x + (size_t)n * Cin * HW;
    for (int it = tid; it < KGIN * NPXP; it += NTHR) {
      const int kg = it / NPXP, px = it - kg * NPXP;
      const int pr = px / PWD, pc = px - pr * PWD;
      const int yy = iy0 + pr, xx = ix0 + pc;
      const bool ok = px < NPX && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const size_t off = ok ? (size_t)yy * W + xx : 0;
      float xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = kg * 8 + j;
        xv[j] = xn[(size_t)(c < Cin ? c : Cin - 1) * HW + off];
      }
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = kg * 8 + j;
        const int cc = c < Cin ? c : Cin - 1;
        const float sc = raw ? 1.f : p.x.cst[(size_t)cc * SC_CST], sh = raw ? 0.f : p.x.cst[(size_t)cc * SC_CST + 1];
        v[j] = (ok && c < Cin) ? sc_pro_affine(xv[j], sc, sh, lo, hi) : 0.f;
      }
      uintx4 t[3];
      irb_split8(v, t);
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3) s_x[((size_t)c3 * KGIN + kg) * NPXP + px] = t[c3];
    }
  }
  // which of the lane's 16 accumulator pixels of an expansion block lie inside the image (e is ZERO-padded for the stencil)
  unsigned emask[MAXEP];
#pragma unroll
  for (int q = 0; q < MAXEP; ++q) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int px = emb[q] * 32 + 8 * (i >> 2) + 4 * lhi + (i & 3);
      const int pr = px / PWD, pc = px - pr * PWD;
      const int yy = iy0 + pr, xx = ix0 + pc;
      m |= (px < NPX && yy >= 0 && yy < H && xx >= 0 && xx < W) ? (1u << i) : 0u;
    }
    emask[q] = m;
  }
  k_store(0);
  __syncthreads();

  floatx16 accp[MAXPP];
#pragma unroll
  for (int q = 0; q < MAXPP; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[q][r] = 0.f;

  const int opx = tid % NOUT, okg = tid / NOUT;         // stencil phase: output pixel, 8-channel group of the chunk
  const int ooy = opx / TW, oox = opx - ooy * TW;

  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    const float* const sk = s_k + buf * IRB_KROWS * HC;
    p_request(c);
    k_request(c + 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- P1: expansion of this wave's (pixel block, hidden block) pairs
#pragma unroll
    for (int q = 0; q < MAXEP; ++q) {
      if (emb[q] < MBE) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKE; ++ks) {
          if (NKE == 1 || ks < nks_e) {          // (uniform; no load inside)
            uintx4 a[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = s_x[((size_t)t * KGIN + 2 * ks + lhi) * NPXP + emb[q] * 32 + l31];
            acc = irb_mfma6(a, be[ks], acc);
          }
        }
        // acc[i] = E[pixel emb*32 + 8*(i/4) + 4*lhi + (i%4)][hidden channel enb*32 + l31]
        const int ch = enb * 32 + l31;
        const float sc = sk[9 * HC + ch], sh = sk[10 * HC + ch];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int px = emb[q] * 32 + 8 * (i >> 2) + 4 * lhi + (i & 3);
          const float v = __builtin_amdgcn_fmed3f(fmaf(acc[i], sc, sh), 0.f, 6.f);
          s_e[px * EPITCH + ch] = ((emask[q] >> i) & 1u) ? v : 0.f;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    e_request(c + 1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // ---- P2: depthwise 3x3 + BN_d + ReLU6 + split for (output pixel opx, channels okg*8 .. +7)
    {
      float d8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) d8[j] = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float* ep = s_e + ((ooy * S + kh) * PWD + oox * S + kw) * EPITCH + okg * 8;
          const float4 e0 = *reinterpret_cast<const float4*>(ep), e1 = *reinterpret_cast<const float4*>(ep + 4);
          const float* wq = sk + (kh * 3 + kw) * HC + okg * 8;
          const float4 w0 = *reinterpret_cast<const float4*>(wq), w1 = *reinterpret_cast<const float4*>(wq + 4);
          d8[0] = fmaf(e0.x, w0.x, d8[0]); d8[1] = fmaf(e0.y, w0.y, d8[1]); d8[2] = fmaf(e0.z, w0.z, d8[2]); d8[3] = fmaf(e0.w, w0.w, d8[3]);
          d8[4] = fmaf(e1.x, w1.x, d8[4]); d8[5] = fmaf(e1.y, w1.y, d8[5]); d8[6] = fmaf(e1.z, w1.z, d8[6]); d8[7] = fmaf(e1.w, w1.w, d8[7]);
        }
      const float* cq = sk + 11 * HC + okg * 8;
      const float4 s0 = *reinterpret_cast<const float4*>(cq), s1 = *reinterpret_cast<const float4*>(cq + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(cq + HC), h1 = *reinterpret_cast<const float4*>(cq + HC + 4);
      const float scd[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, shd[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) d8[j] = __builtin_amdgcn_fmed3f(fmaf(d8[j], scd[j], shd[j]), 0.f, 6.f);
      uintx4 t[3];
      irb_split8(d8, t);
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3) s_d[((size_t)c3 * KGC + okg) * NOUT + opx] = t[c3];
    }
    k_store(buf ^ 1);          // (past the last chunk: the same values again, never read)
    __syncthreads();
    // ---- P3: projection of this wave's (pixel block, cout block) pairs, K = the chunk's HC hidden channels
    // (the pairs' chains interleave; pairs past the wave's count multiply the clamped duplicates of pair 0 into an accumulator that is
    // never stored)
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
      uintx4 a[MAXPP][3];
#pragma unroll
      for (int q = 0; q < MAXPP; ++q)
#pragma unroll
        for (int t = 0; t < 3; ++t) a[q][t] = s_d[((size_t)t * KGC + 2 * ks + lhi) * NOUT + pmb[q] * 32 + l31];
      constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
      if (MAXPP > 1 || wave < PP) {      // (uniform)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int q = 0; q < MAXPP; ++q) accp[q] = irb_mfma(a[q][PA[j]], bp[q][ks][PB[j]], accp[q]);
      }
    }
  }

  // ---- epilogue: accp[q][i] = P[output pixel pmb*32 + 8*(i/4) + 4*lhi + (i%4)][cout pcb*32 + l31]
  const bool vec4 = (Wo & 3) == 0;
  const bool xraw = p.x.mode == SC_SRC_RAW;
  float zmx = 0.f;
#pragma unroll
  for (int q = 0; q < MAXPP; ++q) {
    if (wave + NW * q >= PP) continue;
    const int co = pcb[q] * 32 + l31;
    if (co >= Cout) continue;
    float scp = 1.f, shp = 0.f, xsc = 1.f, xsh = 0.f;
    if (p.residual) {
      scp = p.cst_p[(size_t)co * SC_CST]; shp = p.cst_p[(size_t)co * SC_CST + 1];
      if (!xraw) { xsc = p.x.cst[(size_t)co * SC_CST]; xsh = p.x.cst[(size_t)co * SC_CST + 1]; }
    }
    float* const ob = p.out + ((size_t)n * Cout + co) * HWo;
    const float* const xb = p.x.x + ((size_t)n * Cin + co) * HW;       // (residual: Cin == Cout)
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      const int o0 = pmb[q] * 32 + 8 * jq + 4 * lhi;
      const int oy = o0 / TW, ox = o0 - oy * TW;
      const int y = y0 + oy, x = x0 + ox;
      if (y >= Ho || x >= Wo) continue;
      float v[4] = {accp[q][4 * jq], accp[q][4 * jq + 1], accp[q][4 * jq + 2], accp[q][4 * jq + 3]};
      const size_t idx = (size_t)y * Wo + x;      // (residual blocks have stride 1: the input plane is the output plane)
      if (vec4) {          // W % 4 == 0 and x % 4 == 0: the four pixels are inside the row
        if (p.residual) {
          const float4 xv = *reinterpret_cast<const float4*>(xb + idx);
          v[0] = fmaf(v[0], scp, shp) + fmaf(xv.x, xsc, xsh); v[1] = fmaf(v[1], scp, shp) + fmaf(xv.y, xsc, xsh);
          v[2] = fmaf(v[2], scp, shp) + fmaf(xv.z, xsc, xsh); v[3] = fmaf(v[3], scp, shp) + fmaf(xv.w, xsc, xsh);
          zmx = fmaxf(fmaxf(zmx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        *reinterpret_cast<float4*>(ob + idx) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (x + u >= Wo) continue;
          float o = v[u];
          if (p.residual) { o = fmaf(o, scp, shp) + fmaf(xb[idx + u], xsc, xsh); zmx = fmaxf(zmx, fabsf(o)); }
          ob[idx + u] = o;
        }
      }
    }
  }
  if (p.residual && p.zmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zmx = fmaxf(zmx, __shfl_xor(zmx, o, 64));
    if (lane == 0 && zmx > __builtin_nontemporal_load(p.zmax)) atomicMax(reinterpret_cast<unsigned*>(p.zmax), __builtin_bit_cast(unsigned, zmx));
  }
}

template <int TH, int TW, int HC, int NKE, int MAXPP, int OCC, int S = 1>
static int irb_launch3(const IrbP& p, hipStream_t st) {
  const size_t lds = irb_smem_bytes<TH, TW, HC, S>(p.nks_e);
  static bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_irb<TH, TW, HC, NKE, MAXPP, OCC, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  if (!attr_ok) { sc_set_error("sc_irb_eval: cannot raise the dynamic LDS limit"); return SC_ERR_LAUNCH; }
  hipLaunchKernelGGL((k_irb<TH, TW, HC, NKE, MAXPP, OCC, S>), dim3((unsigned)(p.N * p.tiles_x * p.tiles_y)), dim3(IrbCfg<TH, TW, HC, S>::NTHR), lds, st, p);
  SC_LAUNCH_OK("sc_irb_eval");
  return SC_OK;
}
template <int TH, int TW, int HC, int MAXPP, int OCC, int NKEMAX, int S = 1>
static int irb_launch(const IrbP& p, hipStream_t st) {
  if (p.nks_e <= 2) return irb_launch3<TH, TW, HC, 2, MAXPP, OCC, S>(p, st);
  if (p.nks_e <= 4) return irb_launch3<TH, TW, HC, 4, MAXPP, OCC, S>(p, st);
  if (p.nks_e <= 6) return irb_launch3<TH, TW, HC, 6, MAXPP, OCC, S>(p, st);
  if constexpr (NKEMAX >= 10) return irb_launch3<TH, TW, HC, 10, MAXPP, OCC, S>(p, st);
  sc_set_error("sc_irb_eval: internal dispatch error"); return SC_ERR_ARG;
}

// Tilings (tools/bench_irb.py):
//   A  4 x 8 output pixels, 64-channel chunks, 4 waves; two work-groups per CU while the patch fits 80 KB of LDS (Cin <= 96, Cout <= 128)
//   B  8 x 8, 32-channel chunks, 4 waves  (hidden % 64 != 0)
//   C  8 x 8, 64-channel chunks, 8 waves  (Cin <= 96): every work-group streams the block's whole filter set, so twice the pixels per
//      work-group is half the L2 -> L1 filter traffic of A -- measured LEVEL with A on the 32 x 32 blocks (35.3 / 56.5 vs 33.7 / 55.1 us on
//      features.8 / .12 at batch 16) and slower at 64 x 64 (65 vs 51 us): the filter traffic is not what a chunk waits for either; kept
//      behind STARCOP_IRB_CFG=2 as the measured alternative
enum { IRB_A = 0, IRB_B = 1, IRB_C = 2 };
static inline int irb_pairs_per_wave(int cfg, int Cout) {
  const int CB = (Cout + 31) / 32;
  return cfg == IRB_A ? (CB + 3) / 4 : (cfg == IRB_B ? (2 * CB + 3) / 4 : (2 * CB + 7) / 8);
}
static inline size_t irb_cfg_lds(int cfg, int nks_e) {
  return cfg == IRB_A ? irb_smem_bytes<4, 8, 64>(nks_e) : (cfg == IRB_B ? irb_smem_bytes<8, 8, 32>(nks_e) : irb_smem_bytes<8, 8, 64>(nks_e));
}
static inline bool irb_cfg_ok(int cfg, int Cin, int hid, int Cout) {
  const int nks_e = (Cin + 15) / 16;
  if (cfg != IRB_B && hid % 64) return false;
  if (irb_cfg_lds(cfg, nks_e) > 160 * 1024) return false;
  if (cfg == IRB_C && nks_e > 6) return false;               // (8 waves: 256 registers per lane)
  const int pw = irb_pairs_per_wave(cfg, Cout);
  if (cfg == IRB_C) return pw <= 1 || (pw == 2 && nks_e <= 4);      // (two pairs per wave with Cin > 64 would spill)
  return pw <= 3;
}
static inline bool irb_occ2(int Cin, int Cout) {      // tiling A with two resident work-groups (one projection pair per wave: no spills at 256 registers)
  return irb_smem_bytes<4, 8, 64>((Cin + 15) / 16) <= 80 * 1024 && Cin <= 96 && irb_pairs_per_wave(IRB_A, Cout) <= 1;
}
static int irb_pick_cfg(int Cin, int hid, int Cout) {
  static const int forced = [] { const char* e = getenv("STARCOP_IRB_CFG"); return e ? atoi(e) : -1; }();      // development knob (A/B runs)
  if (forced >= 0 && forced <= 2 && irb_cfg_ok(forced, Cin, hid, Cout)) return forced;
  if (irb_cfg_ok(IRB_A, Cin, hid, Cout)) return IRB_A;      // (C measured level with A at 32 x 32, 25 % slower at 64 x 64 and on scenes)
  if (irb_cfg_ok(IRB_C, Cin, hid, Cout)) return IRB_C;
  return irb_cfg_ok(IRB_B, Cin, hid, Cout) ? IRB_B : -1;
}

// stride 2 (features.7 / .14): tiling A on the 9 x 17 input patch of a 4 x 8 output tile, one work-group per CU
static inline bool irb_s2_ok(int Cin, int hid, int Cout) {
  return hid % 64 == 0 && Cin <= 96 && irb_smem_bytes<4, 8, 64, 2>((Cin + 15) / 16) <= 160 * 1024 && irb_pairs_per_wave(IRB_A, Cout) <= 2;
}

}  // namespace

extern "C" int sc_irb_supported(int Cin, int hidden, int Cout, int H, int W, int stride) {
  if ((stride != 1 && stride != 2) || Cin < 8 || Cin > 160 || hidden < 32 || hidden % 32 || Cout < 1 || Cout > 384 || H < 1 || W < 1) return 0;
  if (stride == 2) return irb_s2_ok(Cin, hidden, Cout) ? 1 : 0;
  return irb_pick_cfg(Cin, hidden, Cout) >= 0 ? 1 : 0;
}

extern "C" int sc_irb_eval(const sc_irb_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_irb_eval: null args");
  SC_REQUIRE(a->stride == 1 || a->stride == 2, "sc_irb_eval: stride 1 or 2 (got %d)", a->stride);
  SC_REQUIRE(a->stride == 1 || !a->residual, "sc_irb_eval: a stride-2 block has no residual connection");
  SC_REQUIRE(sc_irb_supported(a->Cin, a->hidden, a->Cout, a->H, a->W, a->stride), "sc_irb_eval: unsupported block shape (Cin %d, hidden %d, Cout %d, %d x %d)",
             a->Cin, a->hidden, a->Cout, a->H, a->W);
  SC_REQUIRE(a->N > 0 && a->x.x && a->x.C == a->Cin && a->x.up == 0, "sc_irb_eval: bad input source");
  SC_REQUIRE(a->x.mode == SC_SRC_RAW || (a->x.mode == SC_SRC_AFFINE && a->x.cst), "sc_irb_eval: the input is a RAW or AFFINE source");
  SC_REQUIRE(a->wpk_expand && a->wpk_project && a->cst_expand && a->w_dw && a->cst_dw && a->out, "sc_irb_eval: null operand");
  SC_REQUIRE((((uintptr_t)a->wpk_expand | (uintptr_t)a->wpk_project) & 15) == 0, "sc_irb_eval: packed filters must be 16-byte aligned");
  SC_REQUIRE(!a->residual || (a->cst_project && a->Cin == a->Cout), "sc_irb_eval: a residual block needs BN_p's constants and Cin == Cout");
  SC_REQUIRE(((uintptr_t)a->out & 15) == 0 && ((uintptr_t)a->x.x & 15) == 0, "sc_irb_eval: tensors must be 16-byte aligned");
  IrbP p;
  p.x = to_srcd(a->x);
  p.we = reinterpret_cast<const uintx4*>(a->wpk_expand); p.wp = reinterpret_cast<const uintx4*>(a->wpk_project);
  p.cst_e = a->cst_expand; p.wdw = a->w_dw; p.cst_d = a->cst_dw; p.cst_p = a->cst_project;
  p.out = a->out; p.zmax = a->z_absmax;
  p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.hid = a->hidden; p.Cout = a->Cout;
  p.Ho = (a->H - 1) / a->stride + 1; p.Wo = (a->W - 1) / a->stride + 1;
  p.nks_e = (a->Cin + 15) / 16; p.nks_p = (a->hidden + 15) / 16; p.residual = a->residual ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (a->stride == 2) {
    p.tiles_x = (p.Wo + 7) / 8; p.tiles_y = (p.Ho + 3) / 4;
    return irb_pairs_per_wave(IRB_A, a->Cout) <= 1 ? irb_launch<4, 8, 64, 1, 1, 6, 2>(p, st) : irb_launch<4, 8, 64, 2, 1, 6, 2>(p, st);
  }
  const int cfg = irb_pick_cfg(a->Cin, a->hidden, a->Cout);
  const int mpp = irb_pairs_per_wave(cfg, a->Cout);
  p.tiles_x = (a->W + 7) / 8; p.tiles_y = cfg == IRB_A ? (a->H + 3) / 4 : (a->H + 7) / 8;
  if (cfg == IRB_A) {
    if (irb_occ2(a->Cin, a->Cout)) return irb_launch<4, 8, 64, 1, 2, 6>(p, st);
    if (mpp <= 1) return irb_launch<4, 8, 64, 1, 1, 10>(p, st);
    if (mpp == 2) return irb_launch<4, 8, 64, 2, 1, 10>(p, st);
    return irb_launch<4, 8, 64, 3, 1, 10>(p, st);
  }
  if (cfg == IRB_C) return mpp <= 1 ? irb_launch<8, 8, 64, 1, 1, 6>(p, st) : irb_launch<8, 8, 64, 2, 1, 6>(p, st);
  if (mpp <= 1) return irb_launch<8, 8, 32, 1, 1, 10>(p, st);
  if (mpp == 2) return irb_launch<8, 8, 32, 2, 1, 10>(p, st);
  return irb_launch<8, 8, 32, 3, 1, 10>(p, st);
}
