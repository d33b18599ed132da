// Weight gradient of the decoder's conv1 layers w.r.t. the filters of the UP-SAMPLED channels as a plain GEMM.
//
// smp's DecoderBlock runs conv3x3(cat([F.interpolate(prev, x2, nearest), skip])) (starcop/models/model_module.py:244-251).  For an
// up-sampled channel the input is constant over 2x2 blocks of the output grid, so
//   dW[co][ci][kh][kw] = sum_p dy[co][p] * up(x)[ci][p + (kh-1, kw-1)]
//                      = sum_q x[ci][q] * S_(kh,kw)[co][q],     S_(kh,kw)[q] = sum_{d in {0,1}^2} dy[2q + (1-kh, 1-kw) + d]   (zero outside)
// -- the nine taps are nine PLAIN GEMMs over the low-resolution pixels q between x and tap-aligned 2x2 box sums of dy: no spatial
// shift of an operand, a quarter of the multiply-adds of the 3x3 form (k_wgrad3_bx3 with an up-sampled source: 434 / 187 / 221 / 355 us
// on decoder.blocks.0-3 at 16 x 512^2, all of it felt by the step: without any weight-gradient launch the step takes 8.55 instead of
// 11.47 ms although they run on their own stream -- DESIGN.md section 15).
//   1. k_spw_dysum : (g, y) of the layer's output -> BatchNorm / activation backward on load -> the nine box sums, split exactly into
//                    two fp16 terms (the arithmetic of conv_bx3.hip; the power-of-two gradient scale / 4 for the four addends)
//                    -> S0, S1 [9 * Cout][Kp] fp16, K-contiguous (K = N * Hl * Wl low-resolution pixels)
//   2. k_spw_xsplit: prev -> BatchNorm + ReLU on load -> two fp16 terms -> X0, X1 [Cup][Kp]
//   3. k_spw_gemm  : C[r][c] = sum_k S0 X1 + S1 X0 + S0 X0 on v_mfma_f32_32x32x16_f16, operands staged by straight 16-byte copies (no
//                    prologue, no split in the loop), 256 x 128 tiles, K slices -> partials [slice][9 * Cout][Cup]
//   4. k_spw_reduce: fixed-order sum of the slices (bit-reproducible) -> dw[co][ci][tap] inside the layer's [Cout][Cin][3][3] gradient
// The skip channels' columns of dw come from the 3x3 kernel on the skip source alone (sc_wgrad_scatter_cols puts them in place).
#include "sc_common.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) unsigned int uintx4;
typedef __attribute__((ext_vector_type(8))) _Float16 halfx8;
typedef __attribute__((ext_vector_type(2))) _Float16 halfx2;

constexpr float SPW_HMAX = 65504.f;

__device__ __forceinline__ void spw_split(float a, unsigned short& h0, unsigned short& h1) {
  a = __builtin_amdgcn_fmed3f(a, -SPW_HMAX, SPW_HMAX);
  const _Float16 t0 = (_Float16)a;
  const _Float16 t1 = (_Float16)(a - (float)t0);
  h0 = __builtin_bit_cast(unsigned short, t0); h1 = __builtin_bit_cast(unsigned short, t1);
}

// ---- 1. box sums of dy --------------------------------------------------------------------------------------------------
// block = (n, co, 8 x 32 low-resolution tile): the processed 18 x 66 high-resolution patch of dy through LDS, nine sums per pixel
struct SpwDy {
  SrcD dy;                  // SC_SRC_BNBWD (g, y, constants) or RAW / AFFINE; [N][Cout][2 Hl][2 Wl]
  const float* absmax;
  unsigned short* S0; unsigned short* S1;
  float* scal;              // scal[0] <- the gradient operand scale
  int N, Cout, Hl, Wl;
  long Kp;
};

__device__ __forceinline__ float spw_grad_scale(const float* absmax) {      // h_grad_scale of conv_bx3.hip, / 4 for the box sum
  const float M = absmax ? *absmax : 0.f;
  if (!(M > 0.f) || !(M < 3.0e38f)) return 0.25f;
  int e;
  (void)frexpf(M, &e);
  e = 3 - e;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.f, e);
}

__global__ __launch_bounds__(256) void k_spw_dysum(const SpwDy p) {
  constexpr int TH = 8, TW = 32, PH = 2 * TH + 2, PW = 2 * TW + 2;
  __shared__ float s_d[PH][PW + 1];
  const int tid = threadIdx.x;
  const int tiles_x = (p.Wl + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int co = blockIdx.y, n = blockIdx.z;
  const int H = 2 * p.Hl, W = 2 * p.Wl;
  const float s = spw_grad_scale(p.absmax);
  if (blockIdx.x == 0 && co == 0 && n == 0 && tid == 0) p.scal[0] = s;
  const size_t plane = (size_t)H * W;
  const float* gb = p.dy.x + ((size_t)n * p.Cout + co) * plane;
  const float* yb = p.dy.aux ? p.dy.aux + ((size_t)n * p.Cout + co) * plane : nullptr;
  float c0 = 1.f, c1 = 0.f, cA = 1.f, cB = 0.f, cD = 0.f;
  if (p.dy.cst && p.dy.mode != SC_SRC_RAW) {
    const float* c = p.dy.cst + (size_t)co * SC_CST;
    c0 = c[0]; c1 = c[1]; cA = c[2]; cB = c[3]; cD = c[4];
  }
  const float lo = sc_act_lo(p.dy.act), hi = sc_act_hi(p.dy.act);
  const int y0 = 2 * ty * TH - 1, x0 = 2 * tx * TW - 1;
  for (int e = tid; e < PH * PW; e += 256) {
    const int r = e / PW, c = e - r * PW;
    const int y = y0 + r, x = x0 + c;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const size_t o = (size_t)y * W + x;
      const float g = gb[o];
      if (p.dy.mode == SC_SRC_BNBWD) v = sc_pro_bnbwd(g, yb[o], c0, c1, cA, cB, cD, lo, hi);
      else if (p.dy.mode == SC_SRC_AFFINE) v = sc_pro_affine(g, c0, c1, lo, hi);
      else v = g;
      v *= s;
    }
    s_d[r][c] = v;
  }
  __syncthreads();
  const int qi = tid >> 5, qj = tid & 31;
  const int i = ty * TH + qi, j = tx * TW + qj;
  if (i >= p.Hl || j >= p.Wl) return;
  // window rows / columns 0..3 = high-resolution 2q-1 .. 2q+2; tap kh sums rows (2 - kh, 3 - kh) of the window
  float w[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) w[r][c] = s_d[2 * qi + r][2 * qj + c];
  const size_t k = ((size_t)n * p.Hl + i) * p.Wl + j;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    float rs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) rs[c] = w[2 - kh][c] + w[3 - kh][c];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float v = rs[2 - kw] + rs[3 - kw];
      unsigned short h0, h1;
      spw_split(v, h0, h1);
      const size_t row = (size_t)(kh * 3 + kw) * p.Cout + co;
      p.S0[row * p.Kp + k] = h0;
      p.S1[row * p.Kp + k] = h1;
    }
  }
}

// zero the K padding [K, Kp) of a [rows][Kp] fp16 pair (the GEMM reads whole 32-pixel stages)
__global__ void k_spw_zero_tail(unsigned short* A0, unsigned short* A1, long rows, long K, long Kp) {
  const long n = rows * (Kp - K);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (Kp - K), c = K + i % (Kp - K);
    A0[r * Kp + c] = 0; A1[r * Kp + c] = 0;
  }
}

// ---- 2. the low-resolution source, activated and split ---------------------------------------------------------------------
struct SpwX {
  SrcD x;                   // [N][Cup][Hl][Wl], RAW or AFFINE
  const float* xb;          // activation bound (h_act_scale) or NULL
  unsigned short* X0; unsigned short* X1;
  float* scal;              // scal[1] <- the activation operand scale
  int N, Cup, HW;
  long Kp;
};
__device__ __forceinline__ float spw_act_scale(const float* xb) {
  float M = xb ? *xb : 0.f;
  if (!(M * 2.f > 32752.f)) return 2.f;
  M = fminf(M, 3.0e38f);
  int e;
  (void)frexpf(32752.f / M, &e);
  e = e - 1 < -120 ? -120 : e - 1;
  return ldexpf(1.f, e);
}
__global__ __launch_bounds__(256) void k_spw_xsplit(const SpwX p) {
  const float s = spw_act_scale(p.xb);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.scal[1] = s;
  const int ci = blockIdx.y;
  float sc = 1.f, sh = 0.f;
  if (p.x.cst && p.x.mode != SC_SRC_RAW) { sc = p.x.cst[(size_t)ci * SC_CST]; sh = p.x.cst[(size_t)ci * SC_CST + 1]; }
  const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
  const long K = (long)p.N * p.HW;
  for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < K; k += (long)gridDim.x * blockDim.x) {
    const long n = k / p.HW, q = k - n * p.HW;
    const float v = sc_pro_affine(p.x.x[((size_t)n * p.Cup + ci) * p.HW + q], sc, sh, lo, hi) * s;
    unsigned short h0, h1;
    spw_split(v, h0, h1);
    p.X0[(size_t)ci * p.Kp + k] = h0;
    p.X1[(size_t)ci * p.Kp + k] = h1;
  }
}

// ---- 3. the GEMM ---------------------------------------------------------------------------------------------------------------
// C[r][c] += sum_k A0[r][k] B1[c][k] + A1[r][k] B0[c][k] + A0[r][k] B0[c][k];  A = S (M rows), B = X (Nc rows), both K-contiguous fp16.
// Work-group = 8 waves = 4 (rows) x 2 (columns); tile 256 x (64 * CB); wave = 2 x CB blocks of 32 x 32.  K stage = 32: LDS rows of
// 4 entries (8 values each) at a pitch of 5 (groups of 8 lanes then touch all 32 banks once), double-buffered, one barrier per stage;
// the next stage's six 16-byte global loads per thread are in flight during the MFMAs.
struct SpwG {
  const unsigned short* A0; const unsigned short* A1; const unsigned short* B0; const unsigned short* B1;
  const float* scal;
  float* part;              // [nsl][Mp][Np]
  int M, Nc, Mp, Np;
  long Kp; int kstages;     // stages of 32 per slice
};

template <int CB>
__global__ __launch_bounds__(512, 2) void k_spw_gemm(const SpwG p) {
  constexpr int MT = 256, NT = 64 * CB, P = 5;
  constexpr int AE = MT * 4, BE = NT * 4;                       // 16-byte entries per term and stage
  constexpr int NA = 2 * AE / 512, NB = 2 * BE / 512;           // entries per thread: both terms
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uintx4* const s_a = reinterpret_cast<uintx4*>(smem);                              // [2 buf][2 terms][MT * P]
  uintx4* const s_b = s_a + 2 * 2 * MT * P;                                          // [2 buf][2 terms][NT * P]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int mt = blockIdx.y, nt = blockIdx.z, sl = blockIdx.x;
  const long k0 = (long)sl * p.kstages * 32;
  const long kend = k0 + (long)p.kstages * 32 < p.Kp ? k0 + (long)p.kstages * 32 : p.Kp;
  const int nst = (int)((kend - k0) >> 5);

  // staging map: entry e of a term -> (row e >> 2, piece e & 3); rows beyond M / Nc re-read the last row (masked in the epilogue)
  const unsigned short* ga[NA]; const unsigned short* gb[NB];
  int la[NA], lb[NB];
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    const int e = tid + 512 * u, term = e / AE, f = e - term * AE, row = f >> 2, pc = f & 3;
    const int gr = mt * MT + row < p.M ? mt * MT + row : p.M - 1;
    ga[u] = (term ? p.A1 : p.A0) + (size_t)gr * p.Kp + k0 + pc * 8;
    la[u] = term * (MT * P) + row * P + pc;
  }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int e = tid + 512 * u, term = e / BE, f = e - term * BE, row = f >> 2, pc = f & 3;
    const int gr = nt * NT + row < p.Nc ? nt * NT + row : p.Nc - 1;
    gb[u] = (term ? p.B1 : p.B0) + (size_t)gr * p.Kp + k0 + pc * 8;
    lb[u] = term * (NT * P) + row * P + pc;
  }
  uintx4 ra[NA], rb[NB];
  auto gload = [&](int s) {
#pragma unroll
    for (int u = 0; u < NA; ++u) ra[u] = *reinterpret_cast<const uintx4*>(ga[u] + (size_t)s * 32);
#pragma unroll
    for (int u = 0; u < NB; ++u) rb[u] = *reinterpret_cast<const uintx4*>(gb[u] + (size_t)s * 32);
  };
  auto lstore = [&](int buf) {
    uintx4* const sa = s_a + buf * (2 * MT * P);
    uintx4* const sb = s_b + buf * (2 * NT * P);
#pragma unroll
    for (int u = 0; u < NA; ++u) sa[la[u]] = ra[u];
#pragma unroll
    for (int u = 0; u < NB; ++u) sb[lb[u]] = rb[u];
  };

  floatx16 acc[2][CB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  lstore(0);
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    const int buf = s & 1;
    gload(s + 1 < nst ? s + 1 : s);                        // (the last stage re-requests itself: loads stay unconditional)
    __builtin_amdgcn_sched_barrier(0);
    const uintx4* const sa = s_a + buf * (2 * MT * P);
    const uintx4* const sb = s_b + buf * (2 * NT * P);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      halfx8 A[2][2], B[CB][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) A[i][c] = __builtin_bit_cast(halfx8, sa[c * (MT * P) + (wm * 64 + i * 32 + l31) * P + ks * 2 + lhi]);
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) B[j][c] = __builtin_bit_cast(halfx8, sb[c * (NT * P) + (wn * 32 * CB + j * 32 + l31) * P + ks * 2 + lhi]);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i][t == 1 ? 1 : 0], B[j][t == 0 ? 1 : 0], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: D row = (r & 3) + 8 (r >> 2) + 4 lhi -> GEMM row (S row), D column l31 -> GEMM column (input channel)
  const float hinv = 1.f / (p.scal[0] * p.scal[1]);
  float* const pb = p.part + (size_t)sl * p.Mp * p.Np;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int col = nt * NT + wn * 32 * CB + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * MT + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < p.M && col < p.Nc) pb[(size_t)row * p.Np + col] = acc[i][j][r] * hinv;
      }
    }
}

// ---- 4. dw[co][col_off + ci][tap] = sum_slices part[s][tap * Cout + co][ci] ----------------------------------------------------------
__global__ void k_spw_reduce(const float* __restrict__ part, float* __restrict__ dw, int nsl, int Cout, int Cup, int CinTot, int Mp, int Np) {
  const size_t total = (size_t)9 * Cout * Cup;
  const size_t stride = (size_t)Mp * Np;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cup);
    const size_t r = i / Cup;                 // tap * Cout + co
    const int co = (int)(r % Cout), tap = (int)(r / Cout);
    const float* src = part + r * Np + ci;
    float s = 0.f;
    for (int k = 0; k < nsl; ++k) s += src[(size_t)k * stride];
    dw[((size_t)co * CinTot + ci) * 9 + tap] = s;
  }
}

__global__ void k_scatter_cols(const float* __restrict__ src, float* __restrict__ dw, int Cout, int Cc, int CinTot, int off) {
  const size_t total = (size_t)Cout * Cc * 9;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t t = i % 9, c = (i / 9) % Cc, co = i / (9 * (size_t)Cc);
    dw[(co * CinTot + off + c) * 9 + t] = src[i];
  }
}

struct SpwPlan { long K, Kp; int M, Mp, Np, cb, mtiles, ntiles, nsl, kstages; size_t off_s1, off_x0, off_x1, off_scal, off_part, bytes; };
SpwPlan spw_plan(int N, int H, int W, int Cout, int Cup) {
  SpwPlan pl;
  pl.K = (long)N * (H / 2) * (W / 2);
  pl.Kp = (pl.K + 31) / 32 * 32;
  pl.M = 9 * Cout;
  pl.cb = Cup > 64 ? 2 : 1;
  const int NT = 64 * pl.cb;
  pl.mtiles = (pl.M + 255) / 256; pl.ntiles = (Cup + NT - 1) / NT;
  pl.Mp = pl.mtiles * 256; pl.Np = pl.ntiles * NT;
  const long stages = pl.Kp / 32;
  long want = (768 + (long)pl.mtiles * pl.ntiles - 1) / ((long)pl.mtiles * pl.ntiles);      // ~3 work-groups per CU in flight
  if (want > 64) want = 64;
  if (want > stages) want = stages;
  if (want < 1) want = 1;
  pl.kstages = (int)((stages + want - 1) / want);
  pl.nsl = (int)((stages + pl.kstages - 1) / pl.kstages);
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t sS = al((size_t)pl.M * pl.Kp * 2), sX = al((size_t)Cup * pl.Kp * 2);
  pl.off_s1 = sS; pl.off_x0 = 2 * sS; pl.off_x1 = 2 * sS + sX; pl.off_scal = 2 * sS + 2 * sX; pl.off_part = pl.off_scal + 256;
  pl.bytes = pl.off_part + (size_t)pl.nsl * pl.Mp * pl.Np * 4;
  return pl;
}

}  // namespace

extern "C" size_t sc_sp_wgrad_workspace_bytes(int N, int H, int W, int Cout, int Cup) {
  if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || Cout <= 0 || Cup <= 0) return 0;
  return spw_plan(N, H, W, Cout, Cup).bytes;
}

extern "C" int sc_conv3x3_sp_wgrad(const sc_wgrad_args* a, void* ws, size_t ws_bytes, sc_stream stream) {
  SC_REQUIRE(a != nullptr && ws != nullptr, "sc_conv3x3_sp_wgrad: null argument");
  SC_REQUIRE(a->ks == 3 && a->nsrc == 1 && a->src[0].up == 1, "sc_conv3x3_sp_wgrad: ks = 3, one source: the half-resolution tensor (up = 1)");
  SC_REQUIRE(a->src[0].mode == SC_SRC_RAW || a->src[0].mode == SC_SRC_AFFINE, "sc_conv3x3_sp_wgrad: the input source must be RAW or AFFINE");
  SC_REQUIRE(a->src[0].mode == SC_SRC_RAW || a->src[0].cst != nullptr, "sc_conv3x3_sp_wgrad: the input source needs constants");
  SC_REQUIRE(a->dy.C == a->Cout && a->dy.up == 0 && a->dy.mode != SC_SRC_NORM, "sc_conv3x3_sp_wgrad: unsupported dy source");
  SC_REQUIRE(a->dy.mode != SC_SRC_BNBWD || (a->dy.aux != nullptr && a->dy.cst != nullptr), "sc_conv3x3_sp_wgrad: BNBWD dy needs aux and constants");
  SC_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->H % 2 == 0 && a->W % 2 == 0 && a->Cout > 0, "sc_conv3x3_sp_wgrad: bad shape (even H, W)");
  const int Cup = a->src[0].C;
  SC_REQUIRE(Cup > 0 && Cup <= a->Cin, "sc_conv3x3_sp_wgrad: Cin is the filter's total input channel count (>= the source's %d)", Cup);
  SC_REQUIRE(a->terms == SC_TERMS_F16X2, "sc_conv3x3_sp_wgrad: two-fp16-term arithmetic only (terms = SC_TERMS_F16X2)");
  SC_REQUIRE(((uintptr_t)ws & 255) == 0, "sc_conv3x3_sp_wgrad: workspace must be 256-byte aligned");
  const SpwPlan pl = spw_plan(a->N, a->H, a->W, a->Cout, Cup);
  SC_REQUIRE(ws_bytes >= pl.bytes, "sc_conv3x3_sp_wgrad: workspace too small (%zu < %zu bytes)", ws_bytes, pl.bytes);
  hipStream_t st = (hipStream_t)stream;
  unsigned char* w8 = reinterpret_cast<unsigned char*>(ws);
  unsigned short* S0 = reinterpret_cast<unsigned short*>(w8), *S1 = reinterpret_cast<unsigned short*>(w8 + pl.off_s1);
  unsigned short* X0 = reinterpret_cast<unsigned short*>(w8 + pl.off_x0), *X1 = reinterpret_cast<unsigned short*>(w8 + pl.off_x1);
  float* scal = reinterpret_cast<float*>(w8 + pl.off_scal);
  float* part = reinterpret_cast<float*>(w8 + pl.off_part);
  const int Hl = a->H / 2, Wl = a->W / 2;
  {
    SpwDy p; p.dy = to_srcd(a->dy); p.absmax = a->absmax; p.S0 = S0; p.S1 = S1; p.scal = scal;
    p.N = a->N; p.Cout = a->Cout; p.Hl = Hl; p.Wl = Wl; p.Kp = pl.Kp;
    dim3 grid(((Wl + 31) / 32) * ((Hl + 7) / 8), a->Cout, a->N);
    hipLaunchKernelGGL(k_spw_dysum, grid, dim3(256), 0, st, p);
    SC_LAUNCH_OK("sc_conv3x3_sp_wgrad(dysum)");
  }
  {
    SpwX p; p.x = to_srcd(a->src[0]); p.xb = a->xbound[0]; p.X0 = X0; p.X1 = X1; p.scal = scal;
    p.N = a->N; p.Cup = Cup; p.HW = Hl * Wl; p.Kp = pl.Kp;
    long bx = (pl.K + 255) / 256; if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_spw_xsplit, dim3((unsigned)bx, Cup), dim3(256), 0, st, p);
    SC_LAUNCH_OK("sc_conv3x3_sp_wgrad(xsplit)");
  }
  if (pl.Kp > pl.K) {
    hipLaunchKernelGGL(k_spw_zero_tail, dim3(64), dim3(256), 0, st, S0, S1, (long)pl.M, pl.K, pl.Kp);
    hipLaunchKernelGGL(k_spw_zero_tail, dim3(64), dim3(256), 0, st, X0, X1, (long)Cup, pl.K, pl.Kp);
    SC_LAUNCH_OK("sc_conv3x3_sp_wgrad(zero tail)");
  }
  {
    SpwG p; p.A0 = S0; p.A1 = S1; p.B0 = X0; p.B1 = X1; p.scal = scal; p.part = part;
    p.M = pl.M; p.Nc = Cup; p.Mp = pl.Mp; p.Np = pl.Np; p.Kp = pl.Kp; p.kstages = pl.kstages;
    dim3 grid(pl.nsl, pl.mtiles, pl.ntiles);
    constexpr int lds2 = 2 * 2 * (256 + 128) * 5 * 16, lds1 = 2 * 2 * (256 + 64) * 5 * 16;
    static const bool attr_ok = [] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spw_gemm<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spw_gemm<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1) == hipSuccess;
    }();
    SC_REQUIRE(attr_ok, "sc_conv3x3_sp_wgrad: cannot reserve %d bytes of LDS", lds2);
    if (pl.cb == 2) hipLaunchKernelGGL((k_spw_gemm<2>), grid, dim3(512), lds2, st, p);
    else hipLaunchKernelGGL((k_spw_gemm<1>), grid, dim3(512), lds1, st, p);
    SC_LAUNCH_OK("sc_conv3x3_sp_wgrad(gemm)");
  }
  {
    const size_t total = (size_t)9 * a->Cout * Cup;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(k_spw_reduce, dim3(blocks), dim3(256), 0, st, part, a->dw, pl.nsl, a->Cout, Cup, a->Cin, pl.Mp, pl.Np);
    SC_LAUNCH_OK("sc_conv3x3_sp_wgrad(reduce)");
  }
  return SC_OK;
}

extern "C" int sc_wgrad_scatter_cols(const float* src, float* dw, int Cout, int Ccols, int CinTotal, int col_off, sc_stream stream) {
  SC_REQUIRE(src && dw && Cout > 0 && Ccols > 0 && col_off >= 0 && col_off + Ccols <= CinTotal, "sc_wgrad_scatter_cols: bad argument");
  const size_t total = (size_t)Cout * Ccols * 9;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(k_scatter_cols, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dw, Cout, Ccols, CinTotal, col_off);
  SC_LAUNCH_OK("sc_wgrad_scatter_cols");
  return SC_OK;
}
