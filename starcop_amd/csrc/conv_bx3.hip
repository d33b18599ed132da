// Dense 3x3 NCHW conv2d (stride 1, pad 1) with fp32 accuracy on the bf16 matrix cores of gfx950.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate, so the decoder's 3x3 convolutions (88 % of the U-Net's MACs:
// smp.Unet built at starcop/models/model_module.py:244-251) are MFMA-bound on it.  Here every fp32 operand is split
// exactly into three bf16 terms  a = a0 + a1 + a2  (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1); 3 x 8
// significand bits) and the product is accumulated in fp32 from the six partial products whose weight is >= 2^-24:
//     a*b ~= a1*b1 + a2*b0 + a0*b2 + a1*b0 + a0*b1 + a0*b0
// = 6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 block (192 cycles) instead of 8 x v_mfma_f32_32x32x2_f32 (512 cycles).
// The dropped terms are O(2^-24 |a||b|), the same size as one fp32 rounding; measured against fp64 the result is as
// close as the fp32 MFMA path (tests/test_gpu_ops.py::test_conv_bx3_*).
//
// Same "normalise on load" contract as conv_mfma.hip: the producer's BatchNorm+activation (forward), or the
// BatchNorm/activation backward (dgrad), nearest x2 upsampling and the channel concat are applied while the tile is
// staged; the bf16 split happens in the same pass.  The same kernel computes dgrad from transposed+flipped filters.
//
// GEMM view:  D[co][pixel] = sum_{tap} sum_{ci} Wp[tap][co][ci] * patch[ci][pixel + d(tap)],  K step = 16 channels
//   A (32 x 16): lane l -> W[co = l&31][ci = 8*(l>>5) .. +7]      (one 16-byte LDS read)
//   B (16 x 32): lane l -> patch[ci = 8*(l>>5) .. +7][pixel l&31]  (one 16-byte LDS read)
//   D: col = l&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(l>>5) (cout)
// Work-group = 4 waves, tile = 8 rows x 32 cols x (32*Q couts); wave w owns rows 2w, 2w+1 and all Q cout blocks
// (2 x Q accumulators: every A read is used twice, every B read Q times -> 0.5 KB of LDS per MFMA at Q = 2).
// LDS: patch [3 terms][2 channel halves][10 x 34 pixels] x 16 B (single buffer, next chunk prefetched in registers),
// filters per filter row kh [3 terms][3 kw][2 halves][32Q couts] x 16 B, double buffered.  ~72 KB -> 2 work-groups / CU.
#include "sc_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) unsigned int uintx4;

struct ConvXP {
  SrcD s0, s1;
  const uintx4* wpk;
  int N, H, W, Cout;
  float* out0; float* out1;
  int csplit, accum0, accum1;
  const float* add0; const float* add1;
  float* stats;
};

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

template <int Q>
__global__ __launch_bounds__(256, 2) void k_conv3_bx3(const ConvXP p) {
  constexpr int PR = 10, PC = 34, NPX = PR * PC;     // 8 output rows + halo
  constexpr int CO_T = 32 * Q;
  constexpr int WENT = 18 * CO_T;                    // 16-byte filter entries per (chunk, kh) stage
  constexpr int NWV = (WENT + 255) / 256;
  constexpr int NR = 3;                              // staging rounds: 128 threads per channel half, 3 x 128 >= 340

  __shared__ uintx4 s_p[3][2][NPX];
  __shared__ uintx4 s_w[2][WENT];
  __shared__ float s_red[4][CO_T][2];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n = blockIdx.z, cot = blockIdx.y;
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * 8, x0 = tx * 32;
  const int C0 = p.s0.C;
  const int Cin = C0 + p.s1.C;
  const int nk = (Cin + 15) >> 4;                    // packed filters are zero-padded to nk*16 input channels
  const uintx4* wbase = p.wpk + (size_t)cot * nk * 3 * WENT;

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
  int off0[NR], off1[NR];             // clamped pixel offsets in source 0 / source 1 (they may differ in `up`)
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
      off0[r] = ok ? (y >> up0) * Ws0 + (x >> up0) : 0;
      off1[r] = ok ? (y >> up1) * Ws1 + (x >> up1) : 0;
      inb |= ok ? (1u << r) : 0u;
    }
  }
  float xv[NR][8], av[NR][8];
  uintx4 wv[NWV];
  // per-chunk source description (uniform)
  const float* xp = nullptr; const float* ap = nullptr; const float* cp = nullptr;
  int Cs = 0, smode = 0, cbase = 0; size_t plane = 0; bool second = false;
  float slo = 0.f, shi = 0.f;

  auto select_chunk = [&](int kc) {
    second = kc * 16 >= C0;
    xp = second ? p.s1.x : p.s0.x;
    ap = second ? p.s1.aux : p.s0.aux;
    cp = second ? p.s1.cst : p.s0.cst;
    Cs = second ? p.s1.C : p.s0.C;
    const int up = second ? p.s1.up : p.s0.up;
    smode = second ? p.s1.mode : p.s0.mode;
    const int sact = second ? p.s1.act : p.s0.act;
    slo = sc_act_lo(sact); shi = sc_act_hi(sact);
    plane = (size_t)(H >> up) * (W >> up);
    cbase = kc * 16 + hw * 8 - (second ? C0 : 0);            // first channel (source space) staged by this wave
  };
  auto load_round = [&](int r) {
    const int o = second ? off1[r] : off0[r];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cs = (cbase + j < Cs) ? cbase + j : 0;
      const size_t b = ((size_t)n * Cs + cs) * plane + o;
      xv[r][j] = xp[b];
      av[r][j] = (smode == SC_SRC_BNBWD) ? ap[b] : 0.f;
    }
  };
  auto store_patch = [&]() {
    float4 c0[8]; float c4[8]; bool chok[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      chok[j] = cbase + j < Cs;
      const int cs = chok[j] ? cbase + j : 0;
      if (smode != SC_SRC_RAW) {
        c0[j] = *reinterpret_cast<const float4*>(cp + (size_t)cs * SC_CST);
        c4[j] = cp[(size_t)cs * SC_CST + 4];
      } else {
        c0[j] = make_float4(1.f, 0.f, 0.f, 0.f); c4[j] = 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = sidx + 128 * r;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = (smode == SC_SRC_BNBWD)
                            ? sc_pro_bnbwd(xv[r][j], av[r][j], c0[j].x, c0[j].y, c0[j].z, c0[j].w, c4[j], slo, shi)
                            : sc_pro_affine(xv[r][j], c0[j].x, c0[j].y, slo, shi);
        v[j] = (((inb >> r) & 1u) && chok[j]) ? t : 0.f;
      }
      uintx4 t0, t1, t2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned a, b, c;
        split3x2(v[2 * j], v[2 * j + 1], a, b, c);
        t0[j] = a; t1[j] = b; t2[j] = c;
      }
      if (e < NPX) { s_p[0][hw][e] = t0; s_p[1][hw][e] = t1; s_p[2][hw][e] = t2; }
    }
  };
  auto load_w = [&](int s) {
    const uintx4* src = wbase + (size_t)s * WENT;
#pragma unroll
    for (int j = 0; j < NWV; ++j) {
      const int i = tid + 256 * j;
      wv[j] = src[i < WENT ? i : WENT - 1];
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NWV; ++j) {
      const int i = tid + 256 * j;
      if (i < WENT) s_w[buf][i] = wv[j];
    }
  };
  auto compute = [&](int kh, int buf) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      bf16x8 A[Q][3];
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          A[q][c] = __builtin_bit_cast(bf16x8, s_w[buf][((c * 3 + kw) * 2 + lhi) * CO_T + q * 32 + l31]);
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        bf16x8 B[3];
        const int e = (2 * wave + pp + kh) * PC + l31 + kw;
#pragma unroll
        for (int c = 0; c < 3; ++c) B[c] = __builtin_bit_cast(bf16x8, s_p[c][lhi][e]);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          floatx16 d = acc[pp][q];
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q][1], B[1], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q][2], B[0], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q][0], B[2], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q][1], B[0], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q][0], B[1], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q][0], B[0], d, 0, 0, 0);
          acc[pp][q] = d;
        }
      }
    }
  };

  // ---- pipeline ----
  select_chunk(0);
#pragma unroll
  for (int r = 0; r < NR; ++r) load_round(r);
  load_w(0);
  store_patch();
  store_w(0);
  __syncthreads();
  const int nst = 3 * nk;
  for (int kc = 0; kc < nk; ++kc) {
    const bool more = (kc + 1) < nk;
    if (more) select_chunk(kc + 1);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int s = 3 * kc + kh;
      const bool next = (s + 1) < nst;
      if (next) load_w(s + 1);
      if (more) load_round(kh);
      compute(kh, s & 1);
      if (next) store_w((s + 1) & 1);
      __syncthreads();
    }
    if (more) {
      store_patch();
      __syncthreads();
    }
  }

  // ---- epilogue ----
  const size_t HWs = (size_t)H * W;
  const bool want_stats = p.stats != nullptr;
  const int ox = x0 + l31;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int co = cot * CO_T + col;
      float sv = 0.f, sq = 0.f;
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int oy = y0 + 2 * wave + pp;
        const bool ok = (oy < H) && (ox < W) && (co < p.Cout);
        float v = ok ? acc[pp][q][r] : 0.f;
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
__global__ void k_pack_weights_bx3(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout, int Cin,
                                   int co_t, int tflip, int nchunk, size_t total) {
  const int M = tflip ? Cin : Cout, K = tflip ? Cout : Cin;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int j = (int)(r % 8); r /= 8;
    const int col = (int)(r % co_t); r /= co_t;
    const int half = (int)(r % 2); r /= 2;
    const int kw = (int)(r % 3); r /= 3;
    const int kh = (int)(r % 3); r /= 3;
    const int chunk = (int)(r % nchunk);
    const int mt = (int)(r / nchunk);
    const int m = mt * co_t + col, k = chunk * 16 + half * 8 + j, tap = kh * 3 + kw;
    float v = 0.f;
    if (m < M && k < K) v = tflip ? w[((size_t)k * M + m) * 9 + (8 - tap)] : w[((size_t)m * K + k) * 9 + tap];
    const __bf16 t0 = (__bf16)v;
    float rr = v - (float)t0;
    const __bf16 t1 = (__bf16)rr;
    rr -= (float)t1;
    const __bf16 t2 = (__bf16)rr;
    const size_t stage = ((size_t)mt * nchunk + chunk) * 3 + kh;
    const __bf16 t[3] = {t0, t1, t2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t d = ((((stage * 3 + c) * 3 + kw) * 2 + half) * co_t + col) * 8 + j;
      wpk[d] = __builtin_bit_cast(unsigned short, t[c]);
    }
  }
}

}  // namespace

extern "C" size_t sc_packed_weight_floats_bx3(int Cout, int Cin, int co_t, int transpose_flip) {
  const int M = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  const size_t mt = (M + co_t - 1) / co_t, nchunk = (K + 15) / 16;
  return mt * nchunk * 3 * 18 * (size_t)co_t * 4;     // 16-byte entries -> floats
}

extern "C" int sc_pack_weights_bx3(const float* w, float* wpk, int Cout, int Cin, int co_t, int transpose_flip,
                                   sc_stream stream) {
  SC_REQUIRE(w && wpk && Cout > 0 && Cin > 0, "sc_pack_weights_bx3: bad argument");
  SC_REQUIRE(co_t == 32 || co_t == 64, "sc_pack_weights_bx3: co_t must be 32 or 64 (got %d)", co_t);
  SC_REQUIRE(((uintptr_t)wpk & 15) == 0, "sc_pack_weights_bx3: destination must be 16-byte aligned");
  const int M = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  const int nchunk = (K + 15) / 16;
  const size_t total = (size_t)((M + co_t - 1) / co_t) * nchunk * 9 * 2 * co_t * 8;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(k_pack_weights_bx3, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<unsigned short*>(wpk), Cout, Cin, co_t, transpose_flip, nchunk, total);
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
  }
  ConvXP p;
  p.s0 = to_srcd(a->src[0]);
  p.s1 = a->nsrc == 2 ? to_srcd(a->src[1]) : empty_srcd();
  p.wpk = reinterpret_cast<const uintx4*>(a->wpk); p.N = a->N; p.H = a->H; p.W = a->W; p.Cout = a->Cout;
  p.out0 = a->out0; p.out1 = a->out1; p.csplit = a->csplit; p.accum0 = a->accum0; p.accum1 = a->accum1;
  p.add0 = a->add0; p.add1 = a->add1; p.stats = a->stats;
  const int co_tiles = (a->Cout + a->co_t - 1) / a->co_t;
  dim3 grid(((a->W + 31) / 32) * ((a->H + 7) / 8), co_tiles, a->N);
  hipStream_t st = (hipStream_t)stream;
  if (a->co_t == 64) hipLaunchKernelGGL((k_conv3_bx3<2>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_conv3_bx3<1>), grid, dim3(256), 0, st, p);
  SC_LAUNCH_OK("sc_conv3x3_bx3");
  return SC_OK;
}
