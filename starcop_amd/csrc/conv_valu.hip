// HBM-bound convolutions of the MobileNetV2-U-Net that do not belong on the matrix cores:
//   depthwise 3x3 (stride 1|2)         -- torchvision InvertedResidual conv.{0|1}.0
//   stem 3x3 stride 2, Cin(<=8) -> 32  -- encoder.features.0.0 (+ DataNormalizer fused on load)
//   head 3x3, Cin(16) -> 1, bias       -- segmentation_head.0
// forward / backward-data / backward-weight, all with the "normalise on load" prologues of
// sc_common.h.  Reference call site of the whole network: starcop/models/model_module.py:244-251.
#include <type_traits>
#include "sc_common.h"
#ifndef SC_HEAD_RB
#define SC_HEAD_RB 18     // patch rows (x2 loads) a wave keeps in flight while staging the head forward tile
#endif

namespace {

__device__ __forceinline__ float load_src(const SrcD& s, size_t idx, int c) {
  const float x = s.x[idx];
  if (s.mode == SC_SRC_RAW) return x;
  const float4 c0 = *reinterpret_cast<const float4*>(s.cst + (size_t)c * SC_CST);
  const float c4 = s.cst[(size_t)c * SC_CST + 4];
  const float au = (s.mode == SC_SRC_BNBWD) ? s.aux[idx] : 0.f;
  return sc_prologue(s.mode, s.act, x, au, c0, c4);
}

// block-wide sum of up to NV values per thread; result valid in thread 0.. (returned to all of wave 0)
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* s_tmp /* [4][NV] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float s = wave_sum(v[k]);
    if (lane == 0) s_tmp[wave * NV + k] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = s_tmp[k] + s_tmp[NV + k] + s_tmp[2 * NV + k] + s_tmp[3 * NV + k];
}

// ---------------------------------------------------------------- depthwise
// One (n, c) plane tile per block: TW x (TB*R) outputs (TB = 256/TW rows per pass, R passes), R outputs per thread.
// The input patch is staged ONCE through LDS with the producer's BatchNorm+ReLU6 applied on the way (branch-free
// prologue, clamped unconditional loads issued SC_DW_LB at a time), so every HBM element is read once per tile and
// each thread keeps R stores + ~R*S*S loads in flight: these kernels are HBM-latency bound with small tiles.
constexpr int SC_DW_LB = 8;

template <int PH, int PW, int PWP, typename F>
__device__ __forceinline__ void dw_stage(float* s, const float* __restrict__ xb, int y0, int x0, int Hin, int Win, F&& pro) {
  constexpr int NE = PH * PW, NIT = (NE + 255) / 256;
#pragma unroll
  for (int i0 = 0; i0 < NIT; i0 += SC_DW_LB) {
    float v[SC_DW_LB];
#pragma unroll
    for (int j = 0; j < SC_DW_LB; ++j) {
      if (i0 + j < NIT) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / PW, cc = e - r * PW;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (e < NE) && (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
        v[j] = xb[ok ? iy * Win + ix : 0];
      }
    }
#pragma unroll
    for (int j = 0; j < SC_DW_LB; ++j) {
      if (i0 + j < NIT) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / PW, cc = e - r * PW;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
        if (e < NE) s[r * PWP + cc] = ok ? pro(v[j]) : 0.f;
      }
    }
  }
}

// same, for a BatchNorm-backward source: two tensors (g, y) per element
template <int PH, int PW, int PWP, typename F>
__device__ __forceinline__ void dw_stage2(float* s, const float* __restrict__ gb, const float* __restrict__ yb, int y0, int x0,
                                          int Hin, int Win, F&& pro) {
  constexpr int NE = PH * PW, NIT = (NE + 255) / 256, LB = SC_DW_LB / 2;
#pragma unroll
  for (int i0 = 0; i0 < NIT; i0 += LB) {
    float g[LB], yv[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      if (i0 + j < NIT) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / PW, cc = e - r * PW;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (e < NE) && (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
        const int o = ok ? iy * Win + ix : 0;
        g[j] = gb[o]; yv[j] = yb[o];
      }
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      if (i0 + j < NIT) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / PW, cc = e - r * PW;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
        if (e < NE) s[r * PWP + cc] = ok ? pro(g[j], yv[j]) : 0.f;
      }
    }
  }
}

// V4: float4 staging of the TW * S aligned interior columns (see k_dw_bwd), interior at LDS column 4
template <int S, int TW, int R, bool V4>
__global__ __launch_bounds__(256) void k_dw_fwd(const SrcD in, const float* __restrict__ w, float* __restrict__ out,
                                                int NC, int C, int Hin, int Win, int Hout, int Wout, float* stats, const BnTailD tail) {
  constexpr int TB = 256 / TW, TH = TB * R;
  constexpr int CO = V4 ? 3 : 0, IC = TW * S;
  constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PWP = V4 ? IC + 8 : (PW | 1);
  __shared__ __attribute__((aligned(16))) float s_x[PH * PWP];
  __shared__ float s_tmp[8];
  // XCD-aware numbering (see k_dw_bwd): the tiles of one (n, c) plane are 8 apart in the linear work-group id -> one XCD, one L2
  const int tiles_x = (Wout + TW - 1) / TW;
  const int tiles = tiles_x * ((Hout + TH - 1) / TH);
  const int grp = blockIdx.x / (8 * tiles), within = blockIdx.x - grp * 8 * tiles;
  const int plane = grp * 8 + (within & 7), tile = within >> 3;
  if (plane >= NC) return;
  const int n = plane / C, c = plane - n * C;
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
  float sc = 1.f, sh = 0.f;          // (requested after the first batch of patch loads, see k_dw_bwd)
  const float lo = sc_act_lo(in.act), hi = sc_act_hi(in.act);
  const float* xb = in.x + ((size_t)n * C + c) * Hin * Win;
  // stride 1: every load of the patch is in flight before the first use (one memory round trip per work-group)
  constexpr int NE = PH * PW, NIT = (NE + 255) / 256, LB = (S == 1) ? NIT : SC_DW_LB;   // measured: stride-2 patches (33 loads per thread) are faster in batches of 8
  const int y0 = ty0 * S - 1, x0 = tx0 * S - 1;
  float wk[9];
  if constexpr (V4) {
    constexpr int IC4 = IC / 4, NI4 = (PH * IC4 + 255) / 256, NHC = (S == 1) ? 2 : 1;     // halo columns: left (+ right at stride 1)
    static_assert(NHC * PH <= 256, "one halo element per thread");
    float4 v[NI4];
#pragma unroll
    for (int i = 0; i < NI4; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / IC4, c4 = e % IC4;
      const int iy = y0 + r, ix = x0 + 1 + 4 * c4;
      const bool ok = (r < PH) && (iy >= 0) && (iy < Hin) && (ix < Win);
      v[i] = *reinterpret_cast<const float4*>(xb + (ok ? iy * Win + ix : 0));
    }
    float hv;
    {
      const int e = threadIdx.x;
      const int r = (NHC == 2) ? e >> 1 : e, cc = (NHC == 2 && (e & 1)) ? IC + 1 : 0;
      const int iy = y0 + r, ix = x0 + cc;
      const bool ok = (e < NHC * PH) && (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
      hv = xb[ok ? iy * Win + ix : 0];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (in.mode != SC_SRC_RAW) { sc = in.cst[(size_t)c * SC_CST]; sh = in.cst[(size_t)c * SC_CST + 1]; }
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];
#pragma unroll
    for (int i = 0; i < NI4; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / IC4, c4 = e % IC4;
      const int iy = y0 + r, ix = x0 + 1 + 4 * c4;
      const bool ok = (iy >= 0) && (iy < Hin) && (ix < Win);
      float4 t;
      t.x = ok ? sc_pro_affine(v[i].x, sc, sh, lo, hi) : 0.f; t.y = ok ? sc_pro_affine(v[i].y, sc, sh, lo, hi) : 0.f;
      t.z = ok ? sc_pro_affine(v[i].z, sc, sh, lo, hi) : 0.f; t.w = ok ? sc_pro_affine(v[i].w, sc, sh, lo, hi) : 0.f;
      if (r < PH) *reinterpret_cast<float4*>(&s_x[r * PWP + 4 + 4 * c4]) = t;
    }
    {
      const int e = threadIdx.x;
      const int r = (NHC == 2) ? e >> 1 : e, cc = (NHC == 2 && (e & 1)) ? IC + 1 : 0;
      const int iy = y0 + r, ix = x0 + cc;
      const bool ok = (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
      if (e < NHC * PH) s_x[r * PWP + CO + cc] = ok ? sc_pro_affine(hv, sc, sh, lo, hi) : 0.f;
    }
  } else {
#pragma unroll
  for (int i0 = 0; i0 < NIT; i0 += LB) {
    float v[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j)
      if (i0 + j < NIT) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / PW, cc = e - r * PW;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (e < NE) && (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
        v[j] = xb[ok ? iy * Win + ix : 0];
      }
    if (i0 == 0) {
      __builtin_amdgcn_sched_barrier(0);
      if (in.mode != SC_SRC_RAW) { sc = in.cst[(size_t)c * SC_CST]; sh = in.cst[(size_t)c * SC_CST + 1]; }
#pragma unroll
      for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];
    }
#pragma unroll
    for (int j = 0; j < LB; ++j)
      if (i0 + j < NIT) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / PW, cc = e - r * PW;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
        if (e < NE) s_x[r * PWP + cc] = ok ? sc_pro_affine(v[j], sc, sh, lo, hi) : 0.f;
      }
  }
  }
  __syncthreads();
  const int ty = threadIdx.x / TW, tx = threadIdx.x % TW;
  const int ox = tx0 + tx;
  float* ob = out + ((size_t)n * C + c) * Hout * Wout;
  float sv[2] = {0.f, 0.f};
  if (S == 1) {
    // a thread owns R consecutive rows of one column: 3x3 register window, three LDS reads per output row
    const int row0 = ty * R;
    float a[3][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) a[j][q] = s_x[(row0 + j) * PWP + CO + tx + q];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int row = row0 + k, oy = ty0 + row;
#pragma unroll
      for (int q = 0; q < 3; ++q) a[2][q] = s_x[(row + 2) * PWP + CO + tx + q];
      float acc = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc = fmaf(wk[kh * 3 + kw], a[kh][kw], acc);
      if ((oy < Hout) && (ox < Wout)) {
        ob[(size_t)oy * Wout + ox] = acc;
        sv[0] += acc; sv[1] = fmaf(acc, acc, sv[1]);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) { a[0][q] = a[1][q]; a[1][q] = a[2][q]; }
    }
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int row = ty + k * TB, oy = ty0 + row;
      float acc = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc = fmaf(wk[kh * 3 + kw], s_x[(row * S + kh) * PWP + CO + tx * S + kw], acc);
      if ((oy < Hout) && (ox < Wout)) {
        ob[(size_t)oy * Wout + ox] = acc;
        sv[0] += acc; sv[1] = fmaf(acc, acc, sv[1]);
      }
    }
  }
  if (stats) {
    block_sum<2>(sv, s_tmp);
    if (tail.tickets == nullptr) {
      if (threadIdx.x < 2) stats[(((size_t)n * tiles + tile) * C + c) * 2 + threadIdx.x] = sv[threadIdx.x];
    } else {          // producer-tail finalize (sc_common.h): the channel's last (n, tile) arrival writes its BatchNorm constants
      __shared__ int s_last;
      if (threadIdx.x == 0) s_last = bn_tail_arrive(tail, stats + (((size_t)n * tiles + tile) * C + c) * 2, sv[0], sv[1], c) ? 1 : 0;
      __syncthreads();
      if (s_last && threadIdx.x < 64) bn_tail_channel(tail, stats, C, c, threadIdx.x);
    }
  }
}

// dy patch (BatchNorm/ReLU6 backward applied on load) staged once; dx tile = TW x (TB*R) input pixels
template <int S, int TW, int R>
__global__ __launch_bounds__(256) void k_dw_dgrad(const SrcD dy, const float* __restrict__ w, float* __restrict__ dx,
                                                  int accum, int C, int Hin, int Win, int Hout, int Wout) {
  constexpr int TB = 256 / TW, TH = TB * R;
  constexpr int PH = (S == 1) ? TH + 2 : TH / 2 + 1, PW = (S == 1) ? TW + 2 : TW / 2 + 1, PWP = PW | 1;
  static_assert(S == 1 || (TH % 2 == 0 && TW % 2 == 0), "stride-2 tiles must be even");
  __shared__ float s_d[PH * PWP];
  const int c = blockIdx.y, n = blockIdx.z;
  const int tiles_x = (Win + TW - 1) / TW;
  const int iy0 = (blockIdx.x / tiles_x) * TH, ix0 = (blockIdx.x % tiles_x) * TW;
  const int oyb = (S == 1) ? iy0 - 1 : iy0 / 2, oxb = (S == 1) ? ix0 - 1 : ix0 / 2;
  float4 c0 = make_float4(1.f, 0.f, 0.f, 0.f); float c4 = 0.f;
  if (dy.mode != SC_SRC_RAW) { c0 = *reinterpret_cast<const float4*>(dy.cst + (size_t)c * SC_CST); c4 = dy.cst[(size_t)c * SC_CST + 4]; }
  const float lo = sc_act_lo(dy.act), hi = sc_act_hi(dy.act);
  const size_t obase = ((size_t)n * C + c) * Hout * Wout;
  const float* gb = dy.x + obase;
  if (dy.mode == SC_SRC_BNBWD)
    dw_stage2<PH, PW, PWP>(s_d, gb, dy.aux + obase, oyb, oxb, Hout, Wout,
                           [&](float g, float yv) { return sc_pro_bnbwd(g, yv, c0.x, c0.y, c0.z, c0.w, c4, lo, hi); });
  else
    dw_stage<PH, PW, PWP>(s_d, gb, oyb, oxb, Hout, Wout, [&](float g) { return sc_pro_affine(g, c0.x, c0.y, lo, hi); });
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];
  __syncthreads();
  const int ty = threadIdx.x / TW, tx = threadIdx.x % TW;
  const int ix = ix0 + tx;
  float* db = dx + ((size_t)n * C + c) * Hin * Win;
  float prev[R];
  if (accum) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int iy = iy0 + ty + k * TB;
      prev[k] = (iy < Hin && ix < Win) ? db[(size_t)iy * Win + ix] : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int iy = iy0 + ty + k * TB;
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int t = iy + 1 - kh;                 // = oy * S
      if (S == 2 && ((t & 1) || t < 0)) continue;
      const int r = (S == 1 ? t : t / 2) - oyb;  // t >= -1 only when S == 1 (zero-padded patch row)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int u = ix + 1 - kw;
        if (S == 2 && ((u & 1) || u < 0)) continue;
        const int cc = (S == 1 ? u : u / 2) - oxb;
        acc = fmaf(wk[kh * 3 + kw], s_d[r * PWP + cc], acc);
      }
    }
    if (iy < Hin && ix < Win) db[(size_t)iy * Win + ix] = accum ? prev[k] + acc : acc;
  }
}

// dW[c][tap] += sum over this block's (n, tile) list of dy * x; 9 double atomics per block
template <int S, int TW, int R>
__global__ __launch_bounds__(256) void k_dw_wgrad(const SrcD dy, const SrcD in, double* __restrict__ dw_acc, int N, int C,
                                                  int Hin, int Win, int Hout, int Wout, int dy_bcast) {
  constexpr int TB = 256 / TW, TH = TB * R;
  constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PWP = PW | 1;
  __shared__ float s_x[PH * PWP];
  __shared__ float s_tmp[36];
  const int c = blockIdx.y;
  const int tiles_x = (Wout + TW - 1) / TW, tiles_y = (Hout + TH - 1) / TH;
  const int per_img = tiles_x * tiles_y;
  const long T = (long)N * per_img;
  float xs = 1.f, xh = 0.f;
  if (in.mode != SC_SRC_RAW) { xs = in.cst[(size_t)c * SC_CST]; xh = in.cst[(size_t)c * SC_CST + 1]; }
  const float xlo = sc_act_lo(in.act), xhi = sc_act_hi(in.act);
  float4 c0 = make_float4(1.f, 0.f, 0.f, 0.f); float c4 = 0.f;
  const int cd = dy_bcast ? 0 : c, Cd = dy_bcast ? 1 : C;      // dy_bcast: one dy plane shared by every channel (head conv)
  if (dy.mode != SC_SRC_RAW) { c0 = *reinterpret_cast<const float4*>(dy.cst + (size_t)cd * SC_CST); c4 = dy.cst[(size_t)cd * SC_CST + 4]; }
  const float dlo = sc_act_lo(dy.act), dhi = sc_act_hi(dy.act);
  const bool bnb = dy.mode == SC_SRC_BNBWD;
  const int ty = threadIdx.x / TW, tx = threadIdx.x % TW;
  float prod[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) prod[t] = 0.f;
  for (long t = blockIdx.x; t < T; t += gridDim.x) {
    const int n = (int)(t / per_img);
    const int rem = (int)(t - (long)n * per_img);
    const int ty0 = (rem / tiles_x) * TH, tx0 = (rem % tiles_x) * TW;
    const float* xb = in.x + ((size_t)n * C + c) * Hin * Win;
    // this thread's R dy values first (their latency hides behind the patch staging)
    const int ox = tx0 + tx;
    const size_t dbase = ((size_t)n * Cd + cd) * Hout * Wout;
    float g[R], yv[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int oy = ty0 + ty + k * TB;
      const bool ok = (oy < Hout) && (ox < Wout);
      const size_t o = dbase + (ok ? (size_t)oy * Wout + ox : 0);
      g[k] = dy.x[o];
      yv[k] = bnb ? dy.aux[o] : 0.f;
    }
    __syncthreads();
    dw_stage<PH, PW, PWP>(s_x, xb, ty0 * S - 1, tx0 * S - 1, Hin, Win, [&](float v) { return sc_pro_affine(v, xs, xh, xlo, xhi); });
    __syncthreads();
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int row = ty + k * TB, oy = ty0 + row;
      const bool ok = (oy < Hout) && (ox < Wout);
      float dyv = bnb ? sc_pro_bnbwd(g[k], yv[k], c0.x, c0.y, c0.z, c0.w, c4, dlo, dhi) : sc_pro_affine(g[k], c0.x, c0.y, dlo, dhi);
      dyv = ok ? dyv : 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) prod[kh * 3 + kw] = fmaf(dyv, s_x[(row * S + kh) * PWP + tx * S + kw], prod[kh * 3 + kw]);
    }
  }
  block_sum<9>(prod, s_tmp);
  if (threadIdx.x < 9) atomicAdd(&dw_acc[c * 9 + threadIdx.x], (double)prod[threadIdx.x]);
}

// Depthwise backward in ONE pass over its three big tensors: dx (backward-data), dW (backward-weight, fp64 atomics) and -- when the
// input is a BatchNorm'd tensor -- the BatchNorm-backward partial sums of that INPUT (sum g_bn, sum g_bn * xhat over the tile,
// g_bn = dx * act'(BN(y_in))), which sc_bn_bwd_reduce would otherwise re-stream dx and y_in for.  In an inverted-residual block
// the tensors on either side of the depthwise conv are the 6x-expanded ones, i.e. the HBM traffic of the block's backward: the
// separate kernels read (g_d, y_d) twice and (dx, y_in) once more than this one.
// Tile = TW x TH INPUT pixels; a thread owns R CONSECUTIVE rows of one column and slides a 3x3 register window down them (three
// LDS reads per patch per row instead of nine).  The input patch is kept RAW in LDS (its centre is the y of xhat; BatchNorm +
// ReLU6 are two instructions on the three values a row step reads), the dy patch is stored after its BatchNorm-backward prologue.
// Patches are staged row-wise: lane = column, no per-element index division.
template <int PH, int TW, int PWP, typename F>
__device__ __forceinline__ void dw_stage_rows(float* s, const float* __restrict__ xb, int y0, int x0, int H, int W, F&& pro) {
  // interior columns 1..TW of the (TW + 2)-wide patch
  constexpr int NI = (PH * TW + 255) / 256;
#pragma unroll
  for (int i0 = 0; i0 < NI; i0 += SC_DW_LB) {
    float v[SC_DW_LB];
#pragma unroll
    for (int j = 0; j < SC_DW_LB; ++j)
      if (i0 + j < NI) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / TW, cc = e % TW + 1;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (r < PH) && (iy >= 0) && (iy < H) && (ix < W);
        v[j] = xb[ok ? iy * W + ix : 0];
      }
#pragma unroll
    for (int j = 0; j < SC_DW_LB; ++j)
      if (i0 + j < NI) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / TW, cc = e % TW + 1;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (iy >= 0) && (iy < H) && (ix < W);
        if (r < PH) s[r * PWP + cc] = ok ? pro(v[j]) : 0.f;
      }
  }
  // the two halo columns
  for (int e = threadIdx.x; e < 2 * PH; e += 256) {
    const int r = e >> 1, cc = (e & 1) ? TW + 1 : 0;
    const int iy = y0 + r, ix = x0 + cc;
    const bool ok = (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
    s[r * PWP + cc] = ok ? pro(xb[iy * W + ix]) : 0.f;
  }
}
template <int PH, int TW, int PWP, typename F>
__device__ __forceinline__ void dw_stage_rows2(float* s, const float* __restrict__ gb, const float* __restrict__ yb, int y0, int x0,
                                               int H, int W, F&& pro) {
  constexpr int NI = (PH * TW + 255) / 256, LB = SC_DW_LB / 2;
#pragma unroll
  for (int i0 = 0; i0 < NI; i0 += LB) {
    float g[LB], yv[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j)
      if (i0 + j < NI) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / TW, cc = e % TW + 1;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (r < PH) && (iy >= 0) && (iy < H) && (ix < W);
        const int o = ok ? iy * W + ix : 0;
        g[j] = gb[o]; yv[j] = yb[o];
      }
#pragma unroll
    for (int j = 0; j < LB; ++j)
      if (i0 + j < NI) {
        const int e = threadIdx.x + (i0 + j) * 256;
        const int r = e / TW, cc = e % TW + 1;
        const int iy = y0 + r, ix = x0 + cc;
        const bool ok = (iy >= 0) && (iy < H) && (ix < W);
        if (r < PH) s[r * PWP + cc] = ok ? pro(g[j], yv[j]) : 0.f;
      }
  }
  for (int e = threadIdx.x; e < 2 * PH; e += 256) {
    const int r = e >> 1, cc = (e & 1) ? TW + 1 : 0;
    const int iy = y0 + r, ix = x0 + cc;
    const bool ok = (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
    s[r * PWP + cc] = ok ? pro(gb[iy * W + ix], yb[iy * W + ix]) : 0.f;
  }
}

// V4 (widths divisible by 4, 16-byte aligned tensors): the interior columns of both patches are staged as float4 -- a quarter of
// the load instructions, index arithmetic and bounds tests of the element-wise form (these kernels are issue-bound: ~1100 VALU
// per wave for 8 outputs per thread, 40 % of them staging).  The interior then starts at LDS column 4 (CO = 3 columns of offset)
// so that the stores are aligned ds_write_b128.
template <int S, int TW, int R, bool V4>
__global__ __launch_bounds__(256) void k_dw_bwd(const SrcD dy, const SrcD in, const float* __restrict__ w, float* __restrict__ dx,
                                                double* __restrict__ dw_acc, double* __restrict__ in_sums, int NC, int C, int Hin,
                                                int Win, int Hout, int Wout) {
  constexpr int TB = 256 / TW, TH = TB * R;
  constexpr int CO = V4 ? 3 : 0, DCO = (S == 1) ? CO : 0;      // LDS column offsets of the input / dy patches
  constexpr int PHD = (S == 1) ? TH + 2 : TH / 2 + 1, PWD = (S == 1) ? TW + 2 : TW / 2 + 1;
  constexpr int PWDP = V4 ? ((PWD + DCO + 3) & ~3) : (PWD | 1);
  constexpr int PHX = TH + 2, PWX = TW + 2, PWXP = V4 ? TW + 8 : (PWX | 1);
  static_assert(S == 1 || (TH % 2 == 0 && TW % 2 == 0), "stride-2 tiles must be even");
  __shared__ __attribute__((aligned(16))) float s_d[PHD * PWDP];
  __shared__ __attribute__((aligned(16))) float s_x[PHX * PWXP];      // RAW input values (zero outside the image)
  __shared__ float s_tmp[44];
  // XCD-aware numbering: work-groups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  The tiles
  // of one (n, c) plane share halo rows / columns, so they are numbered 8 apart: a plane's tiles all run on ONE XCD, close in time.
  const int tiles_x = (Win + TW - 1) / TW;
  const int tiles = tiles_x * ((Hin + TH - 1) / TH);
  const int planes = NC;                           // = N * C (the launcher pads the grid to whole groups of 8 planes)
  const int grp = blockIdx.x / (8 * tiles), within = blockIdx.x - grp * 8 * tiles;
  const int plane = grp * 8 + (within & 7), tile = within >> 3;
  if (plane >= planes) return;
  const int n = plane / C, c = plane - n * C;
  const int iy0 = (tile / tiles_x) * TH, ix0 = (tile % tiles_x) * TW;
  const int oyb = (S == 1) ? iy0 - 1 : iy0 / 2, oxb = (S == 1) ? ix0 - 1 : ix0 / 2;
  // (the per-channel constants are requested AFTER the patch loads below: scalar loads return out of order, so the wait for
  // the tensor pointers in front of the first patch load would otherwise also wait for them -- a dependent scalar-memory round
  // trip per work-group before anything is in flight)
  float4 c0 = make_float4(1.f, 0.f, 0.f, 0.f); float c4 = 0.f;
  const float dlo = sc_act_lo(dy.act), dhi = sc_act_hi(dy.act);
  float xs = 1.f, xh = 0.f, xmean = 0.f, xinv = 1.f;
  const float xlo = sc_act_lo(in.act), xhi = sc_act_hi(in.act);
  const size_t obase = ((size_t)n * C + c) * Hout * Wout, ibase = ((size_t)n * C + c) * Hin * Win;
  // ---- staging: EVERY global load of the tile (dy patch as g and y, raw input patch, halo columns) is issued before the first
  // one is consumed -- one memory round trip per work-group instead of one per batch (these kernels are latency-bound: ~30 KB per
  // work-group, a few hundred instructions per thread)
  const bool bnb = dy.mode == SC_SRC_BNBWD;
  const float* gbp = dy.x + obase;
  const float* ybp = bnb ? dy.aux + obase : dy.x + obase;
  const float* xbp = in.x + ibase;
  auto dpro = [&](float g, float yv) {
    return bnb ? sc_pro_bnbwd(g, yv, c0.x, c0.y, c0.z, c0.w, c4, dlo, dhi) : sc_pro_affine(g, c0.x, c0.y, dlo, dhi);
  };
  constexpr int DCOLS = (S == 1) ? TW : PWD;                 // S == 1: interior columns row-wise (+ 2 halo columns); S == 2: whole patch
  constexpr int DOFF = (S == 1) ? 1 : 0;
  // V4: DC4 / XC4 float4 columns per patch row (S == 2: the dy patch is TW/2 aligned columns + one more, staged like a halo column)
  constexpr int DC4 = ((S == 1) ? TW : TW / 2) / 4, XC4 = TW / 4;
  constexpr int NID = V4 ? (PHD * DC4 + 255) / 256 : (PHD * DCOLS + 255) / 256, NIX = V4 ? (PHX * XC4 + 255) / 256 : (PHX * TW + 255) / 256;
  typedef typename std::conditional<V4, float4, float>::type stage_t;
  stage_t vg[NID], vy[NID], vx[NIX];
  float hg = 0.f, hy = 0.f, hx = 0.f, hg2 = 0.f, hy2 = 0.f;
  if constexpr (V4) {
#pragma unroll
    for (int i = 0; i < NID; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / DC4, c4 = e % DC4;
      const int oy = oyb + r, ox = oxb + DOFF + 4 * c4;
      const bool ok = (r < PHD) && (oy >= 0) && (oy < Hout) && (ox < Wout);
      const int o = ok ? oy * Wout + ox : 0;
      vg[i] = *reinterpret_cast<const float4*>(gbp + o); vy[i] = *reinterpret_cast<const float4*>(ybp + o);
    }
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / XC4, c4 = e % XC4;
      const int iy = iy0 - 1 + r, ixx = ix0 + 4 * c4;
      const bool ok = (r < PHX) && (iy >= 0) && (iy < Hin) && (ixx < Win);
      vx[i] = *reinterpret_cast<const float4*>(xbp + (ok ? iy * Win + ixx : 0));
    }
  } else {
#pragma unroll
    for (int i = 0; i < NID; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / DCOLS, cc = e % DCOLS + DOFF;
      const int oy = oyb + r, ox = oxb + cc;
      const bool ok = (r < PHD) && (oy >= 0) && (oy < Hout) && (ox >= 0) && (ox < Wout);
      const int o = ok ? oy * Wout + ox : 0;
      vg[i] = gbp[o]; vy[i] = ybp[o];
    }
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / TW, cc = e % TW + 1;
      const int iy = iy0 - 1 + r, ixx = ix0 - 1 + cc;
      const bool ok = (r < PHX) && (iy >= 0) && (iy < Hin) && (ixx < Win);
      vx[i] = xbp[ok ? iy * Win + ixx : 0];
    }
  }
  static_assert(2 * PHX <= 256 && 2 * PHD <= 512, "halo columns: at most one (input) / two (dy) elements per thread");
  {
    const int e = threadIdx.x;
    const int r = e >> 1, cc = (e & 1) ? TW + 1 : 0;
    const int iy = iy0 - 1 + r, ixx = ix0 - 1 + cc;
    const bool ok = (e < 2 * PHX) && (iy >= 0) && (iy < Hin) && (ixx >= 0) && (ixx < Win);
    hx = xbp[ok ? iy * Win + ixx : 0];
    if (S == 1) {
      const int oy = oyb + r, ox = oxb + cc;
      const bool okd = (e < 2 * PHD) && (oy >= 0) && (oy < Hout) && (ox >= 0) && (ox < Wout);
      const int o = okd ? oy * Wout + ox : 0;
      hg = gbp[o]; hy = ybp[o];
    } else if (V4) {                     // S == 2: patch column TW/2 (row e)
      const int oy = oyb + e, ox = oxb + TW / 2;
      const bool okd = (e < PHD) && (oy >= 0) && (oy < Hout) && (ox < Wout);
      const int o = okd ? oy * Wout + ox : 0;
      hg = gbp[o]; hy = ybp[o];
    }
    (void)hg2; (void)hy2;
  }
  // ---- all loads are in flight: the channel's constants, then consume
  __builtin_amdgcn_sched_barrier(0);
  if (dy.mode != SC_SRC_RAW) { c0 = *reinterpret_cast<const float4*>(dy.cst + (size_t)c * SC_CST); c4 = dy.cst[(size_t)c * SC_CST + 4]; }
  if (in.mode != SC_SRC_RAW) {
    const float4 ci = *reinterpret_cast<const float4*>(in.cst + (size_t)c * SC_CST);
    xs = ci.x; xh = ci.y; xmean = ci.z; xinv = ci.w;
  }
  if constexpr (V4) {
#pragma unroll
    for (int i = 0; i < NID; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / DC4, c4 = e % DC4;
      const int oy = oyb + r, ox = oxb + DOFF + 4 * c4;
      const bool ok = (oy >= 0) && (oy < Hout) && (ox < Wout);
      float4 t;
      t.x = ok ? dpro(vg[i].x, vy[i].x) : 0.f; t.y = ok ? dpro(vg[i].y, vy[i].y) : 0.f;
      t.z = ok ? dpro(vg[i].z, vy[i].z) : 0.f; t.w = ok ? dpro(vg[i].w, vy[i].w) : 0.f;
      if (r < PHD) *reinterpret_cast<float4*>(&s_d[r * PWDP + DCO + DOFF + 4 * c4]) = t;
    }
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / XC4, c4 = e % XC4;
      const int iy = iy0 - 1 + r, ixx = ix0 + 4 * c4;
      const bool ok = (iy >= 0) && (iy < Hin) && (ixx < Win);
      if (r < PHX) *reinterpret_cast<float4*>(&s_x[r * PWXP + CO + 1 + 4 * c4]) = ok ? vx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NID; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / DCOLS, cc = e % DCOLS + DOFF;
      const int oy = oyb + r, ox = oxb + cc;
      const bool ok = (oy >= 0) && (oy < Hout) && (ox >= 0) && (ox < Wout);
      if (r < PHD) s_d[r * PWDP + cc] = ok ? dpro(vg[i], vy[i]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / TW, cc = e % TW + 1;
      const int iy = iy0 - 1 + r, ixx = ix0 - 1 + cc;
      const bool ok = (iy >= 0) && (iy < Hin) && (ixx < Win);
      if (r < PHX) s_x[r * PWXP + cc] = ok ? vx[i] : 0.f;
    }
  }
  {
    const int e = threadIdx.x;
    const int r = e >> 1, cc = (e & 1) ? TW + 1 : 0;
    const int iy = iy0 - 1 + r, ixx = ix0 - 1 + cc;
    if (e < 2 * PHX) s_x[r * PWXP + CO + cc] = ((iy >= 0) && (iy < Hin) && (ixx >= 0) && (ixx < Win)) ? hx : 0.f;
    if (S == 1 && e < 2 * PHD) {
      const int oy = oyb + r, ox = oxb + cc;
      s_d[r * PWDP + DCO + cc] = ((oy >= 0) && (oy < Hout) && (ox >= 0) && (ox < Wout)) ? dpro(hg, hy) : 0.f;
    }
    if (S == 2 && V4 && e < PHD) {
      const int oy = oyb + e, ox = oxb + TW / 2;
      s_d[e * PWDP + TW / 2] = ((oy >= 0) && (oy < Hout) && (ox < Wout)) ? dpro(hg, hy) : 0.f;
    }
  }
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];
  __syncthreads();
  const int ty = threadIdx.x / TW, tx = threadIdx.x % TW;
  const int ix = ix0 + tx;
  const int row0 = ty * R;                         // this thread's rows of the tile: row0 .. row0 + R - 1, consecutive
  // zero padding applies to the ACTIVATED input: columns / rows outside the image contribute 0 whatever BN(0) is
  const bool cok[3] = {ix - 1 >= 0 && ix - 1 < Win, ix < Win, ix + 1 < Win};
  auto xact = [&](float raw, bool ok) { return ok ? sc_pro_affine(raw, xs, xh, xlo, xhi) : 0.f; };
  float* db = dx + ibase;
  float red[2] = {0.f, 0.f};
  float prod[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) prod[t] = 0.f;
  if (S == 1) {
    float xr[3][3], xa[3][3], dd[3][3];            // raw input, activated input, dy: rows (k, k+1, k+2) of the patches, columns tx..tx+2
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool rok = (iy0 - 1 + row0 + j >= 0) && (iy0 - 1 + row0 + j < Hin);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        xr[j][q] = s_x[(row0 + j) * PWXP + CO + tx + q];
        xa[j][q] = xact(xr[j][q], rok && cok[q]);
        dd[j][q] = s_d[(row0 + j) * PWDP + DCO + tx + q];
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int row = row0 + k, iy = iy0 + row;
      const bool rok = iy + 1 < Hin;               // patch row (row + 2) is image row iy + 1 >= 0
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        xr[2][q] = s_x[(row + 2) * PWXP + CO + tx + q];
        xa[2][q] = xact(xr[2][q], rok && cok[q]);
        dd[2][q] = s_d[(row + 2) * PWDP + DCO + tx + q];
      }
      float acc = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc = fmaf(wk[kh * 3 + kw], dd[2 - kh][2 - kw], acc);
      const bool ok = iy < Hin && ix < Win;
      const float dyv = ok ? dd[1][1] : 0.f;       // S == 1: Hout == Hin, Wout == Win
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) prod[kh * 3 + kw] = fmaf(dyv, xa[kh][kw], prod[kh * 3 + kw]);
      if (ok) {
        db[(size_t)iy * Win + ix] = acc;
        const float yraw = xr[1][1];
        const float yh = fmaf(yraw, xs, xh);
        const float gb = (yh > xlo && yh < xhi) ? acc : 0.f;
        red[0] += gb;
        red[1] = fmaf(gb, (yraw - xmean) * xinv, red[1]);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        xr[0][q] = xr[1][q]; xr[1][q] = xr[2][q];
        xa[0][q] = xa[1][q]; xa[1][q] = xa[2][q];
        dd[0][q] = dd[1][q]; dd[1][q] = dd[2][q];
      }
    }
  } else {
    // stride 2: a thread owns 2 x 2 blocks of the input tile -- block (a, b) = input rows 2a, 2a+1 / columns 2b, 2b+1 = output pixel
    // (a, b).  The taps that reach each of its four pixels are fixed by the pixel's parity (1, 2, 2 and 4 of them), so nothing
    // diverges: 9 FMAs for the four dx values, 9 for the filter gradient of output (a, b), one 3 x 3 window of the input and a 2 x 2
    // window of dy from LDS.  (Before: one pixel per thread and row with `if (parity) continue` inside the tap loops -- the column
    // parity alternates from lane to lane, so every wave walked all nine taps with half its lanes masked, ~110 VALU per pixel.)
    // The FMA order per pixel is the one of k_dw_dgrad (kh outer, kw inner), so dx stays bit-identical to it.
    constexpr int BW = TW / 2, NB = (TH / 2) * BW;
    for (int blk = threadIdx.x; blk < NB; blk += 256) {
      const int a = blk / BW, b = blk - a * BW;
      const int oy = iy0 / 2 + a, ox = ix0 / 2 + b;
      const float d00 = s_d[a * PWDP + b], d01 = s_d[a * PWDP + b + 1];
      const float d10 = s_d[(a + 1) * PWDP + b], d11 = s_d[(a + 1) * PWDP + b + 1];
      float xr[3][3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xr[kh][kw] = s_x[(2 * a + kh) * PWXP + CO + 2 * b + kw];
      const float dyv = (oy < Hout && ox < Wout) ? d00 : 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int yy = iy0 - 1 + 2 * a + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int xx = ix0 - 1 + 2 * b + kw;
          const bool ok = yy >= 0 && yy < Hin && xx >= 0 && xx < Win;
          prod[kh * 3 + kw] = fmaf(dyv, xact(xr[kh][kw], ok), prod[kh * 3 + kw]);
        }
      }
      float o[2][2];
      o[0][0] = fmaf(wk[4], d00, 0.f);
      o[0][1] = fmaf(wk[5], d00, fmaf(wk[3], d01, 0.f));
      o[1][0] = fmaf(wk[7], d00, fmaf(wk[1], d10, 0.f));
      o[1][1] = fmaf(wk[8], d00, fmaf(wk[6], d01, fmaf(wk[2], d10, fmaf(wk[0], d11, 0.f))));
      const int iy = iy0 + 2 * a, ixx = ix0 + 2 * b;
      if (iy < Hin && ixx < Win) {                 // (even sizes: a block is inside the image or outside as a whole)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          *reinterpret_cast<float2*>(db + (size_t)(iy + i) * Win + ixx) = make_float2(o[i][0], o[i][1]);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float yraw = xr[i + 1][j + 1];
            const float yh = fmaf(yraw, xs, xh);
            const float gb = (yh > xlo && yh < xhi) ? o[i][j] : 0.f;
            red[0] += gb;
            red[1] = fmaf(gb, (yraw - xmean) * xinv, red[1]);
          }
        }
      }
    }
  }
  float all[11];
#pragma unroll
  for (int t = 0; t < 9; ++t) all[t] = prod[t];
  all[9] = red[0]; all[10] = red[1];
  block_sum<11>(all, s_tmp);
  if (threadIdx.x < 9) atomicAdd(&dw_acc[c * 9 + threadIdx.x], (double)all[threadIdx.x]);
  if (in_sums && threadIdx.x < 2) in_sums[(((size_t)n * tiles + tile) * C + c) * 2 + threadIdx.x] = (double)all[9 + threadIdx.x];
}

// The same fused backward for whole 16x16 / 32x32 planes at stride 1 (features.8-13 and 15-17 at 512^2 input: up to 15360 planes
// per layer).  One plane is one tile there, and a 256-thread work-group per plane spends its time on launch, barrier and an
// 11-value block reduction (16x16: 65 us per layer = 1 TB/s, slower than the three separate kernels).  Here a WAVE owns a plane:
// G float4 loads per lane and tensor (PS = 16: G = 1, four lanes per row; PS = 32: G = 4, two lanes per row), a wave-private LDS
// patch with a zero border, 4 G outputs per lane, DPP wave sums, atomics from lane 0.
template <int PS>
__global__ __launch_bounds__(256) void k_dw_bwd_plane(const SrcD dy, const SrcD in, const float* __restrict__ w, float* __restrict__ dx,
                                                      double* __restrict__ dw_acc, double* __restrict__ in_sums, int C) {
  constexpr int P = PS + 8, PR = PS + 2;         // patch pitch (interior at columns 4..PS+3, halo columns 3 and PS+4), patch rows
  constexpr int LPR = (PS == 16) ? 4 : 2, G = PS / (4 * LPR), HW = PS * PS;
  __shared__ __attribute__((aligned(16))) float s_d[4][PR * P];
  __shared__ __attribute__((aligned(16))) float s_x[4][PR * P];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int plane = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);      // (the launcher guarantees N * C % 4 == 0)
  const int n = plane / C, c = plane - n * C;
  const bool bnb = dy.mode == SC_SRC_BNBWD;
  const int r = lane / LPR, q = lane % LPR;      // this lane's image row and its run of 4 G columns
  const size_t base = (size_t)plane * HW + (size_t)r * PS + q * 4 * G;
  float4 g4[G], y4[G], x4[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    g4[g] = *reinterpret_cast<const float4*>(dy.x + base + 4 * g);
    y4[g] = bnb ? *reinterpret_cast<const float4*>(dy.aux + base + 4 * g) : g4[g];
    x4[g] = *reinterpret_cast<const float4*>(in.x + base + 4 * g);
  }
  float4 c0 = make_float4(1.f, 0.f, 0.f, 0.f); float c4 = 0.f;
  float xs = 1.f, xh = 0.f, xmean = 0.f, xinv = 1.f;
  if (dy.mode != SC_SRC_RAW) { c0 = *reinterpret_cast<const float4*>(dy.cst + (size_t)c * SC_CST); c4 = dy.cst[(size_t)c * SC_CST + 4]; }
  if (in.mode != SC_SRC_RAW) {
    const float4 ci = *reinterpret_cast<const float4*>(in.cst + (size_t)c * SC_CST);
    xs = ci.x; xh = ci.y; xmean = ci.z; xinv = ci.w;
  }
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];
  const float dlo = sc_act_lo(dy.act), dhi = sc_act_hi(dy.act), xlo = sc_act_lo(in.act), xhi = sc_act_hi(in.act);
  auto dpro = [&](float g, float yv) {
    return bnb ? sc_pro_bnbwd(g, yv, c0.x, c0.y, c0.z, c0.w, c4, dlo, dhi) : sc_pro_affine(g, c0.x, c0.y, dlo, dhi);
  };
  float* sd = s_d[wave];
  float* sx = s_x[wave];
  // zero the patches (the border is what matters), then the interior: LDS operations of one wave execute in order
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = lane; i < PR * P / 4; i += 64) { reinterpret_cast<float4*>(sd)[i] = z4; reinterpret_cast<float4*>(sx)[i] = z4; }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float4 t;
    t.x = dpro(g4[g].x, y4[g].x); t.y = dpro(g4[g].y, y4[g].y); t.z = dpro(g4[g].z, y4[g].z); t.w = dpro(g4[g].w, y4[g].w);
    *reinterpret_cast<float4*>(&sd[(r + 1) * P + 4 + 4 * (q * G + g)]) = t;
    *reinterpret_cast<float4*>(&sx[(r + 1) * P + 4 + 4 * (q * G + g)]) = x4[g];
  }
  __syncthreads();
  float prod[9], red[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t) prod[t] = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int cb = 4 * (q * G + g);              // first image column of this group of four
    // rows r-1 .. r+1 (patch rows r .. r+2), image columns cb - 1 .. cb + 4 (patch columns cb + 3 .. cb + 8)
    float D[3][6], XA[3][6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* dr = &sd[(r + j) * P + cb + 3];
      const float* xr = &sx[(r + j) * P + cb + 3];
      const float4 dm = *reinterpret_cast<const float4*>(dr + 1), xm = *reinterpret_cast<const float4*>(xr + 1);
      D[j][0] = dr[0]; D[j][1] = dm.x; D[j][2] = dm.y; D[j][3] = dm.z; D[j][4] = dm.w; D[j][5] = dr[5];
      const float xv[6] = {xr[0], xm.x, xm.y, xm.z, xm.w, xr[5]};
      const bool rok = (r - 1 + j >= 0) && (r - 1 + j < PS);
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        const int col = cb - 1 + m;
        // zero padding applies to the ACTIVATED input (see k_dw_bwd)
        XA[j][m] = (rok && col >= 0 && col < PS) ? sc_pro_affine(xv[m], xs, xh, xlo, xhi) : 0.f;
      }
    }
    float o[4];
    const float xraw[4] = {x4[g].x, x4[g].y, x4[g].z, x4[g].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float acc = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc = fmaf(wk[kh * 3 + kw], D[2 - kh][i + 2 - kw], acc);
      const float dyv = D[1][i + 1];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) prod[kh * 3 + kw] = fmaf(dyv, XA[kh][i + kw], prod[kh * 3 + kw]);
      o[i] = acc;
      const float yh = fmaf(xraw[i], xs, xh);
      const float gb = (yh > xlo && yh < xhi) ? acc : 0.f;
      red[0] += gb;
      red[1] = fmaf(gb, (xraw[i] - xmean) * xinv, red[1]);
    }
    *reinterpret_cast<float4*>(dx + base + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
  }
  float all[11];
#pragma unroll
  for (int t = 0; t < 9; ++t) all[t] = wave_sum(prod[t]);
  all[9] = wave_sum(red[0]); all[10] = wave_sum(red[1]);
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) atomicAdd(&dw_acc[c * 9 + t], (double)all[t]);
    if (in_sums) { in_sums[((size_t)n * C + c) * 2] = (double)all[9]; in_sums[((size_t)n * C + c) * 2 + 1] = (double)all[10]; }
  }
}

// forward for whole 16x16 / 32x32 planes at stride 1, a wave per plane (see k_dw_bwd_plane)
template <int PS>
__global__ __launch_bounds__(256) void k_dw_fwd_plane(const SrcD in, const float* __restrict__ w, float* __restrict__ out, int C,
                                                      float* __restrict__ stats, const BnTailD tail) {
  constexpr int P = PS + 8, PR = PS + 2;
  constexpr int LPR = (PS == 16) ? 4 : 2, G = PS / (4 * LPR), HW = PS * PS;
  __shared__ __attribute__((aligned(16))) float s_x[4][PR * P];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int plane = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  const int n = plane / C, c = plane - n * C;
  const int r = lane / LPR, q = lane % LPR;
  const size_t base = (size_t)plane * HW + (size_t)r * PS + q * 4 * G;
  float4 x4[G];
#pragma unroll
  for (int g = 0; g < G; ++g) x4[g] = *reinterpret_cast<const float4*>(in.x + base + 4 * g);
  float sc = 1.f, sh = 0.f;
  if (in.mode != SC_SRC_RAW) { sc = in.cst[(size_t)c * SC_CST]; sh = in.cst[(size_t)c * SC_CST + 1]; }
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];
  const float lo = sc_act_lo(in.act), hi = sc_act_hi(in.act);
  float* sx = s_x[wave];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = lane; i < PR * P / 4; i += 64) reinterpret_cast<float4*>(sx)[i] = z4;
#pragma unroll
  for (int g = 0; g < G; ++g)
    *reinterpret_cast<float4*>(&sx[(r + 1) * P + 4 + 4 * (q * G + g)]) =
        make_float4(sc_pro_affine(x4[g].x, sc, sh, lo, hi), sc_pro_affine(x4[g].y, sc, sh, lo, hi), sc_pro_affine(x4[g].z, sc, sh, lo, hi),
                    sc_pro_affine(x4[g].w, sc, sh, lo, hi));
  __syncthreads();
  float sv = 0.f, sq = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int cb = 4 * (q * G + g);
    float A[3][6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* xr = &sx[(r + j) * P + cb + 3];
      const float4 xm = *reinterpret_cast<const float4*>(xr + 1);
      A[j][0] = xr[0]; A[j][1] = xm.x; A[j][2] = xm.y; A[j][3] = xm.z; A[j][4] = xm.w; A[j][5] = xr[5];
    }
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float acc = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc = fmaf(wk[kh * 3 + kw], A[kh][i + kw], acc);
      o[i] = acc; sv += acc; sq = fmaf(acc, acc, sq);
    }
    *reinterpret_cast<float4*>(out + base + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (stats) {
    sv = wave_sum(sv); sq = wave_sum(sq);
    if (tail.tickets == nullptr) {
      if (lane == 0) { stats[((size_t)n * C + c) * 2] = sv; stats[((size_t)n * C + c) * 2 + 1] = sq; }
    } else {                                // producer-tail finalize: one row per image, the channel's last image arrives here
      int last = 0;
      if (lane == 0) last = bn_tail_arrive(tail, stats + ((size_t)n * C + c) * 2, sv, sq, c) ? 1 : 0;
      last = __builtin_amdgcn_readfirstlane(last);
      if (last) bn_tail_channel(tail, stats, C, c, lane);
    }
  }
}

// ---------------------------------------------------------------- stem (3x3 s2, Cin<=8 -> 32)
constexpr int STEM_CO = 32;
constexpr int STEM_MAXCI = 8;

// CINT: compile-time bound of the channel loops (4 for the 4-channel HyperSTARCOP input, STEM_MAXCI otherwise)
template <int CINT>
__global__ __launch_bounds__(256) void k_stem_fwd(const SrcD in, const float* __restrict__ w, float* __restrict__ out,
                                                  int Cin, int Hin, int Win, int Hout, int Wout, float* stats) {
  __shared__ float s_in[CINT][17 * 66];
  __shared__ float s_red[8][STEM_CO][2];        // per half-wave partial sums (DPP reductions, no LDS crossbar traffic)
  const int n = blockIdx.z;
  // tile = 32 x 8 outputs: a wave stores two 128-byte row segments per channel (16 x 16 tiles: four 64-byte pieces)
  const int tiles_x = (Wout + 31) >> 5;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy = ty * 8 + (threadIdx.x >> 5), ox = tx * 32 + (threadIdx.x & 31);
  const bool ok = (oy < Hout) && (ox < Wout);
  // The 17 x 65 input patch of the 8 x 32 output tile is staged ONCE per channel through LDS: every element is loaded and run
  // through the source prologue (the normaliser's division for raw products) once instead of up to nine times, all loads of the
  // thread are in flight together (clamped, unconditional), and the per-channel constants are uniform.  (Before: 36 loads per
  // thread, each under its own bounds branch with the constants fetched behind it -- 36 dependent memory round trips.)
  constexpr int PS = 65, PSP = 66, NE = 17 * PS, NIT = (NE + 255) / 256;
  {
    const int iy0 = ty * 16 - 1, ix0 = tx * 64 - 1;
    float raw[CINT][NIT];
#pragma unroll
    for (int ci = 0; ci < CINT; ++ci) {
      if (ci < Cin) {
        const float* xb = in.x + ((size_t)n * Cin + ci) * Hin * Win;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int e = threadIdx.x + i * 256;
          const int r = e / PS, cc = e - r * PS;
          const int iy = iy0 + r, ix = ix0 + cc;
          const bool inb = (e < NE) && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
          raw[ci][i] = xb[inb ? (size_t)iy * Win + ix : 0];
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < CINT; ++ci) {
      if (ci < Cin) {
        float4 c0 = make_float4(1.f, 0.f, 0.f, 0.f); float c4 = 0.f;
        if (in.mode != SC_SRC_RAW) { c0 = *reinterpret_cast<const float4*>(in.cst + (size_t)ci * SC_CST); c4 = in.cst[(size_t)ci * SC_CST + 4]; }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int e = threadIdx.x + i * 256;
          const int r = e / PS, cc = e - r * PS;
          const int iy = iy0 + r, ix = ix0 + cc;
          const bool inb = iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
          const float t = (in.mode == SC_SRC_RAW) ? raw[ci][i] : sc_prologue(in.mode, in.act, raw[ci][i], 0.f, c0, c4);
          if (e < NE) s_in[ci][r * PSP + cc] = inb ? t : 0.f;
        }
      }
    }
  }
  __syncthreads();
  float v[CINT * 9];
#pragma unroll
  for (int ci = 0; ci < CINT; ++ci)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        v[ci * 9 + kh * 3 + kw] = (ci < Cin) ? s_in[ci][(2 * (threadIdx.x >> 5) + kh) * PSP + 2 * (threadIdx.x & 31) + kw] : 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t HWo = (size_t)Hout * Wout;
  for (int co = 0; co < STEM_CO; ++co) {
    float acc = 0.f;
#pragma unroll
    for (int ci = 0; ci < CINT; ++ci) {
      if (ci < Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(w[(co * Cin + ci) * 9 + t], v[ci * 9 + t], acc);      // (uniform index: scalar loads, SGPR operands)
      }
    }
    if (ok) out[((size_t)n * STEM_CO + co) * HWo + (size_t)oy * Wout + ox] = acc;
    if (stats) {
      const float a = ok ? acc : 0.f;
      const float s = half_sum32(a), ss = half_sum32(a * a);
      if ((lane & 31) == SC_HALF_SUM_LANE) { s_red[wave * 2 + (lane >> 5)][co][0] = s; s_red[wave * 2 + (lane >> 5)][co][1] = ss; }
    }
  }
  if (stats) {
    __syncthreads();
    if (threadIdx.x < STEM_CO * 2) {
      const int co = threadIdx.x >> 1, k = threadIdx.x & 1;
      float t = 0.f;
#pragma unroll
      for (int h = 0; h < 8; ++h) t += s_red[h][co][k];
      stats[(stat_row() * STEM_CO + co) * 2 + k] = t;
    }
  }
}

// The 4-channel stem (the HyperSTARCOP input) on v_mfma_f32_16x16x4_f32: K step = one filter tap, lane group kq = input channel, and the
// PIXELS PERMUTED as in k_pw_stream (conv_mfma.hip) -- lane (n, kq): output row 2*wave + (n >> 3), columns 4*(n & 7) + j for MFMA j --
// so that a lane ends with four consecutive output pixels of its couts: 16-byte stores instead of one 4-byte store per (cout, pixel),
// and 72 MFMAs per 64 pixels instead of 1152 FMAs per pixel on the VALU (which alone was 31 us at 16 x 512^2 against 25 us of traffic).
// Same tile (8 x 32 outputs), grid, patch staging and statistics rows as k_stem_fwd<4>; the patch's channel pitch is 1 (mod 64) words
// so that the 64 lanes of an operand read (stride 8 words along n, two rows, four channels) fall into 64 different banks.
__global__ __launch_bounds__(256) void k_stem_fwd4m(const SrcD in, const float* __restrict__ w, float* __restrict__ out,
                                                   int Cin, int Hin, int Win, int Hout, int Wout, float* stats) {
  constexpr int PS = 65, PSP = 66, NE = 17 * PS, NIT = (NE + 255) / 256, CP = 17 * PSP + 31;       // 1153 = 1 (mod 64)
  static_assert(CP % 64 == 1, "channel pitch");
  __shared__ float s_in[4 * CP];
  __shared__ float s_red[4][STEM_CO][2];
  const int n = blockIdx.z;
  const int tiles_x = (Wout + 31) >> 5;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n16 = lane & 15, kq = lane >> 4;
  // the filter: A[cb][tap] = w[16 cb + n16][kq][tap]  (channels past Cin: zeros)
  float A[2][9];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int t = 0; t < 9; ++t) A[cb][t] = kq < Cin ? w[((cb * 16 + n16) * Cin + kq) * 9 + t] : 0.f;
  {
    const int iy0 = ty * 16 - 1, ix0 = tx * 64 - 1;
    float raw[4][NIT];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const float* xb = in.x + ((size_t)n * Cin + (ci < Cin ? ci : 0)) * Hin * Win;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int e = threadIdx.x + i * 256;
        const int r = e / PS, cc = e - r * PS;
        const int iy = iy0 + r, ix = ix0 + cc;
        const bool inb = (e < NE) && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
        raw[ci][i] = xb[inb ? (size_t)iy * Win + ix : 0];
      }
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      float4 c0 = make_float4(1.f, 0.f, 0.f, 0.f); float c4 = 0.f;
      if (in.mode != SC_SRC_RAW && ci < Cin) { c0 = *reinterpret_cast<const float4*>(in.cst + (size_t)ci * SC_CST); c4 = in.cst[(size_t)ci * SC_CST + 4]; }
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int e = threadIdx.x + i * 256;
        const int r = e / PS, cc = e - r * PS;
        const int iy = iy0 + r, ix = ix0 + cc;
        const bool inb = iy >= 0 && iy < Hin && ix >= 0 && ix < Win && ci < Cin;
        const float t = (in.mode == SC_SRC_RAW) ? raw[ci][i] : sc_prologue(in.mode, in.act, raw[ci][i], 0.f, c0, c4);
        if (e < NE) s_in[ci * CP + r * PSP + cc] = inb ? t : 0.f;
      }
    }
  }
  __syncthreads();
  const int orow = 2 * wave + (n16 >> 3), ocol = 4 * (n16 & 7);
  floatx4 acc[2][4];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[cb][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  const float* sp = s_in + kq * CP + (2 * orow) * PSP + 2 * ocol;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      float b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = sp[kh * PSP + 2 * j + kw];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[cb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cb][kh * 3 + kw], b[j], acc[cb][j], 0, 0, 0);
    }
  const int oy = ty * 8 + orow, ox = tx * 32 + ocol;
  const bool ok = (oy < Hout) && (ox < Wout);                 // (Wout % 4 == 0: the four columns are inside or outside together)
  const size_t HWo = (size_t)Hout * Wout;
  float* ob = out + (size_t)n * STEM_CO * HWo + (size_t)oy * Wout + ox;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cb * 16 + 4 * kq + r;
      const float4 o = make_float4(acc[cb][0][r], acc[cb][1][r], acc[cb][2][r], acc[cb][3][r]);
      if (ok) *reinterpret_cast<float4*>(ob + (size_t)co * HWo) = o;
      if (stats) {
        float sv = ok ? (o.x + o.y) + (o.z + o.w) : 0.f;
        float sq = ok ? fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, o.w * o.w))) : 0.f;
        sv = row_sum16(sv); sq = row_sum16(sq);
        if (n16 == 0) { s_red[wave][co][0] = sv; s_red[wave][co][1] = sq; }
      }
    }
  if (stats) {
    __syncthreads();
    if (threadIdx.x < STEM_CO * 2) {
      const int co = threadIdx.x >> 1, k = threadIdx.x & 1;
      stats[(stat_row() * STEM_CO + co) * 2 + k] = (s_red[0][co][k] + s_red[1][co][k]) + (s_red[2][co][k] + s_red[3][co][k]);
    }
  }
}

// dW[co][ci][tap] = sum dy[co][oy][ox] * in[ci][2oy+kh-1][2ox+kw-1] as a GEMM on v_mfma_f32_16x16x4_f32:
//   D[co][j] (j = ci*9+tap, padded to 16*NJB) = sum_px A[co][px] * B[px][j];  A = dy tile (BatchNorm/ReLU6 backward
//   applied on load), B gathered from the normalised input patch (lane j carries its own (ci,kh,kw) offset).
// Tile = 4 output rows x 32 px, wave w owns row w and writes its own partial row (part[block*4 + w][co][j]).
template <int NJB>
__global__ __launch_bounds__(256) void k_stem_wgrad(const SrcD dy, const SrcD in, float* __restrict__ part, int N, int Cin,
                                                    int Hin, int Win, int Hout, int Wout) {
  constexpr int TR = 4, TC = 32, PA = 130;
  constexpr int IR = 2 * TR + 1, IC = 2 * TC + 1, ICP = 67;
  __shared__ float s_dy[STEM_CO * PA];
  __shared__ float s_x[STEM_MAXCI * IR * ICP];
  __shared__ __attribute__((aligned(16))) float s_c[STEM_CO * SC_CST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int J = Cin * 9;
  for (int i = tid; i < STEM_CO * SC_CST; i += 256)
    s_c[i] = (dy.cst && dy.mode != SC_SRC_RAW) ? dy.cst[i] : ((i % SC_CST) == 0 ? 1.f : 0.f);
  // per-lane gather offsets of the B operand
  int boff[NJB]; bool bok[NJB];
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb) {
    const int j = jb * 16 + l15;
    bok[jb] = j < J;
    const int jj = bok[jb] ? j : 0;
    const int ci = jj / 9, tap = jj - ci * 9;
    boff[jb] = (ci * IR + tap / 3) * ICP + (tap % 3);
  }
  floatx4 acc[2][NJB];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) acc[cb][jb] = (floatx4){0.f, 0.f, 0.f, 0.f};
  const float dlo = sc_act_lo(dy.act), dhi = sc_act_hi(dy.act);
  const bool bnb = dy.mode == SC_SRC_BNBWD;
  const int tiles_x = (Wout + TC - 1) / TC, tiles_y = (Hout + TR - 1) / TR;
  const long T = (long)N * tiles_x * tiles_y;
  const size_t HWo = (size_t)Hout * Wout;
  for (long t = blockIdx.x; t < T; t += gridDim.x) {
    const int n = (int)(t / (tiles_x * tiles_y));
    const int rem = (int)(t - (long)n * tiles_x * tiles_y);
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int oy0 = ty * TR, ox0 = tx * TC;
    __syncthreads();
    {   // dy tile: thread = pixel (tid & 127), channels (tid >> 7) + 2 i
      const int px = tid & 127, c0i = tid >> 7;
      const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
      const bool ok = (oy < Hout) && (ox < Wout);
      const size_t base = ((size_t)n * STEM_CO + c0i) * HWo + (ok ? (size_t)oy * Wout + ox : 0);
      const float* gp = dy.x + base;
      const float* yp = bnb ? dy.aux + base : gp;
      float g[16], yv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { g[i] = gp[(size_t)(2 * i) * HWo]; yv[i] = yp[(size_t)(2 * i) * HWo]; }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ch = c0i + 2 * i;
        const float4 c0 = *reinterpret_cast<const float4*>(&s_c[ch * SC_CST]);
        const float v = bnb ? sc_pro_bnbwd(g[i], yv[i], c0.x, c0.y, c0.z, c0.w, s_c[ch * SC_CST + 4], dlo, dhi)
                            : sc_pro_affine(g[i], c0.x, c0.y, dlo, dhi);
        s_dy[ch * PA + px] = ok ? v : 0.f;
      }
    }
    for (int i = tid; i < Cin * IR * IC; i += 256) {
      const int ci = i / (IR * IC), e = i - ci * (IR * IC);
      const int r = e / IC, cc = e - r * IC;
      const int iy = 2 * oy0 - 1 + r, ix = 2 * ox0 - 1 + cc;
      float v = 0.f;
      if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) v = load_src(in, ((size_t)n * Cin + ci) * Hin * Win + (size_t)iy * Win + ix, ci);
      s_x[(ci * IR + r) * ICP + cc] = v;
    }
    __syncthreads();
    const int row = wave;
#pragma unroll 2
    for (int q = 0; q < 8; ++q) {
      const int px = 4 * q + lq;
      const float a0 = s_dy[l15 * PA + row * 32 + px];
      const float a1 = s_dy[(16 + l15) * PA + row * 32 + px];
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        const float bv = s_x[boff[jb] + (2 * row) * ICP + 2 * px];
        const float b = bok[jb] ? bv : 0.f;
        acc[0][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][jb], 0, 0, 0);
        acc[1][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][jb], 0, 0, 0);
      }
    }
  }
  // the four waves' accumulators are summed through LDS in a fixed order (((w0 + w1) + w2) + w3): ONE partial row per work-group
  // (a row per wave left 4096 rows = 19 MB for the reduction kernels: 72 us, more than half of this kernel's own time)
  __syncthreads();
  float* s_sum = s_x;                              // 32 * J <= 2304 floats
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = cb * 16 + 4 * lq + r, j = jb * 16 + l15;
            if (j < J) {
              const float prev = wv ? s_sum[co * J + j] : 0.f;
              s_sum[co * J + j] = prev + acc[cb][jb][r];
            }
          }
    }
    __syncthreads();
  }
  float* pr = part + (size_t)blockIdx.x * (STEM_CO * J);      // [co][ci][tap] (OIHW order)
  for (int i = tid; i < STEM_CO * J; i += 256) pr[i] = s_sum[i];
}

// ---------------------------------------------------------------- head (3x3, Cin<=32 -> 1, bias)
constexpr int HEAD_MAXCI = 32;
constexpr int HT_R = 8, HT_C = 32;

// stage the activated input patch [Cin][HT_R+2][PC] once (branch-free BatchNorm+ReLU prologue, clamped loads).
// CIN > 0: every global load of the patch is issued before the first LDS store (one latency, not one per element).
template <int PC, int CIN>
__device__ __forceinline__ void head_stage_patch(const SrcD& in, float* s_in, int n, int Cin, int H, int W, int y0, int x0) {
  constexpr int PR = HT_R + 2, PW = HT_C + 2;
  const float lo = sc_act_lo(in.act), hi = sc_act_hi(in.act);
  const bool raw = in.mode == SC_SRC_RAW;
  if (CIN > 0) {
    constexpr int NIT = ((CIN > 0 ? CIN : 1) * PR * PW + 255) / 256;
    float xv[NIT], sc[NIT], sh[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + 256 * it;
      const int ii = i < CIN * PR * PW ? i : 0;
      const int ci = ii / (PR * PW), e = ii - ci * (PR * PW);
      const int r = e / PW, cc = e - r * PW;
      const int y = y0 - 1 + r, x = x0 - 1 + cc;
      const bool ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
      xv[it] = in.x[((size_t)n * CIN + ci) * H * W + (ok ? (size_t)y * W + x : 0)];
      sc[it] = raw ? 1.f : in.cst[(size_t)ci * SC_CST];
      sh[it] = raw ? 0.f : in.cst[(size_t)ci * SC_CST + 1];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + 256 * it;
      if (i < CIN * PR * PW) {
        const int ci = i / (PR * PW), e = i - ci * (PR * PW);
        const int r = e / PW, cc = e - r * PW;
        const int y = y0 - 1 + r, x = x0 - 1 + cc;
        const bool ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
        s_in[(ci * PR + r) * PC + cc] = ok ? sc_pro_affine(xv[it], sc[it], sh[it], lo, hi) : 0.f;
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < Cin * PR * PW; i += 256) {
    const int ci = i / (PR * PW), e = i - ci * (PR * PW);
    const int r = e / PW, cc = e - r * PW;
    const int y = y0 - 1 + r, x = x0 - 1 + cc;
    const bool ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
    const float xv = in.x[((size_t)n * Cin + ci) * H * W + (ok ? (size_t)y * W + x : 0)];
    const float sc = raw ? 1.f : in.cst[(size_t)ci * SC_CST], sh = raw ? 0.f : in.cst[(size_t)ci * SC_CST + 1];
    s_in[(ci * PR + r) * PC + cc] = ok ? sc_pro_affine(xv, sc, sh, lo, hi) : 0.f;
  }
}

// CIN > 0: compile-time channel count -> the 9*CIN filter taps are scalar (SGPR) operands; CIN == 0: generic, taps in LDS
// Register-blocked head forward for Cin = 16: tile 16 rows x 64 cols, a thread owns 4 horizontally adjacent outputs, so
// one (ci, kh) needs 6 consecutive inputs = one 16-byte + one 8-byte LDS read for 12 FMAs (the one-output-per-thread
// form does 144 4-byte LDS reads per output and is LDS-issue bound at 4x the HBM time).  Channels in two passes of 8
// (39 KB of LDS -> 4 work-groups per CU).
__global__ __launch_bounds__(256, 4) void k_head_fwd16(const SrcD in, const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ out, int H, int W) {
  constexpr int CIN = 16, CP = 8, TR = 16, TC = 64, PR = TR + 2, PW = TC + 2, PC = 68;
  __shared__ __attribute__((aligned(16))) float s_in[CP * PR * PC];
  __shared__ __attribute__((aligned(16))) float s_w[CIN * 12];           // 9 taps per channel, padded to 12
  const int n = blockIdx.z;
  const int tiles_x = (W + TC - 1) / TC;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TR, x0 = tx * TC;
  if (threadIdx.x < CIN * 12) { const int c = threadIdx.x / 12, t = threadIdx.x - 12 * c; s_w[threadIdx.x] = t < 9 ? w[c * 9 + t] : 0.f; }
  const float lo = sc_act_lo(in.act), hi = sc_act_hi(in.act);
  const bool raw = in.mode == SC_SRC_RAW;
  // scale / shift of the 16 channels through LDS, requested before the patch: read per staged row inside the loop they were a
  // (scalar) memory round trip per iteration
  __shared__ float s_sc[CIN * 2];
  if (threadIdx.x >= 224 && threadIdx.x < 224 + CIN * 2) {
    const int i = threadIdx.x - 224, c = i >> 1, h = i & 1;
    s_sc[i] = raw ? (h ? 0.f : 1.f) : in.cst[(size_t)c * SC_CST + h];
  }
  __syncthreads();
  const int py = threadIdx.x >> 4, pxg = threadIdx.x & 15;
  float acc[4];
  const float b0 = bias ? bias[0] : 0.f;
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[o] = b0;
#pragma unroll
  for (int pass = 0; pass < CIN / CP; ++pass) {
    if (pass) __syncthreads();
    // staging: a wave takes whole patch rows (uniform channel / row -> scalar constants and row addresses), lane = column;
    // lanes 0,1 also fetch the two right-halo columns; 6 rows (12 loads) in flight
    {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      constexpr int NROW = CP * PR, RB = SC_HEAD_RB;
      const int xa = x0 - 1 + lane, xb = x0 + 63 + lane;                 // xb only for lane < 2
      const bool oka = (xa >= 0) && (xa < W), okb = (lane < 2) && (xb < W);
#pragma unroll 1
      for (int rr0 = wave; rr0 < NROW; rr0 += 4 * RB) {
        float va[RB], vb[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int rr = rr0 + 4 * u;
          const int rc = rr < NROW ? rr : 0;
          const int ci = rc / PR, r = rc - ci * PR;
          const int y = y0 - 1 + r;
          const bool oky = (y >= 0) && (y < H);
          const float* row = in.x + ((size_t)n * CIN + pass * CP + ci) * H * W + (size_t)(oky ? y : 0) * W;
          va[u] = row[oka ? xa : 0];
          vb[u] = row[okb ? xb : 0];
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int rr = rr0 + 4 * u;
          if (rr < NROW) {
            const int ci = rr / PR, r = rr - ci * PR;
            const int y = y0 - 1 + r;
            const bool oky = (y >= 0) && (y < H);
            const int cg = pass * CP + ci;
            const float sc = s_sc[cg * 2], sh = s_sc[cg * 2 + 1];
            s_in[(ci * PR + r) * PC + lane] = (oky && oka) ? sc_pro_affine(va[u], sc, sh, lo, hi) : 0.f;
            if (lane < 2) s_in[(ci * PR + r) * PC + 64 + lane] = (oky && okb) ? sc_pro_affine(vb[u], sc, sh, lo, hi) : 0.f;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int ci = 0; ci < CP; ++ci) {
      const float4* wp = reinterpret_cast<const float4*>(&s_w[(pass * CP + ci) * 12]);     // broadcast reads
      const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
      const float wk[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x};
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float* rp = &s_in[(ci * PR + py + kh) * PC + 4 * pxg];
        const float4 a = *reinterpret_cast<const float4*>(rp);
        const float2 b = *reinterpret_cast<const float2*>(rp + 4);
        const float v[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[o] = fmaf(wk[kh * 3 + kw], v[o + kw], acc[o]);
      }
    }
  }
  const int y = y0 + py, x = x0 + 4 * pxg;
  if (y >= H) return;
  float* op = out + (size_t)n * H * W + (size_t)y * W + x;
  if (x + 3 < W && ((W & 3) == 0)) {
    *reinterpret_cast<float4*>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
#pragma unroll
    for (int o = 0; o < 4; ++o) if (x + o < W) op[o] = acc[o];
  }
}

// k_head_fwd16 with 16-byte staging (W % 4 == 0, 16-byte aligned planes): the 4-byte form above waits for memory six times per
// pass (a wave takes its 36 patch rows in batches of six, one 256-byte request per row) and runs at 2 TB/s of its 268 MB at 16 x 512^2.
// Here a wave request covers FOUR patch rows (16 lanes x 16 bytes each), a wave's nine requests of a pass are all in flight together,
// and the second pass's requests are issued before the first pass's stencil.  Patch row layout: left halo at [3], columns at [4, 68)
// (16-byte aligned), right halo at [68] = slot 0 of the next row (unused there).
__global__ __launch_bounds__(256, 4) void k_head_fwd16v(const SrcD in, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ out, int H, int W) {
  constexpr int CIN = 16, CP = 8, TR = 16, TC = 64, PR = TR + 2, PC = 68, NROW = CP * PR, QW = NROW / 16, NHALO = NROW * 2;
  static_assert(NROW % 16 == 0 && NHALO <= 512, "four waves x QW requests of four rows; two halo elements per thread");
  __shared__ __attribute__((aligned(16))) float s_in[NROW * PC + 4];
  __shared__ __attribute__((aligned(16))) float s_w[CIN * 12];           // 9 taps per channel, padded to 12
  __shared__ float s_sc[CIN * 2];
  const int n = blockIdx.z;
  const int tiles_x = (W + TC - 1) / TC;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TR, x0 = tx * TC;
  const float lo = sc_act_lo(in.act), hi = sc_act_hi(in.act);
  const bool raw = in.mode == SC_SRC_RAW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane >> 4, xg = x0 + 4 * (lane & 15);
  const bool okx = xg < W;                                               // (W % 4 == 0: a 16-byte group is inside or outside as a whole)
  const float* plane0 = in.x + (size_t)n * CIN * H * W;
  float4 v[QW];
  float hv[2];
  auto issue = [&](int pass) {
#pragma unroll
    for (int u = 0; u < QW; ++u) {
      const int rr = 4 * (wave + 4 * u) + lr, ci = rr / PR, y = y0 - 1 + rr - ci * PR;
      const bool oky = (y >= 0) && (y < H);
      v[u] = *reinterpret_cast<const float4*>(plane0 + ((size_t)(pass * CP + ci) * H + (oky ? y : 0)) * W + (okx ? xg : 0));
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = threadIdx.x + 256 * h, ec = e < NHALO ? e : 0, rr = ec >> 1, ci = rr / PR, y = y0 - 1 + rr - ci * PR;
      const int x = (ec & 1) ? x0 + TC : x0 - 1;
      const bool ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
      hv[h] = plane0[((size_t)(pass * CP + ci) * H + (ok ? y : 0)) * W + (ok ? x : 0)];
    }
  };
  auto commit = [&](int pass) {
#pragma unroll
    for (int u = 0; u < QW; ++u) {
      const int rr = 4 * (wave + 4 * u) + lr, ci = rr / PR, y = y0 - 1 + rr - ci * PR;
      const bool ok = (y >= 0) && (y < H) && okx;
      const float sc = s_sc[(pass * CP + ci) * 2], sh = s_sc[(pass * CP + ci) * 2 + 1];
      float4 t;
      t.x = ok ? sc_pro_affine(v[u].x, sc, sh, lo, hi) : 0.f; t.y = ok ? sc_pro_affine(v[u].y, sc, sh, lo, hi) : 0.f;
      t.z = ok ? sc_pro_affine(v[u].z, sc, sh, lo, hi) : 0.f; t.w = ok ? sc_pro_affine(v[u].w, sc, sh, lo, hi) : 0.f;
      *reinterpret_cast<float4*>(&s_in[rr * PC + 4 + 4 * (lane & 15)]) = t;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = threadIdx.x + 256 * h;
      if (e < NHALO) {
        const int rr = e >> 1, ci = rr / PR, y = y0 - 1 + rr - ci * PR;
        const int x = (e & 1) ? x0 + TC : x0 - 1;
        const bool ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
        const float sc = s_sc[(pass * CP + ci) * 2], sh = s_sc[(pass * CP + ci) * 2 + 1];
        s_in[rr * PC + ((e & 1) ? PC : 3)] = ok ? sc_pro_affine(hv[h], sc, sh, lo, hi) : 0.f;
      }
    }
  };
  issue(0);
  if (threadIdx.x < CIN * 12) { const int c = threadIdx.x / 12, t = threadIdx.x - 12 * c; s_w[threadIdx.x] = t < 9 ? w[c * 9 + t] : 0.f; }
  if (threadIdx.x >= 224 && threadIdx.x < 224 + CIN * 2) {
    const int i = threadIdx.x - 224, c = i >> 1, h = i & 1;
    s_sc[i] = raw ? (h ? 0.f : 1.f) : in.cst[(size_t)c * SC_CST + h];
  }
  __syncthreads();
  const int py = threadIdx.x >> 4, pxg = threadIdx.x & 15;
  float acc[4];
  const float b0 = bias ? bias[0] : 0.f;
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[o] = b0;
#pragma unroll
  for (int pass = 0; pass < CIN / CP; ++pass) {
    commit(pass);
    __syncthreads();
    if (pass + 1 < CIN / CP) issue(pass + 1);
#pragma unroll 2
    for (int ci = 0; ci < CP; ++ci) {
      const float4* wp = reinterpret_cast<const float4*>(&s_w[(pass * CP + ci) * 12]);     // broadcast reads
      const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
      const float wk[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x};
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float* rp = &s_in[(ci * PR + py + kh) * PC + 4 * pxg + 3];
        const float4 a = *reinterpret_cast<const float4*>(rp + 1);
        const float vv[6] = {rp[0], a.x, a.y, a.z, a.w, rp[5]};
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[o] = fmaf(wk[kh * 3 + kw], vv[o + kw], acc[o]);
      }
    }
    if (pass + 1 < CIN / CP) __syncthreads();
  }
  const int y = y0 + py, x = x0 + 4 * pxg;
  if (y >= H || x >= W) return;
  *reinterpret_cast<float4*>(out + (size_t)n * H * W + (size_t)y * W + x) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

template <int CIN>
__global__ __launch_bounds__(256) void k_head_fwd(const SrcD in, const float* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ out, int Cin_rt, int H, int W) {
  constexpr int PR = HT_R + 2, PC = HT_C + 2;
  __shared__ float s_in[(CIN > 0 ? CIN : HEAD_MAXCI) * PR * PC];
  __shared__ float s_w[HEAD_MAXCI * 9];
  const int Cin = CIN > 0 ? CIN : Cin_rt;
  const int n = blockIdx.z;
  const int tiles_x = (W + HT_C - 1) / HT_C;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * HT_R, x0 = tx * HT_C;
  if (CIN == 0) for (int i = threadIdx.x; i < Cin * 9; i += 256) s_w[i] = w[i];
  head_stage_patch<PC, CIN>(in, s_in, n, Cin, H, W, y0, x0);
  __syncthreads();
  const int py = threadIdx.x >> 5, px = threadIdx.x & 31;
  const int y = y0 + py, x = x0 + px;
  if (y >= H || x >= W) return;
  float acc = bias ? bias[0] : 0.f;
  if (CIN > 0) {
#pragma unroll
    for (int ci = 0; ci < (CIN > 0 ? CIN : 1); ++ci)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = fmaf(w[ci * 9 + t], s_in[(ci * PR + py + t / 3) * PC + px + (t % 3)], acc);
  } else {
    for (int ci = 0; ci < Cin; ++ci)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = fmaf(s_w[ci * 9 + t], s_in[(ci * PR + py + t / 3) * PC + px + (t % 3)], acc);
  }
  out[(size_t)n * H * W + (size_t)y * W + x] = acc;
}

__global__ __launch_bounds__(256) void k_head_dgrad(const float* __restrict__ dl, const float* __restrict__ w,
                                                    float* __restrict__ gin, int Cin, int H, int W) {
  constexpr int PR = HT_R + 2, PC = HT_C + 2;
  __shared__ float s_dl[PR * PC];
  __shared__ float s_w[HEAD_MAXCI * 9];
  const int n = blockIdx.z;
  const int tiles_x = (W + HT_C - 1) / HT_C;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * HT_R, x0 = tx * HT_C;
  for (int i = threadIdx.x; i < Cin * 9; i += 256) s_w[i] = w[i];
  for (int i = threadIdx.x; i < PR * PC; i += 256) {
    const int r = i / PC, cc = i - r * PC;
    const int y = y0 - 1 + r, x = x0 - 1 + cc;
    s_dl[i] = (y >= 0 && y < H && x >= 0 && x < W) ? dl[(size_t)n * H * W + (size_t)y * W + x] : 0.f;
  }
  __syncthreads();
  const int py = threadIdx.x >> 5, px = threadIdx.x & 31;
  const int y = y0 + py, x = x0 + px;
  if (y >= H || x >= W) return;
  // gin[ci][y][x] = sum_tap w[ci][kh][kw] * dl[y + 1 - kh][x + 1 - kw]
  float d[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) d[t] = s_dl[(py + 2 - t / 3) * PC + px + 2 - (t % 3)];
  for (int ci = 0; ci < Cin; ++ci) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc = fmaf(s_w[ci * 9 + t], d[t], acc);
    gin[((size_t)n * Cin + ci) * H * W + (size_t)y * W + x] = acc;
  }
}

// Head backward in one pass (Cin = 16): gin, dW and dbias from ONE sweep over (dlogits, x).
//   gin[ci](q)   = sum_t w[ci][t] * dl(q + 1 - t)                         (flipped 3x3 window of dlogits around pixel q)
//   dW[ci][t]    = sum_p dl(p) * xa[ci](p + t - 1) = sum_q xa[ci](q) * dl(q + 1 - t)   -- the SAME window, so the input needs no
//   halo: a thread reads xa[ci](q) of its own pixel only, and the separate kernels' second pass over x (268 MB at 16 x 512^2)
//   and their 16 re-reads of the dlogits plane go away.
// Work-group = 4 waves, each owning 4 input channels; tile = 64 columns x HB_R rows, lanes = columns, rows walked with the next
// row's loads in flight; persistent over tiles so that the 36 wave sums + atomics per wave are paid once per work-group.
constexpr int HB_ROW = 16 * 9 + 1, HB_MAXWG = 1024;       // floats per partial row, most work-groups of a launch
constexpr int HB_R = 32, HB_PF = 4;       // tile rows, rows of load look-ahead (HB_R % HB_PF == 0)
// BNS: the launch also leaves the BatchNorm-backward sums of the INPUT tensor (the gradient it writes is that tensor's complete
// gradient, and its raw values pass through this kernel anyway): per work-group rows bn_sums[wg][16][2] = {sum g', sum g' x_hat},
// g' = gin * act'(BN(y)), and the range hint max |scale g'| -- what sc_bn_bwd_reduce(gin, y) computes in a pass of its own
// (536 MB at 16 x 512^2).
template <bool BNS>
__global__ __launch_bounds__(256) void k_head_bwd16(const float* __restrict__ dl, const SrcD in, const float* __restrict__ w,
                                                    float* __restrict__ gin, float* __restrict__ rows, int N, int H, int W,
                                                    double* __restrict__ bn_sums, float* __restrict__ bn_absmax) {
  constexpr int CIN = 16, PR = HB_R + 2, PC = 64 + 2;
  __shared__ float s_dl[PR * PC];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int c0 = wave * 4;                         // (uniform: the filters and constants below are scalar loads into SGPRs)
  float wk[4][9], sc[4], sh[4], mu[4], is[4];
  float b1[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f}, bmx = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc[j] = 1.f; sh[j] = 0.f; mu[j] = 0.f; is[j] = 1.f;
    if (in.mode != SC_SRC_RAW) { sc[j] = in.cst[(size_t)(c0 + j) * SC_CST]; sh[j] = in.cst[(size_t)(c0 + j) * SC_CST + 1]; }
    if (BNS) { mu[j] = in.cst[(size_t)(c0 + j) * SC_CST + 2]; is[j] = in.cst[(size_t)(c0 + j) * SC_CST + 3]; }
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[j][t] = w[(c0 + j) * 9 + t];
  }
  const float lo = sc_act_lo(in.act), hi = sc_act_hi(in.act);
  float prod[4][9], bsum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) prod[j][t] = 0.f;
  const int tiles_x = (W + 63) / 64, tiles_y = (H + HB_R - 1) / HB_R;
  const int T = N * tiles_x * tiles_y;
  const size_t HW = (size_t)H * W;
  for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const int n = tile / (tiles_x * tiles_y), tt = tile - n * tiles_x * tiles_y;
    const int y0 = (tt / tiles_x) * HB_R, x0 = (tt % tiles_x) * 64;
    __syncthreads();                               // (the previous tile's window reads are done)
    for (int i = threadIdx.x; i < PR * PC; i += 256) {
      const int r = i / PC, cc = i - r * PC;
      const int y = y0 - 1 + r, x = x0 - 1 + cc;
      s_dl[i] = (y >= 0 && y < H && x >= 0 && x < W) ? dl[(size_t)n * HW + (size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    const int x = x0 + lane;
    const bool xok = x < W;
    const float* xb = in.x + ((size_t)n * CIN + c0) * HW + (xok ? x : 0);
    float* gb = gin + ((size_t)n * CIN + c0) * HW + x;
    // rows are requested HB_PF ahead (clamped, unconditional): with one row of look-ahead a wave had 1 KB in flight and the CU 8 KB
    float xq[HB_PF][4];
#pragma unroll
    for (int u = 0; u < HB_PF; ++u) {
      const int yc = (y0 + u < H) ? y0 + u : H - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) xq[u][j] = xb[(size_t)j * HW + (size_t)yc * W];
    }
    for (int r0 = 0; r0 < HB_R; r0 += HB_PF) {
#pragma unroll
      for (int u = 0; u < HB_PF; ++u) {
        const int r = r0 + u, y = y0 + r;
        // d[t] = dl(q + 1 - t): patch row (r + 1) + 1 - kh, patch column (lane + 1) + 1 - kw
        float d[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) d[kh * 3 + kw] = s_dl[(r + 2 - kh) * PC + lane + 2 - kw];
        const bool ok = xok && y < H;
        if (wave == 0) bsum += ok ? d[4] : 0.f;
        const int yn = (y + HB_PF < H) ? y + HB_PF : H - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float g = 0.f;
#pragma unroll
          for (int t = 0; t < 9; ++t) g = fmaf(wk[j][t], d[t], g);
          if (ok) gb[(size_t)j * HW + (size_t)y * W] = g;
          if (BNS) {
            const float yh = fmaf(xq[u][j], sc[j], sh[j]);
            const float gq = (ok && yh > lo && yh < hi) ? g : 0.f;
            b1[j] += gq;
            b2[j] = fmaf(gq, (xq[u][j] - mu[j]) * is[j], b2[j]);
            bmx = fmaxf(bmx, fabsf(gq * sc[j]));
          }
          const float xa = ok ? sc_pro_affine(xq[u][j], sc[j], sh[j], lo, hi) : 0.f;
#pragma unroll
          for (int t = 0; t < 9; ++t) prod[j][t] = fmaf(xa, d[t], prod[j][t]);
          xq[u][j] = xb[(size_t)j * HW + (size_t)yn * W];      // refill the slot with row r + HB_PF
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // one partial row per work-group, summed by k_head_bwd_reduce: a thousand work-groups adding into the same 145 doubles
      // serialise on the atomics (measured: 231 us with 512 work-groups, 262 us with 1024)
      const float v = wave_sum(prod[j][t]);
      if (lane == 0) rows[(size_t)blockIdx.x * HB_ROW + (c0 + j) * 9 + t] = v;
    }
  if (wave == 0) {
    bsum = wave_sum(bsum);
    if (lane == 0) rows[(size_t)blockIdx.x * HB_ROW + CIN * 9] = bsum;
  }
  if (BNS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s1 = wave_sum(b1[j]), s2 = wave_sum(b2[j]);
      if (lane == 0) {
        bn_sums[((size_t)blockIdx.x * CIN + c0 + j) * 2] = (double)s1;
        bn_sums[((size_t)blockIdx.x * CIN + c0 + j) * 2 + 1] = (double)s2;
      }
    }
    if (bn_absmax) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) bmx = fmaxf(bmx, __shfl_xor(bmx, o, 64));
      if (lane == 0 && bmx > __builtin_nontemporal_load(bn_absmax)) atomicMax(reinterpret_cast<unsigned*>(bn_absmax), __builtin_bit_cast(unsigned, bmx));
    }
  }
}
// dW / dbias = column sums (in double) of the partial rows: one work-group per column
__global__ __launch_bounds__(256) void k_head_bwd_reduce(const float* __restrict__ rows, int nrows, float* __restrict__ dw, float* __restrict__ dbias) {
  __shared__ double s_tmp[4];
  const int col = blockIdx.x;
  double v = 0.0;
  for (int r = threadIdx.x; r < nrows; r += 256) v += (double)rows[(size_t)r * HB_ROW + col];
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = (s_tmp[0] + s_tmp[1]) + (s_tmp[2] + s_tmp[3]);
    if (col < 16 * 9) dw[col] = (float)v;
    else if (dbias) dbias[0] = (float)v;
  }
}

// sum of a float array into a double accumulator (bias gradient of the head = sum of dlogits)
__global__ __launch_bounds__(256) void k_sum_f32_to_f64(const float* __restrict__ x, size_t n, double* acc) {
  __shared__ float s_tmp[4];
  float v[1] = {0.f};
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v[0] += x[i];
  block_sum<1>(v, s_tmp);
  if (threadIdx.x == 0) atomicAdd(acc, (double)v[0]);
}

__global__ void k_cast_f64_f32(const double* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

int stem_blocks(int N, int Hout, int Wout) {
  long T = (long)N * ((Wout + 31) / 32) * ((Hout + 3) / 4);
  return (int)(T < 1024 ? T : 1024);
}
int head_blocks(int N, int H, int W) {
  long T = (long)N * ((W + HT_C - 1) / HT_C) * ((H + HT_R - 1) / HT_R);
  return (int)(T < 1024 ? T : 1024);
}

}  // namespace

#define SC_DW_DISPATCH4(KERNEL, V4_, W_, GRID, ...)                                                                     \
  do {                                                                                                                  \
    const int tw_ = sc_dw_tile_w(W_);                                                                                   \
    if (stride == 1 && tw_ == 64) hipLaunchKernelGGL((KERNEL<1, 64, 8, V4_>), GRID, dim3(256), 0, st, __VA_ARGS__);     \
    else if (stride == 1 && tw_ == 32) hipLaunchKernelGGL((KERNEL<1, 32, 4, V4_>), GRID, dim3(256), 0, st, __VA_ARGS__);\
    else if (stride == 1) hipLaunchKernelGGL((KERNEL<1, 16, 1, V4_>), GRID, dim3(256), 0, st, __VA_ARGS__);             \
    else if (tw_ == 64) hipLaunchKernelGGL((KERNEL<2, 64, 8, V4_>), GRID, dim3(256), 0, st, __VA_ARGS__);               \
    else if (tw_ == 32) hipLaunchKernelGGL((KERNEL<2, 32, 4, V4_>), GRID, dim3(256), 0, st, __VA_ARGS__);               \
    else hipLaunchKernelGGL((KERNEL<2, 16, 1, V4_>), GRID, dim3(256), 0, st, __VA_ARGS__);                              \
  } while (0)

// tile shape by plane width: (TW, R) = (64, 8) -> 64x32, (32, 4) -> 32x32, (16, 1) -> 16x16 outputs per block
#define SC_DW_DISPATCH(KERNEL, W_, GRID, ...)                                                                      \
  do {                                                                                                             \
    const int tw_ = sc_dw_tile_w(W_);                                                                              \
    if (stride == 1 && tw_ == 64) hipLaunchKernelGGL((KERNEL<1, 64, 8>), GRID, dim3(256), 0, st, __VA_ARGS__);     \
    else if (stride == 1 && tw_ == 32) hipLaunchKernelGGL((KERNEL<1, 32, 4>), GRID, dim3(256), 0, st, __VA_ARGS__);\
    else if (stride == 1) hipLaunchKernelGGL((KERNEL<1, 16, 1>), GRID, dim3(256), 0, st, __VA_ARGS__);             \
    else if (tw_ == 64) hipLaunchKernelGGL((KERNEL<2, 64, 8>), GRID, dim3(256), 0, st, __VA_ARGS__);               \
    else if (tw_ == 32) hipLaunchKernelGGL((KERNEL<2, 32, 4>), GRID, dim3(256), 0, st, __VA_ARGS__);               \
    else hipLaunchKernelGGL((KERNEL<2, 16, 1>), GRID, dim3(256), 0, st, __VA_ARGS__);                              \
  } while (0)

static inline long dw_tiles(int H, int W) {
  const int tw = sc_dw_tile_w(W), th = sc_dw_tile_h(W);
  return (long)((W + tw - 1) / tw) * ((H + th - 1) / th);
}

static int dwconv3x3_fwd(const sc_src* in, const float* w, float* out, int N, int C, int Hin, int Win, int stride, float* stats,
                         const sc_bn_tail* bt, sc_stream stream);
extern "C" int sc_dwconv3x3_fwd(const sc_src* in, const float* w, float* out, int N, int C, int Hin, int Win,
                                int stride, float* stats, sc_stream stream) {
  return dwconv3x3_fwd(in, w, out, N, C, Hin, Win, stride, stats, nullptr, stream);
}
extern "C" int sc_dwconv3x3_fwd_bn(const sc_src* in, const float* w, float* out, int N, int C, int Hin, int Win,
                                   int stride, float* stats, const sc_bn_tail* bn, sc_stream stream) {
  SC_REQUIRE(bn && stats && bn->gamma && bn->beta && bn->running_mean && bn->running_var && bn->cst && bn->tickets,
             "sc_dwconv3x3_fwd_bn: statistics rows, BatchNorm parameters, constants and tickets are required");
  return dwconv3x3_fwd(in, w, out, N, C, Hin, Win, stride, stats, bn, stream);
}
static int dwconv3x3_fwd(const sc_src* in, const float* w, float* out, int N, int C, int Hin, int Win, int stride, float* stats,
                         const sc_bn_tail* bt, sc_stream stream) {
  SC_REQUIRE(in && in->C == C, "sc_dwconv3x3_fwd: source channels != C");
  SC_REQUIRE(stride == 1 || stride == 2, "sc_dwconv3x3_fwd: stride must be 1 or 2");
  SC_REQUIRE((in->mode == SC_SRC_RAW || in->mode == SC_SRC_AFFINE) && in->up == 0, "sc_dwconv3x3_fwd: unsupported source mode");
  const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  const long planes8 = ((long)N * C + 7) / 8 * 8;
  SC_REQUIRE(planes8 * dw_tiles(Hout, Wout) < (1L << 31), "sc_dwconv3x3_fwd: grid too large");
  dim3 grid((unsigned)(planes8 * dw_tiles(Hout, Wout)));
  BnTailD tail{};
  if (bt) {
    tail.gamma = bt->gamma; tail.beta = bt->beta; tail.running_mean = bt->running_mean; tail.running_var = bt->running_var;
    tail.momentum = bt->momentum; tail.eps = bt->eps; tail.cst = bt->cst; tail.act_bound = bt->act_bound;
    tail.tickets = reinterpret_cast<unsigned*>(bt->tickets);
    tail.arrivals = (int)(N * dw_tiles(Hout, Wout));          // statistics rows per channel (sc_stat_rows(SC_STAT_DW, ..))
    tail.count = (double)N * Hout * Wout;
  }
  constexpr bool p16_env = true;
  // (32 x 32 planes through the same kernel measured SLOWER than the work-group-per-plane form -- backward 50 -> 63 us, forward
  // 22 -> 27 us on features.12: 16 pixels per lane, 43 KB of LDS per work-group -- so it is opt-in: STARCOP_DW_P32=1)
  static const bool p32_env = [] { const char* e = getenv("STARCOP_DW_P32"); return e && atoi(e) != 0; }();
  const bool plane_ok = stride == 1 && Hin == Win && ((long)N * C) % 4 == 0 && (((uintptr_t)in->x | (uintptr_t)out) & 15) == 0;
  if (plane_ok && ((p16_env && Hin == 16) || (p16_env && p32_env && Hin == 32))) {      // a wave per plane
    if (Hin == 16) hipLaunchKernelGGL(k_dw_fwd_plane<16>, dim3((unsigned)((long)N * C / 4)), dim3(256), 0, st, to_srcd(*in), w, out, C, stats, tail);
    else hipLaunchKernelGGL(k_dw_fwd_plane<32>, dim3((unsigned)((long)N * C / 4)), dim3(256), 0, st, to_srcd(*in), w, out, C, stats, tail);
    SC_LAUNCH_OK("sc_dwconv3x3_fwd");
    return SC_OK;
  }
  constexpr bool v4_env = true;
  if (v4_env && Win % 4 == 0 && (((uintptr_t)in->x) & 15) == 0)
    SC_DW_DISPATCH4(k_dw_fwd, true, Wout, grid, to_srcd(*in), w, out, N * C, C, Hin, Win, Hout, Wout, stats, tail);
  else
    SC_DW_DISPATCH4(k_dw_fwd, false, Wout, grid, to_srcd(*in), w, out, N * C, C, Hin, Win, Hout, Wout, stats, tail);
  SC_LAUNCH_OK("sc_dwconv3x3_fwd");
  return SC_OK;
}

extern "C" int sc_dwconv3x3_dgrad(const sc_src* dy, const float* w, float* dx, int accum, int N, int C, int Hin,
                                  int Win, int stride, sc_stream stream) {
  SC_REQUIRE(dy && dy->C == C, "sc_dwconv3x3_dgrad: source channels != C");
  SC_REQUIRE(stride == 1 || stride == 2, "sc_dwconv3x3_dgrad: stride must be 1 or 2");
  SC_REQUIRE(dy->mode != SC_SRC_NORM && dy->up == 0, "sc_dwconv3x3_dgrad: unsupported source mode");
  SC_REQUIRE(dy->mode != SC_SRC_BNBWD || dy->aux != nullptr, "sc_dwconv3x3_dgrad: BNBWD source needs aux");
  const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)dw_tiles(Hin, Win), C, N);
  SC_DW_DISPATCH(k_dw_dgrad, Win, grid, to_srcd(*dy), w, dx, accum, C, Hin, Win, Hout, Wout);
  SC_LAUNCH_OK("sc_dwconv3x3_dgrad");
  return SC_OK;
}

extern "C" int sc_dwconv3x3_wgrad(const sc_src* dy, const sc_src* in, double* dw_acc, int N, int C, int Hin, int Win,
                                  int stride, sc_stream stream) {
  SC_REQUIRE(dy && in && dy->C == C && in->C == C, "sc_dwconv3x3_wgrad: source channels != C");
  SC_REQUIRE(stride == 1 || stride == 2, "sc_dwconv3x3_wgrad: stride must be 1 or 2");
  SC_REQUIRE((in->mode == SC_SRC_RAW || in->mode == SC_SRC_AFFINE) && in->up == 0 && dy->mode != SC_SRC_NORM && dy->up == 0,
             "sc_dwconv3x3_wgrad: unsupported source mode");
  const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  const long T = (long)N * dw_tiles(Hout, Wout);
  const long want = 4096 / C > 0 ? 4096 / C : 1;
  dim3 grid((unsigned)(T < want ? T : want), C);
  SC_DW_DISPATCH(k_dw_wgrad, Wout, grid, to_srcd(*dy), to_srcd(*in), dw_acc, N, C, Hin, Win, Hout, Wout, 0);
  SC_LAUNCH_OK("sc_dwconv3x3_wgrad");
  return SC_OK;
}

extern "C" int sc_dwconv3x3_bwd_fused(const sc_src* dy, const sc_src* in, const float* w, float* dx, double* dw_acc,
                                      double* in_sums, int N, int C, int Hin, int Win, int stride, sc_stream stream) {
  SC_REQUIRE(dy && in && w && dx && dw_acc && dy->C == C && in->C == C, "sc_dwconv3x3_bwd_fused: bad argument");
  SC_REQUIRE(stride == 1 || stride == 2, "sc_dwconv3x3_bwd_fused: stride must be 1 or 2");
  SC_REQUIRE((in->mode == SC_SRC_RAW || in->mode == SC_SRC_AFFINE) && in->up == 0 && dy->mode != SC_SRC_NORM && dy->up == 0,
             "sc_dwconv3x3_bwd_fused: unsupported source mode");
  SC_REQUIRE(dy->mode != SC_SRC_BNBWD || dy->aux != nullptr, "sc_dwconv3x3_bwd_fused: BNBWD source needs aux");
  SC_REQUIRE(!in_sums || in->mode == SC_SRC_AFFINE, "sc_dwconv3x3_bwd_fused: BatchNorm-backward sums need an affine (BatchNorm'd) input");
  SC_REQUIRE(stride == 1 || (Hin % 2 == 0 && Win % 2 == 0), "sc_dwconv3x3_bwd_fused: stride 2 needs even input sizes");
  const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  const long planes8 = ((long)N * C + 7) / 8 * 8;
  SC_REQUIRE(planes8 * dw_tiles(Hin, Win) < (1L << 31), "sc_dwconv3x3_bwd_fused: grid too large");
  dim3 grid((unsigned)(planes8 * dw_tiles(Hin, Win)));
  // float4 staging where every patch row is 16-byte aligned (SC_DW_V4=0 forces the element-wise form)
  constexpr bool v4_env = true;
  const bool v4 = v4_env && Win % 4 == 0 && Wout % 4 == 0 &&
                  ((((uintptr_t)dy->x) | ((uintptr_t)dy->aux) | ((uintptr_t)in->x)) & 15) == 0;
  constexpr bool p16_env = true;
  // (32 x 32 planes through the same kernel measured SLOWER than the work-group-per-plane form -- backward 50 -> 63 us, forward
  // 22 -> 27 us on features.12: 16 pixels per lane, 43 KB of LDS per work-group -- so it is opt-in: STARCOP_DW_P32=1)
  static const bool p32_env = [] { const char* e = getenv("STARCOP_DW_P32"); return e && atoi(e) != 0; }();
  const bool plane_ok = v4 && stride == 1 && Hin == Win && ((long)N * C) % 4 == 0 && ((uintptr_t)dx & 15) == 0;
  if (plane_ok && ((p16_env && Hin == 16) || (p16_env && p32_env && Hin == 32))) {      // a wave per plane
    if (Hin == 16) hipLaunchKernelGGL(k_dw_bwd_plane<16>, dim3((unsigned)((long)N * C / 4)), dim3(256), 0, st, to_srcd(*dy), to_srcd(*in), w, dx, dw_acc, in_sums, C);
    else hipLaunchKernelGGL(k_dw_bwd_plane<32>, dim3((unsigned)((long)N * C / 4)), dim3(256), 0, st, to_srcd(*dy), to_srcd(*in), w, dx, dw_acc, in_sums, C);
    SC_LAUNCH_OK("sc_dwconv3x3_bwd_fused");
    return SC_OK;
  }
  if (v4) SC_DW_DISPATCH4(k_dw_bwd, true, Win, grid, to_srcd(*dy), to_srcd(*in), w, dx, dw_acc, in_sums, N * C, C, Hin, Win, Hout, Wout);
  else SC_DW_DISPATCH4(k_dw_bwd, false, Win, grid, to_srcd(*dy), to_srcd(*in), w, dx, dw_acc, in_sums, N * C, C, Hin, Win, Hout, Wout);
  SC_LAUNCH_OK("sc_dwconv3x3_bwd_fused");
  return SC_OK;
}

extern "C" int sc_head_bwd_bn_rows(int N, int H, int W) {
  const long T = (long)N * ((W + 63) / 64) * ((H + HB_R - 1) / HB_R);
  return (int)(T < HB_MAXWG ? T : HB_MAXWG);
}

extern "C" int sc_head_conv_bwd(const float* dlogits, const sc_src* in, const float* w, float* gin, float* part, size_t part_floats,
                                float* dw, float* dbias, int N, int Cin, int H, int W, double* bn_sums, float* bn_absmax,
                                sc_stream stream) {
  SC_REQUIRE(dlogits && in && w && gin && part && dw && in->C == Cin, "sc_head_conv_bwd: bad argument");
  SC_REQUIRE(Cin == 16, "sc_head_conv_bwd: Cin must be 16 (use sc_head_conv_dgrad + sc_head_conv_wgrad otherwise)");
  SC_REQUIRE((in->mode == SC_SRC_RAW || in->mode == SC_SRC_AFFINE) && in->up == 0, "sc_head_conv_bwd: unsupported source mode");
  SC_REQUIRE(part_floats >= sc_head_wgrad_workspace_floats(N, Cin, H, W), "sc_head_conv_bwd: workspace too small");
  SC_REQUIRE(N > 0 && H > 0 && W > 0, "sc_head_conv_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const long T = (long)N * ((W + 63) / 64) * ((H + HB_R - 1) / HB_R);
  const int nwg = (int)(T < HB_MAXWG ? T : HB_MAXWG);       // (all resident: 116 registers -> 4 work-groups per CU)
  SC_REQUIRE(!bn_sums || in->mode == SC_SRC_AFFINE, "sc_head_conv_bwd: the fused BatchNorm-backward sums need the AFFINE source of that BatchNorm");
  if (bn_sums) hipLaunchKernelGGL(k_head_bwd16<true>, dim3(nwg), dim3(256), 0, st, dlogits, to_srcd(*in), w, gin, part, N, H, W, bn_sums, bn_absmax);
  else hipLaunchKernelGGL(k_head_bwd16<false>, dim3(nwg), dim3(256), 0, st, dlogits, to_srcd(*in), w, gin, part, N, H, W, (double*)nullptr, (float*)nullptr);
  SC_LAUNCH_OK("sc_head_conv_bwd");
  hipLaunchKernelGGL(k_head_bwd_reduce, dim3(HB_ROW), dim3(256), 0, st, part, nwg, dw, dbias);
  SC_LAUNCH_OK("sc_head_conv_bwd(reduce)");
  return SC_OK;
}

// every depthwise filter gradient of a backward walk in ONE launch (device descriptor table, built once per plan): the 17 per-layer
// casts each needed their own fork of the weight-gradient stream -- a marker in the main queue that the next kernel waits behind
__global__ __launch_bounds__(256) void k_cast_f64_f32_batch(const sc_cast_desc* __restrict__ descs) {
  const sc_cast_desc d = descs[blockIdx.x];
  for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < d.n; i += (size_t)gridDim.y * 256) d.out[i] = (float)d.in[i];
}
extern "C" int sc_cast_f64_f32_batch(const sc_cast_desc* descs_dev, int n_descs, sc_stream stream) {
  SC_REQUIRE(descs_dev != nullptr && n_descs > 0, "sc_cast_f64_f32_batch: no descriptors");
  hipLaunchKernelGGL(k_cast_f64_f32_batch, dim3(n_descs, 8), dim3(256), 0, (hipStream_t)stream, descs_dev);
  SC_LAUNCH_OK("sc_cast_f64_f32_batch");
  return SC_OK;
}

extern "C" int sc_cast_f64_f32(const double* in, float* out, size_t n, sc_stream stream) {
  if (n == 0) return SC_OK;
  const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(k_cast_f64_f32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, n);
  SC_LAUNCH_OK("sc_cast_f64_f32");
  return SC_OK;
}

extern "C" int sc_stem_conv_fwd(const sc_src* in, const float* w, float* out, int N, int Cin, int Hin, int Win,
                                float* stats, sc_stream stream) {
  SC_REQUIRE(in && in->C == Cin, "sc_stem_conv_fwd: source channels != Cin");
  SC_REQUIRE(Cin >= 1 && Cin <= STEM_MAXCI, "sc_stem_conv_fwd: Cin must be in [1,%d] (got %d)", STEM_MAXCI, Cin);
  SC_REQUIRE(in->mode != SC_SRC_BNBWD && in->up == 0, "sc_stem_conv_fwd: unsupported source mode");
  const int Hout = (Hin - 1) / 2 + 1, Wout = (Win - 1) / 2 + 1;
  dim3 grid(((Wout + 31) / 32) * ((Hout + 7) / 8), 1, N);
  static const bool mfma_off = [] { const char* e = getenv("STARCOP_STEM_MFMA"); return e && atoi(e) == 0; }();      // (same-box A/B)
  if (Cin <= 4 && !mfma_off && Wout % 4 == 0 && ((uintptr_t)out & 15) == 0)
    hipLaunchKernelGGL(k_stem_fwd4m, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), w, out, Cin, Hin, Win, Hout, Wout, stats);
  else if (Cin <= 4) hipLaunchKernelGGL(k_stem_fwd<4>, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), w, out, Cin, Hin, Win, Hout, Wout, stats);
  else hipLaunchKernelGGL(k_stem_fwd<STEM_MAXCI>, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), w, out, Cin, Hin, Win, Hout, Wout, stats);
  SC_LAUNCH_OK("sc_stem_conv_fwd");
  return SC_OK;
}

extern "C" size_t sc_stem_wgrad_workspace_floats(int N, int Cin, int Hin, int Win) {
  const int Hout = (Hin - 1) / 2 + 1, Wout = (Win - 1) / 2 + 1;
  const int nb = stem_blocks(N, Hout, Wout);          // one partial row per work-group
  const size_t E = (size_t)STEM_CO * Cin * 9;
  return (size_t)nb * E + sc_reduce_scratch_floats(nb, E);
}

extern "C" int sc_stem_conv_wgrad(const sc_src* dy, const sc_src* in, float* part, size_t part_floats, float* dw, int N,
                                  int Cin, int Hin, int Win, sc_stream stream) {
  SC_REQUIRE(dy && in && in->C == Cin && dy->C == STEM_CO, "sc_stem_conv_wgrad: bad source channels");
  SC_REQUIRE(Cin >= 1 && Cin <= STEM_MAXCI, "sc_stem_conv_wgrad: Cin must be in [1,%d]", STEM_MAXCI);
  SC_REQUIRE(part_floats >= sc_stem_wgrad_workspace_floats(N, Cin, Hin, Win), "sc_stem_conv_wgrad: workspace too small");
  const int Hout = (Hin - 1) / 2 + 1, Wout = (Win - 1) / 2 + 1;
  const int nb = stem_blocks(N, Hout, Wout);
  const size_t E = (size_t)STEM_CO * Cin * 9;
  hipStream_t st = (hipStream_t)stream;
  const int njb = (Cin * 9 + 15) / 16;
  if (njb <= 3) hipLaunchKernelGGL((k_stem_wgrad<3>), dim3(nb), dim3(256), 0, st, to_srcd(*dy), to_srcd(*in), part, N, Cin, Hin, Win, Hout, Wout);
  else hipLaunchKernelGGL((k_stem_wgrad<5>), dim3(nb), dim3(256), 0, st, to_srcd(*dy), to_srcd(*in), part, N, Cin, Hin, Win, Hout, Wout);
  SC_LAUNCH_OK("sc_stem_conv_wgrad");
  return sc_reduce_rows(part, nb, E, part + (size_t)nb * E, dw, st);
}

extern "C" int sc_head_conv_fwd(const sc_src* in, const float* w, const float* bias, float* out, int N, int Cin,
                                int H, int W, sc_stream stream) {
  SC_REQUIRE(in && in->C == Cin, "sc_head_conv_fwd: source channels != Cin");
  SC_REQUIRE(Cin >= 1 && Cin <= HEAD_MAXCI, "sc_head_conv_fwd: Cin must be in [1,%d]", HEAD_MAXCI);
  SC_REQUIRE((in->mode == SC_SRC_RAW || in->mode == SC_SRC_AFFINE) && in->up == 0, "sc_head_conv_fwd: unsupported source mode");
  dim3 grid(((W + HT_C - 1) / HT_C) * ((H + HT_R - 1) / HT_R), 1, N);
  if (Cin == 16) {
    dim3 g16(((W + 63) / 64) * ((H + 15) / 16), 1, N);
    const bool vec = (W % 4 == 0) && (((uintptr_t)in->x | (uintptr_t)out) & 15) == 0;
    if (vec) hipLaunchKernelGGL(k_head_fwd16v, g16, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), w, bias, out, H, W);
    else hipLaunchKernelGGL(k_head_fwd16, g16, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), w, bias, out, H, W);
  } else {
    hipLaunchKernelGGL((k_head_fwd<0>), grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), w, bias, out, Cin, H, W);
  }
  SC_LAUNCH_OK("sc_head_conv_fwd");
  return SC_OK;
}

extern "C" int sc_head_conv_dgrad(const float* dlogits, const float* w, float* gin, int N, int Cin, int H, int W,
                                  sc_stream stream) {
  SC_REQUIRE(Cin >= 1 && Cin <= HEAD_MAXCI, "sc_head_conv_dgrad: Cin must be in [1,%d]", HEAD_MAXCI);
  dim3 grid(((W + HT_C - 1) / HT_C) * ((H + HT_R - 1) / HT_R), 1, N);
  hipLaunchKernelGGL(k_head_dgrad, grid, dim3(256), 0, (hipStream_t)stream, dlogits, w, gin, Cin, H, W);
  SC_LAUNCH_OK("sc_head_conv_dgrad");
  return SC_OK;
}

extern "C" size_t sc_head_wgrad_workspace_floats(int N, int Cin, int H, int W) {
  (void)N; (void)H; (void)W;
  // Cin*9 + 1 double accumulators (sc_head_conv_wgrad) or the partial rows of sc_head_conv_bwd
  const size_t a = 2 * ((size_t)Cin * 9 + 1) + 2, b = (size_t)HB_MAXWG * HB_ROW;
  return a > b ? a : b;
}

// dW[0][ci][tap] = sum_px dlogits[px] * act(in)[ci][px + d(tap)]: the depthwise weight-gradient kernel with the single
// dlogits plane shared by all input channels; dbias = sum dlogits.
extern "C" int sc_head_conv_wgrad(const float* dlogits, const sc_src* in, float* part, size_t part_floats, float* dw,
                                  float* dbias, int N, int Cin, int H, int W, sc_stream stream) {
  SC_REQUIRE(in && in->C == Cin, "sc_head_conv_wgrad: source channels != Cin");
  SC_REQUIRE(Cin >= 1 && Cin <= HEAD_MAXCI, "sc_head_conv_wgrad: Cin must be in [1,%d]", HEAD_MAXCI);
  SC_REQUIRE((in->mode == SC_SRC_RAW || in->mode == SC_SRC_AFFINE) && in->up == 0, "sc_head_conv_wgrad: unsupported source mode");
  SC_REQUIRE(part_floats >= sc_head_wgrad_workspace_floats(N, Cin, H, W), "sc_head_conv_wgrad: workspace too small");
  SC_REQUIRE(((uintptr_t)part & 7) == 0, "sc_head_conv_wgrad: workspace must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  double* acc = reinterpret_cast<double*>(part);
  const size_t nacc = (size_t)Cin * 9 + 1;
  if (hipMemsetAsync(acc, 0, nacc * sizeof(double), st) != hipSuccess) { sc_set_error("sc_head_conv_wgrad: memset failed"); return SC_ERR_LAUNCH; }
  SrcD dy = empty_srcd();
  dy.x = dlogits; dy.C = 1; dy.mode = SC_SRC_RAW;
  const int stride = 1, Hout = H, Wout = W;
  const long T = (long)N * dw_tiles(Hout, Wout);
  const long want = 4096 / Cin > 0 ? 4096 / Cin : 1;
  dim3 grid((unsigned)(T < want ? T : want), Cin);
  SC_DW_DISPATCH(k_dw_wgrad, Wout, grid, dy, to_srcd(*in), acc, N, Cin, H, W, Hout, Wout, 1);
  SC_LAUNCH_OK("sc_head_conv_wgrad");
  const size_t npx = (size_t)N * H * W;
  hipLaunchKernelGGL(k_sum_f32_to_f64, dim3((unsigned)((npx + 4095) / 4096 > 512 ? 512 : (npx + 4095) / 4096)), dim3(256), 0, st, dlogits, npx, acc + Cin * 9);
  SC_LAUNCH_OK("sc_head_conv_wgrad(bias)");
  hipLaunchKernelGGL(k_cast_f64_f32, dim3(1), dim3(256), 0, st, acc, dw, (size_t)Cin * 9);
  if (dbias) hipLaunchKernelGGL(k_cast_f64_f32, dim3(1), dim3(64), 0, st, acc + Cin * 9, dbias, (size_t)1);
  SC_LAUNCH_OK("sc_head_conv_wgrad(cast)");
  return SC_OK;
}
