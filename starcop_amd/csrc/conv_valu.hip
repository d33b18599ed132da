// HBM-bound convolutions of the MobileNetV2-U-Net that do not belong on the matrix cores:
//   depthwise 3x3 (stride 1|2)         -- torchvision InvertedResidual conv.{0|1}.0
//   stem 3x3 stride 2, Cin(<=8) -> 32  -- encoder.features.0.0 (+ DataNormalizer fused on load)
//   head 3x3, Cin(16) -> 1, bias       -- segmentation_head.0
// forward / backward-data / backward-weight, all with the "normalise on load" prologues of
// sc_common.h.  Reference call site of the whole network: starcop/models/model_module.py:244-251.
#include "sc_common.h"

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
// One (n, c) plane tile per block: TW x TH outputs (TW in {32,16}, TH = 256/TW), one output per thread.  The input
// patch is staged ONCE through LDS with the producer's BatchNorm+ReLU6 applied on the way (branch-free prologue,
// clamped unconditional loads), so every HBM element is read once per tile and transformed once.
template <int S, int TW>
__global__ __launch_bounds__(256) void k_dw_fwd(const SrcD in, const float* __restrict__ w, float* __restrict__ out,
                                                int C, int Hin, int Win, int Hout, int Wout, float* stats) {
  constexpr int TH = 256 / TW;
  constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PWP = PW | 1;
  __shared__ float s_x[PH * PWP];
  __shared__ float s_tmp[8];
  const int c = blockIdx.y, n = blockIdx.z;
  const int tiles_x = (Wout + TW - 1) / TW;
  const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
  float sc = 1.f, sh = 0.f;
  if (in.mode != SC_SRC_RAW) { sc = in.cst[(size_t)c * SC_CST]; sh = in.cst[(size_t)c * SC_CST + 1]; }
  const float lo = sc_act_lo(in.act), hi = sc_act_hi(in.act);
  const float* xb = in.x + ((size_t)n * C + c) * Hin * Win;
  for (int e = threadIdx.x; e < PH * PW; e += 256) {
    const int r = e / PW, cc = e - r * PW;
    const int iy = ty0 * S - 1 + r, ix = tx0 * S - 1 + cc;
    const bool ok = (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
    const float v = sc_pro_affine(xb[ok ? iy * Win + ix : 0], sc, sh, lo, hi);
    s_x[r * PWP + cc] = ok ? v : 0.f;
  }
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];
  __syncthreads();
  const int ty = threadIdx.x / TW, tx = threadIdx.x % TW;
  const int oy = ty0 + ty, ox = tx0 + tx;
  const bool ok = (oy < Hout) && (ox < Wout);
  float acc = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) acc = fmaf(wk[kh * 3 + kw], s_x[(ty * S + kh) * PWP + tx * S + kw], acc);
  if (ok) out[((size_t)n * C + c) * Hout * Wout + (size_t)oy * Wout + ox] = acc;
  if (stats) {
    float v[2] = {ok ? acc : 0.f, ok ? acc * acc : 0.f};
    block_sum<2>(v, s_tmp);
    if (threadIdx.x < 2) stats[(stat_row() * C + c) * 2 + threadIdx.x] = v[threadIdx.x];
  }
}

// dy patch (BatchNorm/ReLU6 backward applied on load) staged once; dx tile = TW x TH input pixels
template <int S, int TW>
__global__ __launch_bounds__(256) void k_dw_dgrad(const SrcD dy, const float* __restrict__ w, float* __restrict__ dx,
                                                  int accum, int C, int Hin, int Win, int Hout, int Wout) {
  constexpr int TH = 256 / TW;
  constexpr int PH = (S == 1) ? TH + 2 : TH / 2 + 1, PW = (S == 1) ? TW + 2 : TW / 2 + 1, PWP = PW | 1;
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
  const float* yb = (dy.mode == SC_SRC_BNBWD) ? dy.aux + obase : gb;
  const bool bnb = dy.mode == SC_SRC_BNBWD;
  for (int e = threadIdx.x; e < PH * PW; e += 256) {
    const int r = e / PW, cc = e - r * PW;
    const int oy = oyb + r, ox = oxb + cc;
    const bool ok = (oy >= 0) && (oy < Hout) && (ox >= 0) && (ox < Wout);
    const int o = ok ? oy * Wout + ox : 0;
    const float g = gb[o], yv = yb[o];
    const float v = bnb ? sc_pro_bnbwd(g, yv, c0.x, c0.y, c0.z, c0.w, c4, lo, hi) : sc_pro_affine(g, c0.x, c0.y, lo, hi);
    s_d[r * PWP + cc] = ok ? v : 0.f;
  }
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];
  __syncthreads();
  const int ty = threadIdx.x / TW, tx = threadIdx.x % TW;
  const int iy = iy0 + ty, ix = ix0 + tx;
  if (iy >= Hin || ix >= Win) return;
  float acc = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int t = iy + 1 - kh;                 // = oy * S
    if (S == 2 && (t & 1)) continue;
    const int r = (S == 1 ? t : t / 2) - oyb;  // t >= -1 only when S == 1 (zero-padded patch row)
    if (S == 2 && t < 0) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int u = ix + 1 - kw;
      if (S == 2 && ((u & 1) || u < 0)) continue;
      const int cc = (S == 1 ? u : u / 2) - oxb;
      acc = fmaf(wk[kh * 3 + kw], s_d[r * PWP + cc], acc);
    }
  }
  const size_t o = ((size_t)n * C + c) * Hin * Win + (size_t)iy * Win + ix;
  dx[o] = accum ? dx[o] + acc : acc;
}

// dW[c][tap] += sum over this block's (n, tile) list of dy * x; 9 double atomics per block
template <int S, int TW>
__global__ __launch_bounds__(256) void k_dw_wgrad(const SrcD dy, const SrcD in, double* __restrict__ dw_acc, int N, int C,
                                                  int Hin, int Win, int Hout, int Wout) {
  constexpr int TH = 256 / TW;
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
  if (dy.mode != SC_SRC_RAW) { c0 = *reinterpret_cast<const float4*>(dy.cst + (size_t)c * SC_CST); c4 = dy.cst[(size_t)c * SC_CST + 4]; }
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
    __syncthreads();
    for (int e = threadIdx.x; e < PH * PW; e += 256) {
      const int r = e / PW, cc = e - r * PW;
      const int iy = ty0 * S - 1 + r, ix = tx0 * S - 1 + cc;
      const bool ok = (iy >= 0) && (iy < Hin) && (ix >= 0) && (ix < Win);
      const float v = sc_pro_affine(xb[ok ? iy * Win + ix : 0], xs, xh, xlo, xhi);
      s_x[r * PWP + cc] = ok ? v : 0.f;
    }
    const int oy = ty0 + ty, ox = tx0 + tx;
    const bool ok = (oy < Hout) && (ox < Wout);
    const size_t o = ((size_t)n * C + c) * Hout * Wout + (ok ? (size_t)oy * Wout + ox : 0);
    const float g = dy.x[o];
    const float yv = bnb ? dy.aux[o] : g;
    float dyv = bnb ? sc_pro_bnbwd(g, yv, c0.x, c0.y, c0.z, c0.w, c4, dlo, dhi) : sc_pro_affine(g, c0.x, c0.y, dlo, dhi);
    dyv = ok ? dyv : 0.f;
    __syncthreads();
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) prod[kh * 3 + kw] = fmaf(dyv, s_x[(ty * S + kh) * PWP + tx * S + kw], prod[kh * 3 + kw]);
  }
  block_sum<9>(prod, s_tmp);
  if (threadIdx.x < 9) atomicAdd(&dw_acc[c * 9 + threadIdx.x], (double)prod[threadIdx.x]);
}

// ---------------------------------------------------------------- stem (3x3 s2, Cin<=8 -> 32)
constexpr int STEM_CO = 32;
constexpr int STEM_MAXCI = 8;

__global__ __launch_bounds__(256) void k_stem_fwd(const SrcD in, const float* __restrict__ w, float* __restrict__ out,
                                                  int Cin, int Hin, int Win, int Hout, int Wout, float* stats) {
  __shared__ float s_w[STEM_CO * STEM_MAXCI * 9];
  __shared__ float s_red[4][STEM_CO][2];
  const int n = blockIdx.z;
  for (int i = threadIdx.x; i < STEM_CO * Cin * 9; i += 256) s_w[i] = w[i];
  const int tiles_x = (Wout + 15) >> 4;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy = ty * 16 + (threadIdx.x >> 4), ox = tx * 16 + (threadIdx.x & 15);
  const bool ok = (oy < Hout) && (ox < Wout);
  float v[STEM_MAXCI * 9];
#pragma unroll
  for (int i = 0; i < STEM_MAXCI * 9; ++i) v[i] = 0.f;
  if (ok) {
#pragma unroll
    for (int ci = 0; ci < STEM_MAXCI; ++ci) {
      if (ci < Cin) {
        const size_t ibase = ((size_t)n * Cin + ci) * Hin * Win;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int iy = oy * 2 + kh - 1;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int ix = ox * 2 + kw - 1;
            if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) v[ci * 9 + kh * 3 + kw] = load_src(in, ibase + (size_t)iy * Win + ix, ci);
          }
        }
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t HWo = (size_t)Hout * Wout;
  for (int co = 0; co < STEM_CO; ++co) {
    float acc = 0.f;
#pragma unroll
    for (int ci = 0; ci < STEM_MAXCI; ++ci) {
      if (ci < Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(s_w[(co * Cin + ci) * 9 + t], v[ci * 9 + t], acc);
      }
    }
    if (ok) out[((size_t)n * STEM_CO + co) * HWo + (size_t)oy * Wout + ox] = acc;
    if (stats) {
      const float a = ok ? acc : 0.f;
      const float s = wave_sum(a), ss = wave_sum(a * a);
      if (lane == 0) { s_red[wave][co][0] = s; s_red[wave][co][1] = ss; }
    }
  }
  if (stats) {
    __syncthreads();
    if (threadIdx.x < STEM_CO * 2) {
      const int co = threadIdx.x >> 1, k = threadIdx.x & 1;
      const float t = s_red[0][co][k] + s_red[1][co][k] + s_red[2][co][k] + s_red[3][co][k];
      stats[(stat_row() * STEM_CO + co) * 2 + k] = t;
    }
  }
}

// dW[co][ci][tap] = sum dy[co][oy][ox] * in[ci][2oy+kh-1][2ox+kw-1]; one partial row per block
// tile = 8 x 32 output pixels; thread = (co = tid&31, group = tid>>5) owns combos j = group + 8k
__global__ __launch_bounds__(256) void k_stem_wgrad(const SrcD dy, const SrcD in, float* __restrict__ part, int N, int Cin,
                                                    int Hin, int Win, int Hout, int Wout) {
  constexpr int TR = 8, TC = 32;
  constexpr int IR = 2 * TR + 1, IC = 2 * TC + 1, ICP = IC + 2;
  __shared__ float s_dy[STEM_CO * (TR * TC + 1)];
  __shared__ float s_x[STEM_MAXCI * IR * ICP];
  const int co = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int ncomb = Cin * 9;
  constexpr int MAXK = (STEM_MAXCI * 9 + 7) / 8;
  float acc[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) acc[k] = 0.f;
  const int tiles_x = (Wout + TC - 1) / TC, tiles_y = (Hout + TR - 1) / TR;
  const long T = (long)N * tiles_x * tiles_y;
  for (long t = blockIdx.x; t < T; t += gridDim.x) {
    const int n = (int)(t / (tiles_x * tiles_y));
    const int rem = (int)(t - (long)n * tiles_x * tiles_y);
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int oy0 = ty * TR, ox0 = tx * TC;
    __syncthreads();
    for (int i = threadIdx.x; i < STEM_CO * TR * TC; i += 256) {
      const int ch = i / (TR * TC), px = i - ch * (TR * TC);
      const int oy = oy0 + px / TC, ox = ox0 + (px % TC);
      float v = 0.f;
      if (oy < Hout && ox < Wout) v = load_src(dy, ((size_t)n * STEM_CO + ch) * Hout * Wout + (size_t)oy * Wout + ox, ch);
      s_dy[ch * (TR * TC + 1) + px] = v;
    }
    for (int i = threadIdx.x; i < Cin * IR * IC; i += 256) {
      const int ci = i / (IR * IC), e = i - ci * (IR * IC);
      const int r = e / IC, cc = e - r * IC;
      const int iy = 2 * oy0 - 1 + r, ix = 2 * ox0 - 1 + cc;
      float v = 0.f;
      if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) v = load_src(in, ((size_t)n * Cin + ci) * Hin * Win + (size_t)iy * Win + ix, ci);
      s_x[(ci * IR + r) * ICP + cc] = v;
    }
    __syncthreads();
    for (int px = 0; px < TR * TC; ++px) {
      const float a = s_dy[co * (TR * TC + 1) + px];
      const int py = px / TC, pxx = px % TC;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int j = grp + 8 * k;
        if (j < ncomb) {
          const int ci = j / 9, tap = j - ci * 9;
          const int kh = tap / 3, kw = tap - kh * 3;
          acc[k] = fmaf(a, s_x[(ci * IR + 2 * py + kh) * ICP + 2 * pxx + kw], acc[k]);
        }
      }
    }
  }
  float* pr = part + (size_t)blockIdx.x * (STEM_CO * ncomb);
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const int j = grp + 8 * k;
    if (j < ncomb) pr[co * ncomb + j] = acc[k];   // [co][ci][tap] (OIHW order)
  }
}

// ---------------------------------------------------------------- head (3x3, Cin<=32 -> 1, bias)
constexpr int HEAD_MAXCI = 32;
constexpr int HT_R = 8, HT_C = 32;

__global__ __launch_bounds__(256) void k_head_fwd(const SrcD in, const float* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ out, int Cin, int H, int W) {
  constexpr int PR = HT_R + 2, PC = HT_C + 2;
  __shared__ float s_in[HEAD_MAXCI * PR * PC];
  __shared__ float s_w[HEAD_MAXCI * 9];
  const int n = blockIdx.z;
  const int tiles_x = (W + HT_C - 1) / HT_C;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * HT_R, x0 = tx * HT_C;
  for (int i = threadIdx.x; i < Cin * 9; i += 256) s_w[i] = w[i];
  for (int i = threadIdx.x; i < Cin * PR * PC; i += 256) {
    const int ci = i / (PR * PC), e = i - ci * (PR * PC);
    const int r = e / PC, cc = e - r * PC;
    const int y = y0 - 1 + r, x = x0 - 1 + cc;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) v = load_src(in, ((size_t)n * Cin + ci) * H * W + (size_t)y * W + x, ci);
    s_in[i] = v;
  }
  __syncthreads();
  const int py = threadIdx.x >> 5, px = threadIdx.x & 31;
  const int y = y0 + py, x = x0 + px;
  if (y >= H || x >= W) return;
  float acc = bias ? bias[0] : 0.f;
  for (int ci = 0; ci < Cin; ++ci) {
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

// part[block][Cin*9 + 1]: dW[ci][tap] then dbias
__global__ __launch_bounds__(256) void k_head_wgrad(const float* __restrict__ dl, const SrcD in, float* __restrict__ part,
                                                    int N, int Cin, int H, int W) {
  constexpr int PR = HT_R + 2, PC = HT_C + 3;   // 35: odd pitch
  __shared__ float s_in[HEAD_MAXCI * PR * PC];
  __shared__ float s_dl[HT_R * HT_C];
  const int nout = Cin * 9 + 1;
  const int t_id = threadIdx.x;
  const bool is_w = t_id < Cin * 9, is_b = t_id == Cin * 9;
  const int ci = is_w ? t_id / 9 : 0, tap = is_w ? t_id - ci * 9 : 0;
  const int kh = tap / 3, kw = tap - kh * 3;
  float acc = 0.f;
  const int tiles_x = (W + HT_C - 1) / HT_C, tiles_y = (H + HT_R - 1) / HT_R;
  const long T = (long)N * tiles_x * tiles_y;
  for (long t = blockIdx.x; t < T; t += gridDim.x) {
    const int n = (int)(t / (tiles_x * tiles_y));
    const int rem = (int)(t - (long)n * tiles_x * tiles_y);
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int y0 = ty * HT_R, x0 = tx * HT_C;
    __syncthreads();
    for (int i = t_id; i < Cin * PR * (HT_C + 2); i += 256) {
      const int c = i / (PR * (HT_C + 2)), e = i - c * (PR * (HT_C + 2));
      const int r = e / (HT_C + 2), cc = e - r * (HT_C + 2);
      const int y = y0 - 1 + r, x = x0 - 1 + cc;
      float v = 0.f;
      if (y >= 0 && y < H && x >= 0 && x < W) v = load_src(in, ((size_t)n * Cin + c) * H * W + (size_t)y * W + x, c);
      s_in[(c * PR + r) * PC + cc] = v;
    }
    {
      const int py = t_id >> 5, px = t_id & 31;
      const int y = y0 + py, x = x0 + px;
      s_dl[t_id] = (y < H && x < W) ? dl[(size_t)n * H * W + (size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    if (is_w) {
      for (int px = 0; px < HT_R * HT_C; ++px) {
        const int py = px >> 5, pxx = px & 31;
        acc = fmaf(s_dl[px], s_in[(ci * PR + py + kh) * PC + pxx + kw], acc);
      }
    } else if (is_b) {
      for (int px = 0; px < HT_R * HT_C; ++px) acc += s_dl[px];
    }
  }
  if (is_w || is_b) part[(size_t)blockIdx.x * nout + t_id] = acc;
}

__global__ void k_split_head(const float* __restrict__ red, float* dw, float* dbias, int nw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nw) dw[i] = red[i];
  else if (i == nw && dbias) dbias[0] = red[nw];
}

__global__ void k_cast_f64_f32(const double* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

int stem_blocks(int N, int Hout, int Wout) {
  long T = (long)N * ((Wout + 31) / 32) * ((Hout + 7) / 8);
  return (int)(T < 512 ? T : 512);
}
int head_blocks(int N, int H, int W) {
  long T = (long)N * ((W + HT_C - 1) / HT_C) * ((H + HT_R - 1) / HT_R);
  return (int)(T < 1024 ? T : 1024);
}

}  // namespace

#define SC_DW_DISPATCH(KERNEL, W_, ...)                                                               \
  do {                                                                                                 \
    if (stride == 1 && (W_) > 16) hipLaunchKernelGGL((KERNEL<1, 32>), grid32, dim3(256), 0, st, __VA_ARGS__);      \
    else if (stride == 1) hipLaunchKernelGGL((KERNEL<1, 16>), grid16, dim3(256), 0, st, __VA_ARGS__);  \
    else if ((W_) > 16) hipLaunchKernelGGL((KERNEL<2, 32>), grid32, dim3(256), 0, st, __VA_ARGS__);    \
    else hipLaunchKernelGGL((KERNEL<2, 16>), grid16, dim3(256), 0, st, __VA_ARGS__);                   \
  } while (0)

extern "C" int sc_dwconv3x3_fwd(const sc_src* in, const float* w, float* out, int N, int C, int Hin, int Win,
                                int stride, float* stats, sc_stream stream) {
  SC_REQUIRE(in && in->C == C, "sc_dwconv3x3_fwd: source channels != C");
  SC_REQUIRE(stride == 1 || stride == 2, "sc_dwconv3x3_fwd: stride must be 1 or 2");
  SC_REQUIRE((in->mode == SC_SRC_RAW || in->mode == SC_SRC_AFFINE) && in->up == 0, "sc_dwconv3x3_fwd: unsupported source mode");
  const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid32(((Wout + 31) / 32) * ((Hout + 7) / 8), C, N), grid16(((Wout + 15) / 16) * ((Hout + 15) / 16), C, N);
  SC_DW_DISPATCH(k_dw_fwd, Wout, to_srcd(*in), w, out, C, Hin, Win, Hout, Wout, stats);
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
  dim3 grid32(((Win + 31) / 32) * ((Hin + 7) / 8), C, N), grid16(((Win + 15) / 16) * ((Hin + 15) / 16), C, N);
  SC_DW_DISPATCH(k_dw_dgrad, Win, to_srcd(*dy), w, dx, accum, C, Hin, Win, Hout, Wout);
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
  const long t32 = (long)N * ((Wout + 31) / 32) * ((Hout + 7) / 8), t16 = (long)N * ((Wout + 15) / 16) * ((Hout + 15) / 16);
  const long want = 4096 / C > 0 ? 4096 / C : 1;
  dim3 grid32((unsigned)(t32 < want ? t32 : want), C), grid16((unsigned)(t16 < want ? t16 : want), C);
  SC_DW_DISPATCH(k_dw_wgrad, Wout, to_srcd(*dy), to_srcd(*in), dw_acc, N, C, Hin, Win, Hout, Wout);
  SC_LAUNCH_OK("sc_dwconv3x3_wgrad");
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
  dim3 grid(((Wout + 15) / 16) * ((Hout + 15) / 16), 1, N);
  hipLaunchKernelGGL(k_stem_fwd, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), w, out, Cin, Hin, Win, Hout, Wout, stats);
  SC_LAUNCH_OK("sc_stem_conv_fwd");
  return SC_OK;
}

extern "C" size_t sc_stem_wgrad_workspace_floats(int N, int Cin, int Hin, int Win) {
  const int Hout = (Hin - 1) / 2 + 1, Wout = (Win - 1) / 2 + 1;
  const int nb = stem_blocks(N, Hout, Wout);
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
  hipLaunchKernelGGL(k_stem_wgrad, dim3(nb), dim3(256), 0, (hipStream_t)stream, to_srcd(*dy), to_srcd(*in), part, N, Cin, Hin, Win, Hout, Wout);
  SC_LAUNCH_OK("sc_stem_conv_wgrad");
  return sc_reduce_rows(part, nb, E, part + (size_t)nb * E, dw, (hipStream_t)stream);
}

extern "C" int sc_head_conv_fwd(const sc_src* in, const float* w, const float* bias, float* out, int N, int Cin,
                                int H, int W, sc_stream stream) {
  SC_REQUIRE(in && in->C == Cin, "sc_head_conv_fwd: source channels != Cin");
  SC_REQUIRE(Cin >= 1 && Cin <= HEAD_MAXCI, "sc_head_conv_fwd: Cin must be in [1,%d]", HEAD_MAXCI);
  SC_REQUIRE(in->mode != SC_SRC_BNBWD && in->up == 0, "sc_head_conv_fwd: unsupported source mode");
  dim3 grid(((W + HT_C - 1) / HT_C) * ((H + HT_R - 1) / HT_R), 1, N);
  hipLaunchKernelGGL(k_head_fwd, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), w, bias, out, Cin, H, W);
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
  const int nb = head_blocks(N, H, W);
  const size_t E = (size_t)Cin * 9 + 1;
  return (size_t)nb * E + sc_reduce_scratch_floats(nb, E) + E;
}

extern "C" int sc_head_conv_wgrad(const float* dlogits, const sc_src* in, float* part, size_t part_floats, float* dw,
                                  float* dbias, int N, int Cin, int H, int W, sc_stream stream) {
  SC_REQUIRE(in && in->C == Cin, "sc_head_conv_wgrad: source channels != Cin");
  SC_REQUIRE(Cin >= 1 && Cin * 9 + 1 <= 256, "sc_head_conv_wgrad: Cin must be in [1,28]");
  SC_REQUIRE(part_floats >= sc_head_wgrad_workspace_floats(N, Cin, H, W), "sc_head_conv_wgrad: workspace too small");
  const int nb = head_blocks(N, H, W);
  const size_t E = (size_t)Cin * 9 + 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_head_wgrad, dim3(nb), dim3(256), 0, st, dlogits, to_srcd(*in), part, N, Cin, H, W);
  SC_LAUNCH_OK("sc_head_conv_wgrad");
  float* scratch = part + (size_t)nb * E;
  float* red = scratch + sc_reduce_scratch_floats(nb, E);
  int rc = sc_reduce_rows(part, nb, E, scratch, red, st);
  if (rc != SC_OK) return rc;
  hipLaunchKernelGGL(k_split_head, dim3(((int)E + 255) / 256), dim3(256), 0, st, red, dw, dbias, Cin * 9);
  SC_LAUNCH_OK("sc_head_split");
  return SC_OK;
}
