// Evaluation masks: thresholded predictions, binary opening with a 3x3 structuring element and the confusion counts of
// many thresholds in one pass (reference: starcop/baselines.py:25-57, starcop/validation.py:38,106,121-127).
// HBM-bound: every prediction / label pixel is read once whatever the number of thresholds.
#include "sc_common.h"

namespace {

constexpr int TW = 64, TH = 16;          // output tile per work-group (256 threads: 64 x 4, four rows per thread)
constexpr int MAXT = 32;

struct ThrConfP {
  const float* pred;
  const float* target;
  const unsigned char* ignore;
  float thr[MAXT];
  int T, se, N, H, W;
  long long* out;          // [N][H][W] (T == 1) or null
  long long* tile_count;   // [N] or null
  long long* cm;           // [N][T][4] or null
  long long* invalid;      // [1] or null
};

// A flat structuring element commutes with thresholding (threshold decomposition): opening(pred > t) == (grey-scale
// opening of pred) > t for every t, with +inf outside the image for the erosion, -inf for the dilation and NaN -> -inf
// (NaN > t is false).  The grey opening is formed once per pixel; every threshold is then a single compare.
template <bool OPEN>
__global__ __launch_bounds__(256) void k_thrconf(const ThrConfP p) {
  __shared__ float s_p[OPEN ? (TH + 4) * (TW + 4) : 1];
  __shared__ float s_e[OPEN ? (TH + 2) * (TW + 2) : 1];     // eroded, outside the image = -inf
  __shared__ int s_cm[MAXT * 2];
  __shared__ int s_tot[2];
  __shared__ int s_cnt;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int n = blockIdx.z, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int H = p.H, W = p.W;
  const size_t plane = (size_t)n * H * W;
  const float INF = __builtin_inff();
  for (int i = tid; i < MAXT * 2; i += 256) s_cm[i] = 0;
  if (tid < 2) s_tot[tid] = 0;
  if (tid == 0) s_cnt = 0;

  if (OPEN) {
    for (int i = tid; i < (TH + 4) * (TW + 4); i += 256) {
      const int yy = y0 - 2 + i / (TW + 4), xx = x0 - 2 + i % (TW + 4);
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      float v = INF;
      if (in) { v = p.pred[plane + (size_t)yy * W + xx]; v = (v != v) ? -INF : v; }
      s_p[i] = v;
    }
  }
  // labels of this thread's four pixels
  bool yv[4], ok[4];
  float pv[4];
  int bad = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int yy = y0 + ty + 4 * j, xx = x0 + tx;
    const bool in = yy < H && xx < W;
    const size_t idx = plane + (size_t)(in ? yy : 0) * W + (in ? xx : 0);
    ok[j] = in;
    yv[j] = false;
    pv[j] = -INF;
    if (!OPEN && in) pv[j] = p.pred[idx];
    if (p.cm && in) {
      const long long yl = (long long)p.target[idx];
      yv[j] = yl == 1;
      if (yl != 0 && yl != 1) { ok[j] = false; ++bad; }
      if (p.ignore && p.ignore[idx]) ok[j] = false;
    }
  }
  if (p.invalid && bad) atomicAdd(reinterpret_cast<unsigned long long*>(p.invalid), (unsigned long long)bad);
  __syncthreads();
  if (OPEN) {
    for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) {
      const int ey = i / (TW + 2), ex = i % (TW + 2);
      const int yy = y0 - 1 + ey, xx = x0 - 1 + ex;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      float e = INF;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if ((p.se >> (3 * r + c)) & 1) e = fminf(e, s_p[(ey + r) * (TW + 4) + ex + c]);
      s_e[i] = in ? e : -INF;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int oy = ty + 4 * j;
      float d = -INF;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if ((p.se >> (3 * (2 - r) + (2 - c))) & 1) d = fmaxf(d, s_e[(oy + r) * (TW + 2) + tx + c]);
      pv[j] = d;
    }
  }

  // label totals of the wave (valid pixels): with the positives per threshold they give all four cells
  unsigned long long ym[4], vm[4];
  if (p.cm) {
    int t0 = 0, t1 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vm[j] = __ballot(ok[j]);
      ym[j] = __ballot(yv[j]) & vm[j];
      t1 += __popcll(ym[j]);
      t0 += __popcll(vm[j] & ~ym[j]);
    }
    if ((tid & 63) == 0) { atomicAdd(&s_tot[0], t0); atomicAdd(&s_tot[1], t1); }
  }
  for (int t = 0; t < p.T; ++t) {
    const float thr = p.thr[t];
    bool m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = pv[j] > thr;
    if (p.out || p.tile_count) {        // single-threshold mask output
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yy = y0 + ty + 4 * j, xx = x0 + tx;
        if (yy < H && xx < W) {
          if (p.out) p.out[plane + (size_t)yy * W + xx] = m[j] ? 1 : 0;
          cnt += m[j] ? 1 : 0;
        }
      }
      if (p.tile_count) {
        cnt = (int)wave_sum((float)cnt);
        if ((tid & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
      }
    }
    if (p.cm) {
      int c01 = 0, c11 = 0;             // predicted positive with label 0 / label 1
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned long long mm = __ballot(m[j]);
        c11 += __popcll(mm & ym[j]);
        c01 += __popcll(mm & vm[j] & ~ym[j]);
      }
      if ((tid & 63) == 0) {
        if (c01) atomicAdd(&s_cm[t * 2 + 0], c01);
        if (c11) atomicAdd(&s_cm[t * 2 + 1], c11);
      }
    }
  }
  __syncthreads();
  if (p.cm && tid < p.T * 4) {
    const int t = tid >> 2, cell = tid & 3, lab = cell >> 1, pos = cell & 1;
    const int npos = s_cm[t * 2 + lab];
    const int v = pos ? npos : s_tot[lab] - npos;
    if (v) atomicAdd(reinterpret_cast<unsigned long long*>(p.cm + ((size_t)n * p.T) * 4 + tid), (unsigned long long)v);
  }
  if (p.tile_count && tid == 0 && s_cnt)
    atomicAdd(reinterpret_cast<unsigned long long*>(p.tile_count + n), (unsigned long long)s_cnt);
}

int launch(const ThrConfP& p, hipStream_t st) {
  dim3 grid((p.W + TW - 1) / TW, (p.H + TH - 1) / TH, p.N);
  if (p.se) hipLaunchKernelGGL(k_thrconf<true>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(k_thrconf<false>, grid, dim3(256), 0, st, p);
  SC_LAUNCH_OK("k_thrconf");
  return SC_OK;
}

}  // namespace

extern "C" int sc_binary_opening(const float* pred, float threshold, int se_bits, int64_t* out, int64_t* tile_count,
                                 int N, int H, int W, sc_stream stream) {
  SC_REQUIRE(pred && (out || tile_count), "sc_binary_opening: null pointer");
  SC_REQUIRE(N > 0 && H > 0 && W > 0 && N <= 65535, "sc_binary_opening: bad dims N=%d H=%d W=%d", N, H, W);
  SC_REQUIRE(se_bits >= 0 && se_bits < 512, "sc_binary_opening: se_bits=%d is not a 3x3 structuring element", se_bits);
  ThrConfP p{};
  p.pred = pred; p.T = 1; p.thr[0] = threshold; p.se = se_bits; p.N = N; p.H = H; p.W = W;
  p.out = reinterpret_cast<long long*>(out); p.tile_count = reinterpret_cast<long long*>(tile_count);
  return launch(p, (hipStream_t)stream);
}

extern "C" int sc_threshold_confusion(const float* pred, const float* target, const unsigned char* ignore,
                                      const float* thresholds, int T, int se_bits, int64_t* cm, int64_t* invalid,
                                      int N, int H, int W, sc_stream stream) {
  SC_REQUIRE(pred && target && thresholds && cm, "sc_threshold_confusion: null pointer");
  SC_REQUIRE(T >= 1 && T <= MAXT, "sc_threshold_confusion: T=%d thresholds (1..%d)", T, MAXT);
  SC_REQUIRE(N > 0 && H > 0 && W > 0 && N <= 65535, "sc_threshold_confusion: bad dims N=%d H=%d W=%d", N, H, W);
  SC_REQUIRE(se_bits >= 0 && se_bits < 512, "sc_threshold_confusion: se_bits=%d is not a 3x3 structuring element", se_bits);
  ThrConfP p{};
  p.pred = pred; p.target = target; p.ignore = ignore; p.T = T; p.se = se_bits; p.N = N; p.H = H; p.W = W;
  for (int t = 0; t < T; ++t) p.thr[t] = thresholds[t];
  p.cm = reinterpret_cast<long long*>(cm); p.invalid = reinterpret_cast<long long*>(invalid);
  return launch(p, (hipStream_t)stream);
}

// ---- training-batch assembly: window crop + rotation + flips of HBM-resident tiles in one gather ----
namespace {

struct GatherP {
  const float* tiles;
  const int* tile; const int* row_off; const int* col_off;
  const float* cs; const float* sn; const int* flags;
  int M, C, Hs, Ws, B, h, w, mode;
  float* out;
};

__global__ __launch_bounds__(256) void k_gather_augment(const GatherP p) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.h * p.w) return;
  int y = i / p.w, x = i - y * p.w;
  const int fl = p.flags[b];
  const int t = p.tile[b], r0 = p.row_off[b], c0 = p.col_off[b];
  const float* src = p.tiles + ((size_t)t * p.C + c) * ((size_t)p.Hs * p.Ws);
  float* dst = p.out + ((size_t)b * p.C + c) * ((size_t)p.h * p.w) + i;
  // the flips were applied last: undo them first
  if (fl & 4) y = p.h - 1 - y;
  if (fl & 2) x = p.w - 1 - x;
  auto at = [&](int yy, int xx) -> float {      // zeros outside the crop window (not outside the stored tile)
    return (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) ? src[(size_t)(r0 + yy) * p.Ws + (c0 + xx)] : 0.f;
  };
  if (!(fl & 1)) { *dst = at(y, x); return; }
  const float cx = 0.5f * (p.w - 1), cy = 0.5f * (p.h - 1);
  const float cs = p.cs[b], sn = p.sn[b];
  const float dx = (float)x - cx, dy = (float)y - cy;
  const float xs = cx + cs * dx - sn * dy, ys = cy + sn * dx + cs * dy;
  if (p.mode == 1) {
    *dst = at((int)nearbyintf(ys), (int)nearbyintf(xs));
    return;
  }
  const float xf = floorf(xs), yf = floorf(ys);
  const int x0 = (int)xf, y0 = (int)yf;
  const float ax = xs - xf, ay = ys - yf;
  const float v = at(y0, x0) * (1.f - ax) * (1.f - ay) + at(y0, x0 + 1) * ax * (1.f - ay) +
                  at(y0 + 1, x0) * (1.f - ax) * ay + at(y0 + 1, x0 + 1) * ax * ay;
  *dst = v;
}

}  // namespace

extern "C" int sc_gather_augment(const float* tiles, int M, int C, int Hs, int Ws, const int32_t* tile, const int32_t* row_off,
                                 const int32_t* col_off, const float* cos_t, const float* sin_t, const int32_t* flags, int B,
                                 int h, int w, int mode, float* out, sc_stream stream) {
  SC_REQUIRE(tiles && tile && row_off && col_off && cos_t && sin_t && flags && out, "sc_gather_augment: null pointer");
  SC_REQUIRE(M > 0 && C > 0 && C <= 65535 && B > 0 && B <= 65535, "sc_gather_augment: bad M=%d C=%d B=%d", M, C, B);
  SC_REQUIRE(h > 0 && w > 0 && h <= Hs && w <= Ws, "sc_gather_augment: window %dx%d does not fit the %dx%d tiles", h, w, Hs, Ws);
  SC_REQUIRE(mode == 0 || mode == 1, "sc_gather_augment: mode=%d (0 bilinear, 1 nearest)", mode);
  GatherP p{tiles, tile, row_off, col_off, cos_t, sin_t, flags, M, C, Hs, Ws, B, h, w, mode, out};
  dim3 grid((h * w + 255) / 256, C, B);
  hipLaunchKernelGGL(k_gather_augment, grid, dim3(256), 0, (hipStream_t)stream, p);
  SC_LAUNCH_OK("k_gather_augment");
  return SC_OK;
}
