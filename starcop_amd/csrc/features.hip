// On-the-fly band-ratio features and the EMIT->AVIRIS range rescale.
//   ratio_2c_match_c_from_sums_outlier / no_outliers   starcop/data/feature_extration.py:37-56
//   weight_mag1c                                         starcop/data/feature_extration.py:32-35
//   EMIT rescale                                         starcop/emit_tools/emit_dataset.py:62-106
// The 5/95-percentile trim needs order statistics of a whole tile: an exact radix select on the float bit
// patterns (4 passes x 8 bits, LDS histograms), one work-group per (tile, rank); numpy's linear interpolation
// between the two neighbouring order statistics is reproduced in fp64.
#include "sc_common.h"

namespace {

__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);      // monotone: smaller float -> smaller key
}
__device__ __forceinline__ float key2f(unsigned k) {
  const unsigned b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(b);
}

constexpr int RS_THREADS = 1024;

// out[tile][r] = rank-th smallest element (0-based) of x[tile][0..n)
__global__ __launch_bounds__(RS_THREADS) void k_order_stat(const float* __restrict__ x, size_t n, const long long* __restrict__ ranks,
                                                           int nranks, float* __restrict__ out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_k;
  const int tile = blockIdx.y, r = blockIdx.x;
  const float* xt = x + (size_t)tile * n;
  unsigned prefix = 0, mask = 0;
  unsigned long long k = (unsigned long long)ranks[r];
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = threadIdx.x; i < 256; i += RS_THREADS) hist[i] = 0;
    __syncthreads();
    for (size_t i = threadIdx.x; i < n; i += RS_THREADS) {
      const unsigned key = f2key(xt[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long cum = 0;
      int b = 0;
      for (; b < 256; ++b) {
        if (cum + hist[b] > k) break;
        cum += hist[b];
      }
      s_prefix = prefix | ((unsigned)b << shift);
      s_k = (unsigned)(k - cum);
    }
    __syncthreads();
    prefix = s_prefix; k = s_k;
    mask |= 255u << shift;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(size_t)tile * nranks + r] = key2f(prefix);
}

// ---- large tiles (a whole scene as one tile: 1.6 M pixels of an EMIT granule): the single-work-group select above walks the tile four
// times with 1024 threads (3 ms per plane).  Here a pass is one launch of many work-groups: every block counts its chunk for all four
// ranks at once in LDS (the ranks have their own prefixes) and flushes to global histograms with integer atomics (exact, order-
// independent); a one-block kernel per pass picks the digit of every rank and clears the histograms.  8 launches, ~60 us per plane.
struct RsState { unsigned prefix[4]; unsigned mask; unsigned pad; unsigned long long k[4]; };

__global__ __launch_bounds__(256) void k_rs_hist(const float* __restrict__ x, size_t n, const RsState* __restrict__ st, int shift,
                                                 unsigned* __restrict__ hist /*[B][4][256]*/) {
  __shared__ unsigned h[4][256];
  const int tile = blockIdx.y;
  const RsState s = st[tile];
  for (int i = threadIdx.x; i < 1024; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const float* xt = x + (size_t)tile * n;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned key = f2key(xt[i]);
    const unsigned km = key & s.mask, bin = (key >> shift) & 255u;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (km == s.prefix[r]) atomicAdd(&h[r][bin], 1u);
  }
  __syncthreads();
  unsigned* g = hist + (size_t)tile * 1024;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const unsigned v = (&h[0][0])[i];
    if (v) atomicAdd(g + i, v);
  }
}
__global__ __launch_bounds__(256) void k_rs_pick(RsState* __restrict__ st, unsigned* __restrict__ hist, int shift, int last, float* __restrict__ out) {
  const int tile = blockIdx.x;
  unsigned* g = hist + (size_t)tile * 1024;
  if (threadIdx.x < 4) {
    const int r = threadIdx.x;
    unsigned long long k = st[tile].k[r], cum = 0;
    int b = 0;
    for (; b < 256; ++b) {
      const unsigned c = g[r * 256 + b];
      if (cum + c > k) break;
      cum += c;
    }
    st[tile].prefix[r] |= (unsigned)b << shift;
    st[tile].k[r] = k - cum;
    if (last) out[(size_t)tile * 4 + r] = key2f(st[tile].prefix[r]);
  }
  __syncthreads();
  if (threadIdx.x == 0) st[tile].mask |= 255u << shift;
  for (int i = threadIdx.x; i < 1024; i += 256) g[i] = 0;
}
__global__ void k_rs_init(RsState* __restrict__ st, unsigned* __restrict__ hist, const long long* __restrict__ ranks, int B) {
  const int tile = blockIdx.x;
  if (threadIdx.x < 4) { st[tile].prefix[threadIdx.x] = 0; st[tile].k[threadIdx.x] = (unsigned long long)ranks[threadIdx.x]; }
  if (threadIdx.x == 0) { st[tile].mask = 0; st[tile].pad = 0; }
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) hist[(size_t)tile * 1024 + i] = 0;
  (void)B;
}
constexpr size_t RS_LARGE_N = 1u << 17;

// bounds[tile] = {lower, upper} percentiles from the 4 order statistics {lo_k, lo_k+1, hi_k, hi_k+1} (numpy 'linear')
__global__ void k_percentile_bounds(const float* __restrict__ os, double t_lo, double t_hi, float* __restrict__ bounds, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double a0 = os[b * 4 + 0], b0 = os[b * 4 + 1], a1 = os[b * 4 + 2], b1 = os[b * 4 + 3];
  const double lo = t_lo >= 0.5 ? b0 - (b0 - a0) * (1.0 - t_lo) : a0 + (b0 - a0) * t_lo;
  const double hi = t_hi >= 0.5 ? b1 - (b1 - a1) * (1.0 - t_hi) : a1 + (b1 - a1) * t_hi;
  bounds[b * 2] = (float)lo; bounds[b * 2 + 1] = (float)hi;
}

__global__ __launch_bounds__(256) void k_trimmed_sum(const float* __restrict__ x, size_t n, const float* __restrict__ bounds,
                                                     double* __restrict__ sums) {
  __shared__ double s_tmp[4];
  const int tile = blockIdx.y;
  const float lo = bounds[tile * 2], hi = bounds[tile * 2 + 1];
  const float* xt = x + (size_t)tile * n;
  double a = 0.0;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = xt[i];
    if (v >= lo && v <= hi) a += (double)v;
  }
  a = wave_sum_d(a);
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[tile], s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3]);
}

__global__ __launch_bounds__(256) void k_band_ratio(const float* __restrict__ bg, const float* __restrict__ sig, float* __restrict__ out,
                                                    size_t n, const double* __restrict__ sum_bg, const double* __restrict__ sum_sig,
                                                    float c_host, float zero_val) {
  const int tile = blockIdx.y;
  const float c = sum_bg ? (float)sum_bg[tile] / (float)sum_sig[tile] : c_host;      // float32 division, as numpy does
  const size_t base = (size_t)tile * n;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float b = bg[base + i], s = sig[base + i];
    const float r = (c * s - b) / (b + 1e-6f);
    out[base + i] = (s < 1e-6f && b < 1e-6f) ? zero_val : r;
  }
}

__global__ __launch_bounds__(256) void k_clip_scale(const float* __restrict__ x, float* __restrict__ out, size_t n, float div, float lo,
                                                    float hi, float mult, int nan_to_num) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float v = x[i] / div;
    v = v < lo ? lo : (v > hi ? hi : v);          // np.clip: NaN stays NaN
    v *= mult;
    if (nan_to_num) {
      if (v != v) v = 0.f;
      else if (v == __builtin_inff()) v = 3.4028234663852886e38f;
      else if (v == -__builtin_inff()) v = -3.4028234663852886e38f;
    }
    out[i] = v;
  }
}

}  // namespace

extern "C" size_t sc_trimmed_sum_workspace_bytes(int B) {
  // order statistics [B][4] + bounds [B][2] floats, the four ranks, then (large tiles) the select state and histograms
  return (size_t)B * 6 * sizeof(float) + 64 + 64 + (size_t)B * (sizeof(RsState) + 1024 * sizeof(unsigned));
}

extern "C" int sc_trimmed_sums(const float* x, int B, size_t n, double p, double* sums, void* work, size_t work_bytes,
                               sc_stream stream) {
  SC_REQUIRE(x && sums && work && B > 0 && n > 1, "sc_trimmed_sums: bad argument");
  SC_REQUIRE(p >= 0.0 && p <= 50.0, "sc_trimmed_sums: percentile must be in [0, 50]");
  SC_REQUIRE(work_bytes >= sc_trimmed_sum_workspace_bytes(B), "sc_trimmed_sums: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // numpy.percentile(method='linear'): virtual index q/100*(n-1)
  const double pos_lo = p / 100.0 * (double)(n - 1), pos_hi = (100.0 - p) / 100.0 * (double)(n - 1);
  long long klo = (long long)pos_lo, khi = (long long)pos_hi;
  const double t_lo = pos_lo - (double)klo, t_hi = pos_hi - (double)khi;
  long long ranks_h[4] = {klo, klo + 1 < (long long)n ? klo + 1 : klo, khi, khi + 1 < (long long)n ? khi + 1 : khi};
  float* os = reinterpret_cast<float*>(work);                 // [B][4]
  float* bounds = os + (size_t)B * 4;                         // [B][2]
  long long* ranks_d = reinterpret_cast<long long*>(reinterpret_cast<char*>(work) + (((size_t)B * 6 * sizeof(float) + 7) & ~(size_t)7));
  if (hipMemcpyAsync(ranks_d, ranks_h, sizeof(ranks_h), hipMemcpyHostToDevice, st) != hipSuccess) {
    sc_set_error("sc_trimmed_sums: rank upload failed"); return SC_ERR_LAUNCH;
  }
  if (n >= RS_LARGE_N) {
    char* base = reinterpret_cast<char*>(ranks_d) + 64;
    RsState* rs = reinterpret_cast<RsState*>(base);
    unsigned* hist = reinterpret_cast<unsigned*>(base + (size_t)B * sizeof(RsState));
    hipLaunchKernelGGL(k_rs_init, dim3(B), dim3(256), 0, st, rs, hist, ranks_d, B);
    const unsigned gx = (unsigned)((n + 8191) / 8192 > 512 ? 512 : (n + 8191) / 8192);
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      hipLaunchKernelGGL(k_rs_hist, dim3(gx, B), dim3(256), 0, st, x, n, (const RsState*)rs, shift, hist);
      hipLaunchKernelGGL(k_rs_pick, dim3(B), dim3(256), 0, st, rs, hist, shift, pass == 3 ? 1 : 0, os);
    }
  } else {
    hipLaunchKernelGGL(k_order_stat, dim3(4, B), dim3(RS_THREADS), 0, st, x, n, ranks_d, 4, os);
  }
  SC_LAUNCH_OK("sc_trimmed_sums(order statistics)");
  hipLaunchKernelGGL(k_percentile_bounds, dim3((B + 63) / 64), dim3(64), 0, st, os, t_lo, t_hi, bounds, B);
  if (hipMemsetAsync(sums, 0, (size_t)B * sizeof(double), st) != hipSuccess) { sc_set_error("sc_trimmed_sums: memset failed"); return SC_ERR_LAUNCH; }
  const unsigned bx = (unsigned)((n + 4095) / 4096 > 64 ? 64 : (n + 4095) / 4096);
  hipLaunchKernelGGL(k_trimmed_sum, dim3(bx, B), dim3(256), 0, st, x, n, bounds, sums);
  SC_LAUNCH_OK("sc_trimmed_sums");
  return SC_OK;
}

extern "C" int sc_band_ratio(const float* background, const float* signal, float* out, int B, size_t n, const double* sum_bg,
                             const double* sum_sig, float c_host, float zero_value_out, sc_stream stream) {
  SC_REQUIRE(background && signal && out && B > 0 && n > 0, "sc_band_ratio: bad argument");
  SC_REQUIRE((sum_bg == nullptr) == (sum_sig == nullptr), "sc_band_ratio: give both trimmed sums or neither");
  const unsigned bx = (unsigned)((n + 1023) / 1024 > 256 ? 256 : (n + 1023) / 1024);
  hipLaunchKernelGGL(k_band_ratio, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, background, signal, out, n, sum_bg, sum_sig,
                     c_host, zero_value_out);
  SC_LAUNCH_OK("sc_band_ratio");
  return SC_OK;
}

extern "C" int sc_clip_scale(const float* x, float* out, size_t n, float div, float lo, float hi, float mult, int nan_to_num,
                             sc_stream stream) {
  SC_REQUIRE(x && out && n > 0, "sc_clip_scale: bad argument");
  const unsigned bx = (unsigned)((n + 1023) / 1024 > 2048 ? 2048 : (n + 1023) / 1024);
  hipLaunchKernelGGL(k_clip_scale, dim3(bx), dim3(256), 0, (hipStream_t)stream, x, out, n, div, lo, hi, mult, nan_to_num);
  SC_LAUNCH_OK("sc_clip_scale");
  return SC_OK;
}
