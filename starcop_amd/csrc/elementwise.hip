// HBM-bound streaming kernels of the U-Net training/inference step: BatchNorm bookkeeping,
// residual add, upsample backward, partial-sum reduction, loss, Adam, masks, band ratio.
// Reference lines are cited at each entry point in include/starcop_hip.h.
#include <stdarg.h>
#include <string.h>
#include "sc_common.h"
#include <atomic>
#include <mutex>

// ------------------------------------------------------------------------------------------
// error plumbing (thread-local message)
static thread_local char g_err[512] = "";
void sc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* sc_last_error(void) { return g_err; }
extern "C" int sc_version(void) { return 100; }
extern "C" int sc_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { sc_set_error("sc_device_check: no HIP device"); return SC_ERR_NODEV; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { sc_set_error("sc_device_check: cannot query device"); return SC_ERR_NODEV; }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    sc_set_error("sc_device_check: device is %s, this library is built for gfx950 only", prop.gcnArchName);
    return SC_ERR_NODEV;
  }
  return SC_OK;
}

namespace {

template <int NV>
__device__ __forceinline__ void block_sum_d(double (&v)[NV], double* s_tmp /* [4][NV] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double s = wave_sum_d(v[k]);
    if (lane == 0) s_tmp[wave * NV + k] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = s_tmp[k] + s_tmp[NV + k] + s_tmp[2 * NV + k] + s_tmp[3 * NV + k];
}

// ---------------------------------------------------------------- BatchNorm
// one block per channel: fp64 sum of the per-work-group partial rows, then the per-channel constants.  NW waves per block: a
// channel's rows are C*8 bytes apart (one 64-byte sector per 8 useful bytes), so the sum is a latency chain of nrows / (64 NW x 8)
// round trips -- 16 waves for the full-resolution layers (32768 rows: 38 -> ~12 us), 4 otherwise
// Pre-reduction for the many-row layers (full resolution: 8192-32768 rows): the rows of ALL channels are read as one contiguous
// stream (consecutive threads = consecutive floats of a row: coalesced, every byte used -- the per-channel walk above touches a
// 64-byte sector per 8 useful bytes, and C blocks each pull the whole buffer), BN_PRE_S work-groups x 1024 threads, fp64 sums, a
// fixed-order LDS reduction -> part[BN_PRE_S][C][2] doubles for k_bn_finalize<4, double>.
constexpr int BN_PRE_S = 64;
__global__ __launch_bounds__(1024) void k_bn_rows_prereduce(const float* __restrict__ stats, int nrows, int C, double* __restrict__ part) {
  __shared__ double s_d[1024];
  const int E = 2 * C;                       // floats per row
  const int rpp = 1024 / E;                  // rows per pass of the block
  const int e = threadIdx.x % E, rs = threadIdx.x / E;
  const int per = (nrows + BN_PRE_S - 1) / BN_PRE_S;
  const int r0 = blockIdx.x * per, r1 = min(r0 + per, nrows);
  double v = 0.0;
  if (rs < rpp) {
    int r = r0 + rs;
    for (; r + 3 * rpp < r1; r += 4 * rpp) {
      const float a = stats[(size_t)r * E + e], b = stats[(size_t)(r + rpp) * E + e], c = stats[(size_t)(r + 2 * rpp) * E + e],
                  d = stats[(size_t)(r + 3 * rpp) * E + e];
      v += (double)a; v += (double)b; v += (double)c; v += (double)d;
    }
    for (; r < r1; r += rpp) v += (double)stats[(size_t)r * E + e];
  }
  s_d[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x < E) {
    double t = 0.0;
    for (int k = 0; k < rpp; ++k) t += s_d[k * E + threadIdx.x];
    part[(size_t)blockIdx.x * E + threadIdx.x] = t;
  }
}

template <int NW, typename TS = float>
__global__ __launch_bounds__(64 * NW) void k_bn_finalize(const TS* __restrict__ stats, int nrows, double count,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* running_mean, float* running_var, float momentum, float eps,
                                                         int training, float* __restrict__ cst, int C, float* act_bound) {
  constexpr int NT = 64 * NW;
  __shared__ double s_tmp[2 * NW];
  const int c = blockIdx.x;
  double mean, var;
  if (training) {
    double v[2] = {0.0, 0.0};
    // eight independent loads in flight per thread
    int r = threadIdx.x;
    for (; r + 7 * NT < nrows; r += 8 * NT) {
      TS t[8][2];
#pragma unroll
      for (int k = 0; k < 8; ++k) { t[k][0] = stats[((size_t)(r + NT * k) * C + c) * 2]; t[k][1] = stats[((size_t)(r + NT * k) * C + c) * 2 + 1]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[0] += (double)t[k][0]; v[1] += (double)t[k][1]; }
    }
    for (; r < nrows; r += NT) {
      v[0] += (double)stats[((size_t)r * C + c) * 2]; v[1] += (double)stats[((size_t)r * C + c) * 2 + 1];
    }
    {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double sw = wave_sum_d(v[k]);
        if (lane == 0) s_tmp[wave * 2 + k] = sw;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += s_tmp[w * 2 + k];       // fixed order: reproducible
        v[k] = t;
      }
    }
    mean = v[0] / count;
    var = v[1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
  } else {
    mean = (double)running_mean[c];
    var = (double)running_var[c];
  }
  if (threadIdx.x != 0) return;
  bn_write_channel(mean, var, count, training, gamma, beta, running_mean, running_var, momentum, eps, cst, c, act_bound);
}

// max over the block of m (>= 0), then *slot = max(*slot, m * factor): the order-independent (hence reproducible) unsigned
// atomicMax on the bit pattern of a non-negative float
__device__ __forceinline__ void block_absmax_to(float m, float factor, float* slot, float* s_m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])) * fabsf(factor);
    // thousands of blocks share the slot: only those that would raise it pay for the atomic (a stale read just means
    // one atomic more; the result is the maximum either way)
    if (t > __builtin_nontemporal_load(slot)) atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, t));
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const float* __restrict__ g, const float* __restrict__ y,
                                                       const float* __restrict__ cst, int act, double* __restrict__ sums, int C, int HW,
                                                       float* __restrict__ absmax, float* __restrict__ act_absmax) {
  __shared__ double s_tmp[8];
  __shared__ float s_m[4];
  const int c = blockIdx.y, n = blockIdx.z;
  const float scale = cst[(size_t)c * SC_CST], shift = cst[(size_t)c * SC_CST + 1];
  const float mean = cst[(size_t)c * SC_CST + 2], invstd = cst[(size_t)c * SC_CST + 3];
  const float lo = sc_act_lo(act), hi = sc_act_hi(act);
  const size_t base = ((size_t)n * C + c) * HW;
  const int start = blockIdx.x * 4096;
  const int end = min(start + 4096, HW);
  // ax: largest |BatchNorm output| seen (the activation clamps only towards zero, so this bounds |act(BN(y))|): the device-side
  // range record of the tensors that feed the two-fp16-term kernels -- one v_max per element in a pass that reads y anyway
  float s1 = 0.f, s2 = 0.f, mx = 0.f, ax = 0.f;
  if ((HW & 3) == 0) {
    for (int i = start + threadIdx.x * 4; i < end; i += 1024) {
      const float4 yv = *reinterpret_cast<const float4*>(y + base + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + base + i);
      const float ya[4] = {yv.x, yv.y, yv.z, yv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float yh = fmaf(ya[k], scale, shift);
        const float gb = (yh > lo && yh < hi) ? ga[k] : 0.f;
        s1 += gb;
        s2 = fmaf(gb, (ya[k] - mean) * invstd, s2);
        mx = fmaxf(mx, fabsf(gb));
        ax = fmaxf(ax, fabsf(yh));
      }
    }
  } else {
    for (int i = start + threadIdx.x; i < end; i += 256) {
      const float yv = y[base + i], gv = g[base + i];
      const float yh = fmaf(yv, scale, shift);
      const float gb = (yh > lo && yh < hi) ? gv : 0.f;
      s1 += gb;
      s2 = fmaf(gb, (yv - mean) * invstd, s2);
      mx = fmaxf(mx, fabsf(gb));
      ax = fmaxf(ax, fabsf(yh));
    }
  }
  double v[2] = {(double)s1, (double)s2};
  block_sum_d<2>(v, s_tmp);
  if (threadIdx.x < 2) sums[(stat_row() * C + c) * 2 + threadIdx.x] = v[threadIdx.x];
  if (absmax) block_absmax_to(mx, scale, absmax, s_m);
  if (act_absmax) {
    __syncthreads();
    block_absmax_to(act == SC_ACT_RELU6 ? fminf(ax, 6.f) : ax, 1.f, act_absmax, s_m);
  }
}

// Low-resolution layers: one block per channel walks all N*HW elements of its channel and finalises in the same launch
// (no partial rows, no second kernel: the two-kernel form costs two ~7 us dependent launches per BatchNorm and, at 16x16 or
// 32x32, 10^4 work-groups of a few hundred elements each).
__global__ __launch_bounds__(256) void k_bn_bwd_small(const float* __restrict__ g, const float* __restrict__ y,
                                                      const float* __restrict__ cst, int act, int N, int C, int HW, double count,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ cst_bwd,
                                                      float* __restrict__ absmax, float* __restrict__ act_absmax) {
  __shared__ double s_tmp[8];
  __shared__ float s_m[4];
  float mx = 0.f, ax = 0.f;
  const int c = blockIdx.x;
  const float scale = cst[(size_t)c * SC_CST], shift = cst[(size_t)c * SC_CST + 1];
  const float mean = cst[(size_t)c * SC_CST + 2], invstd = cst[(size_t)c * SC_CST + 3];
  const float lo = sc_act_lo(act), hi = sc_act_hi(act);
  double v[2] = {0.0, 0.0};
  auto acc4 = [&](const float4& yv, const float4& gv, float& s1, float& s2) {
    const float ya[4] = {yv.x, yv.y, yv.z, yv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float yh = fmaf(ya[k], scale, shift);
      const float gb = (yh > lo && yh < hi) ? ga[k] : 0.f;
      s1 += gb;
      s2 = fmaf(gb, (ya[k] - mean) * invstd, s2);
      mx = fmaxf(mx, fabsf(gb));
      ax = fmaxf(ax, fabsf(yh));
    }
  };
  if ((HW & 3) == 0) {
    // the channel's N planes as ONE list of float4 items, four items (eight 16-byte loads) in flight per thread: with one plane
    // per loop iteration a 32x32 layer was 16 dependent memory round trips (11 us per launch, 25 launches per step)
    const int q = HW >> 2, total = N * q;
    int it = threadIdx.x;
    for (; it + 3 * 256 < total; it += 4 * 256) {
      float4 yv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int item = it + 256 * u, n = item / q, i = (item - n * q) * 4;
        const size_t o = ((size_t)n * C + c) * HW + i;
        yv[u] = *reinterpret_cast<const float4*>(y + o);
        gv[u] = *reinterpret_cast<const float4*>(g + o);
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) acc4(yv[u], gv[u], s1, s2);
      v[0] += (double)s1; v[1] += (double)s2;
    }
    for (; it < total; it += 256) {
      const int n = it / q, i = (it - n * q) * 4;
      const size_t o = ((size_t)n * C + c) * HW + i;
      float s1 = 0.f, s2 = 0.f;
      acc4(*reinterpret_cast<const float4*>(y + o), *reinterpret_cast<const float4*>(g + o), s1, s2);
      v[0] += (double)s1; v[1] += (double)s2;
    }
  } else {
    for (int n = 0; n < N; ++n) {
      const size_t base = ((size_t)n * C + c) * HW;
      float s1 = 0.f, s2 = 0.f;
      for (int i = threadIdx.x; i < HW; i += 256) {
        const float yv = y[base + i], gv = g[base + i];
        const float yh = fmaf(yv, scale, shift);
        const float gb = (yh > lo && yh < hi) ? gv : 0.f;
        s1 += gb;
        s2 = fmaf(gb, (yv - mean) * invstd, s2);
        mx = fmaxf(mx, fabsf(gb));
        ax = fmaxf(ax, fabsf(yh));
      }
      v[0] += (double)s1; v[1] += (double)s2;
    }
  }
  block_sum_d<2>(v, s_tmp);
  if (absmax) block_absmax_to(mx, scale, absmax, s_m);
  if (act_absmax) {
    __syncthreads();
    block_absmax_to(act == SC_ACT_RELU6 ? fminf(ax, 6.f) : ax, 1.f, act_absmax, s_m);
  }
  if (threadIdx.x != 0) return;
  const double t1 = v[0], t2 = v[1];
  if (dbeta) dbeta[c] = (float)t1;
  if (dgamma) dgamma[c] = (float)t2;
  const double c1 = t1 / count, c2 = t2 / count;
  float* o = cst_bwd + (size_t)c * SC_CST;
  o[0] = scale; o[1] = shift;
  o[2] = scale;
  o[3] = (float)(-(double)scale * c2 * (double)invstd);
  o[4] = (float)(-(double)scale * c1 + (double)scale * c2 * (double)invstd * (double)mean);
  o[5] = 0.f; o[6] = 0.f; o[7] = 0.f;
}

template <typename RT>
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const RT* __restrict__ sums, int nrows, double count,
                                                         const float* __restrict__ cst_fwd, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, float* __restrict__ cst_bwd, int C) {
  __shared__ double s_tmp[8];
  const int c = blockIdx.x;
  double v[2] = {0.0, 0.0};
  for (int r = threadIdx.x; r < nrows; r += 256) {
    v[0] += sums[((size_t)r * C + c) * 2];
    v[1] += sums[((size_t)r * C + c) * 2 + 1];
  }
  block_sum_d<2>(v, s_tmp);
  if (threadIdx.x != 0) return;
  const double s1 = v[0], s2 = v[1];
  const double scale = cst_fwd[(size_t)c * SC_CST], mean = cst_fwd[(size_t)c * SC_CST + 2], invstd = cst_fwd[(size_t)c * SC_CST + 3];
  if (dbeta) dbeta[c] = (float)s1;
  if (dgamma) dgamma[c] = (float)s2;
  const double c1 = s1 / count, c2 = s2 / count;
  float* o = cst_bwd + (size_t)c * SC_CST;
  o[0] = cst_fwd[(size_t)c * SC_CST];
  o[1] = cst_fwd[(size_t)c * SC_CST + 1];
  o[2] = (float)scale;                                         // A
  o[3] = (float)(-scale * c2 * invstd);                        // B
  o[4] = (float)(-scale * c1 + scale * c2 * invstd * mean);    // D
  o[5] = 0.f; o[6] = 0.f; o[7] = 0.f;
}

// ---------------------------------------------------------------- elementwise over [N][C][HW]
__device__ __forceinline__ float ld_src(const SrcD& s, size_t idx, float4 c0, float c4) {
  const float x = s.x[idx];
  if (s.mode == SC_SRC_RAW) return x;
  const float au = (s.mode == SC_SRC_BNBWD) ? s.aux[idx] : 0.f;
  return sc_prologue(s.mode, s.act, x, au, c0, c4);
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_add_srcs(const SrcD a, const SrcD b, int has_b, float* __restrict__ out, int C, int HW,
                                                  float* __restrict__ absmax) {
  __shared__ float s_m[4];
  float mx = 0.f;
  const int c = blockIdx.y, n = blockIdx.z;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), b0 = a0; float a4 = 0.f, b4 = 0.f;
  if (a.mode != SC_SRC_RAW) { a0 = *reinterpret_cast<const float4*>(a.cst + (size_t)c * SC_CST); a4 = a.cst[(size_t)c * SC_CST + 4]; }
  if (has_b && b.mode != SC_SRC_RAW) { b0 = *reinterpret_cast<const float4*>(b.cst + (size_t)c * SC_CST); b4 = b.cst[(size_t)c * SC_CST + 4]; }
  const size_t base = ((size_t)n * C + c) * HW;
  const int start = blockIdx.x * 2048, end = min(start + 2048, HW);
  if (VEC) {
    // 16 bytes per lane, both 1024-element halves of the chunk requested before any arithmetic (HW % 4 == 0, 16-byte aligned tensors:
    // the host decides).  The residual add of features.3 at 16 x 128^2 moved 75 MB in 37 us with the 4-byte loop below.
    float4 xa[2], ya[2], xb[2], yb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = start + 4 * threadIdx.x + 1024 * u, ic = i < end ? i : start;
      xa[u] = *reinterpret_cast<const float4*>(a.x + base + ic);
      ya[u] = (a.mode == SC_SRC_BNBWD) ? *reinterpret_cast<const float4*>(a.aux + base + ic) : make_float4(0.f, 0.f, 0.f, 0.f);
      xb[u] = has_b ? *reinterpret_cast<const float4*>(b.x + base + ic) : make_float4(0.f, 0.f, 0.f, 0.f);
      yb[u] = (has_b && b.mode == SC_SRC_BNBWD) ? *reinterpret_cast<const float4*>(b.aux + base + ic) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = start + 4 * threadIdx.x + 1024 * u;
      if (i < end) {
        const float xv[4] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w}, yv[4] = {ya[u].x, ya[u].y, ya[u].z, ya[u].w};
        const float xw[4] = {xb[u].x, xb[u].y, xb[u].z, xb[u].w}, yw[4] = {yb[u].x, yb[u].y, yb[u].z, yb[u].w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v = a.mode == SC_SRC_RAW ? xv[k] : sc_prologue(a.mode, a.act, xv[k], yv[k], a0, a4);
          if (has_b) v += b.mode == SC_SRC_RAW ? xw[k] : sc_prologue(b.mode, b.act, xw[k], yw[k], b0, b4);
          o[k] = v;
          mx = fmaxf(mx, fabsf(v));
        }
        if (out) *reinterpret_cast<float4*>(out + base + i) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  } else {
    for (int i = start + threadIdx.x; i < end; i += 256) {
      float v = ld_src(a, base + i, a0, a4);
      if (has_b) v += ld_src(b, base + i, b0, b4);
      if (out) out[base + i] = v;
      mx = fmaxf(mx, fabsf(v));
    }
  }
  if (absmax) block_absmax_to(mx, 1.f, absmax, s_m);
}

// nn.MaxPool2d(2) of the in-repo UNet (architectures/unet.py:15): out[n,c,y,x] = max of the 2x2 block of v(in) (v = the source's prologue)
__global__ __launch_bounds__(256) void k_maxpool2x2(const SrcD a, float* __restrict__ out, int C, int Hout, int Wout) {
  const int c = blockIdx.y, n = blockIdx.z;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f); float a4 = 0.f;
  if (a.mode != SC_SRC_RAW) { a0 = *reinterpret_cast<const float4*>(a.cst + (size_t)c * SC_CST); a4 = a.cst[(size_t)c * SC_CST + 4]; }
  const size_t ibase = ((size_t)n * C + c) * (size_t)(4 * Hout * Wout), obase = ((size_t)n * C + c) * (size_t)(Hout * Wout);
  const int Win = 2 * Wout;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Hout * Wout; i += gridDim.x * 256) {
    const int y = i / Wout, x = i - y * Wout;
    const size_t p = ibase + (size_t)(2 * y) * Win + 2 * x;
    const float v = fmaxf(fmaxf(ld_src(a, p, a0, a4), ld_src(a, p + 1, a0, a4)), fmaxf(ld_src(a, p + Win, a0, a4), ld_src(a, p + Win + 1, a0, a4)));
    out[obase + i] = v;
  }
}

// F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) (architectures/unet.py:35,39,43): output pixel (Y, X) of a
// (2H, 2W) grid samples the input at Y*(H-1)/(2H-1), X*(W-1)/(2W-1); weights and the two-step lerp as torch's CPU kernel
// (area_pixel_compute_source_index + compute_scales_value: scale = (in-1)/(out-1) in float)
__global__ __launch_bounds__(256) void k_upsample_bilinear2x(const SrcD a, float* __restrict__ out, int C, int Hin, int Win) {
  const int c = blockIdx.y, n = blockIdx.z;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f); float a4 = 0.f;
  if (a.mode != SC_SRC_RAW) { a0 = *reinterpret_cast<const float4*>(a.cst + (size_t)c * SC_CST); a4 = a.cst[(size_t)c * SC_CST + 4]; }
  const int Ho = 2 * Hin, Wo = 2 * Win;
  const float sh = Ho > 1 ? (float)(Hin - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Win - 1) / (float)(Wo - 1) : 0.f;
  const size_t ibase = ((size_t)n * C + c) * (size_t)(Hin * Win), obase = ((size_t)n * C + c) * (size_t)(Ho * Wo);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Ho * Wo; i += gridDim.x * 256) {
    const int Y = i / Wo, X = i - Y * Wo;
    const float fy = sh * (float)Y, fx = sw * (float)X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float v00 = ld_src(a, ibase + (size_t)y0 * Win + x0, a0, a4), v01 = ld_src(a, ibase + (size_t)y0 * Win + x1, a0, a4);
    const float v10 = ld_src(a, ibase + (size_t)y1 * Win + x0, a0, a4), v11 = ld_src(a, ibase + (size_t)y1 * Win + x1, a0, a4);
    out[obase + i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
  }
}

// backward of nn.MaxPool2d(2): the gradient of a pooled pixel goes to the FIRST maximum of its 2x2 window in row-major order (what
// torch's max_pool2d_with_indices records); the window values are read through the source's prologue, like the forward
__global__ __launch_bounds__(256) void k_maxpool2x2_bwd(const SrcD a, const float* __restrict__ gp, float* __restrict__ gin, int accum,
                                                        int C, int Hout, int Wout) {
  const int c = blockIdx.y, n = blockIdx.z;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f); float a4 = 0.f;
  if (a.mode != SC_SRC_RAW) { a0 = *reinterpret_cast<const float4*>(a.cst + (size_t)c * SC_CST); a4 = a.cst[(size_t)c * SC_CST + 4]; }
  const size_t ibase = ((size_t)n * C + c) * (size_t)(4 * Hout * Wout), obase = ((size_t)n * C + c) * (size_t)(Hout * Wout);
  const int Win = 2 * Wout;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Hout * Wout; i += gridDim.x * 256) {
    const int y = i / Wout, x = i - y * Wout;
    const size_t p = ibase + (size_t)(2 * y) * Win + 2 * x;
    const float v[4] = {ld_src(a, p, a0, a4), ld_src(a, p + 1, a0, a4), ld_src(a, p + Win, a0, a4), ld_src(a, p + Win + 1, a0, a4)};
    int best = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if (v[k] > v[best]) best = k;
    const float g = gp[obase + i];
    const size_t q[4] = {p, p + 1, p + Win, p + Win + 1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float t = (k == best) ? g : 0.f;
      gin[q[k]] = accum ? gin[q[k]] + t : t;
    }
  }
}

// backward of the bilinear x2 upsampling (align_corners=True) as a GATHER (deterministic, no atomics): input pixel (y, x) collects
// every output pixel whose two source rows / columns include it, with exactly the forward kernel's weights
__device__ __forceinline__ void bil_src(int Y, int Hin, float sh, int& y0, int& y1, float& ly) {
  const float fy = sh * (float)Y;
  y0 = (int)fy; y1 = y0 + (y0 < Hin - 1 ? 1 : 0); ly = fy - (float)y0;
}
__global__ __launch_bounds__(256) void k_upsample_bilinear2x_bwd(const float* __restrict__ gout, float* __restrict__ gin, int C, int Hin, int Win) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int Ho = 2 * Hin, Wo = 2 * Win;
  const float sh = Ho > 1 ? (float)(Hin - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Win - 1) / (float)(Wo - 1) : 0.f;
  const size_t ibase = ((size_t)n * C + c) * (size_t)(Hin * Win), obase = ((size_t)n * C + c) * (size_t)(Ho * Wo);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Hin * Win; i += gridDim.x * 256) {
    const int y = i / Win, x = i - y * Win;
    // output rows whose source position lies in (y - 1, y + 1): Y in [(y - 1) / sh, (y + 1) / sh]; two spare rows on either side
    const int Ylo = sh > 0.f ? max(0, (int)((float)(y - 1) / sh) - 1) : 0, Yhi = sh > 0.f ? min(Ho - 1, (int)((float)(y + 1) / sh) + 2) : Ho - 1;
    const int Xlo = sw > 0.f ? max(0, (int)((float)(x - 1) / sw) - 1) : 0, Xhi = sw > 0.f ? min(Wo - 1, (int)((float)(x + 1) / sw) + 2) : Wo - 1;
    float acc = 0.f;
    for (int Y = Ylo; Y <= Yhi; ++Y) {
      int y0, y1; float ly;
      bil_src(Y, Hin, sh, y0, y1, ly);
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int X = Xlo; X <= Xhi; ++X) {
        int x0, x1; float lx;
        bil_src(X, Win, sw, x0, x1, lx);
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        if (wx != 0.f) row = fmaf(wx, gout[obase + (size_t)Y * Wo + X], row);
      }
      acc = fmaf(wy, row, acc);
    }
    gin[ibase + i] = acc;
  }
}

__global__ __launch_bounds__(256) void k_downsum2x2(const float* __restrict__ in, float* __restrict__ out, int accum,
                                                    int Hout, int Wout, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wout);
    size_t r = i / Wout;
    const int y = (int)(r % Hout);
    const size_t plane = r / Hout;
    const float* p = in + (plane * (2 * Hout) + 2 * y) * (size_t)(2 * Wout) + 2 * x;
    const float2 t0 = *reinterpret_cast<const float2*>(p);
    const float2 t1 = *reinterpret_cast<const float2*>(p + 2 * Wout);
    const float s = (t0.x + t0.y) + (t1.x + t1.y);
    out[i] = accum ? out[i] + s : s;
  }
}

__global__ void k_fill_f64(double* p, double v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// out[r][i] = sum_{k<16} part[16r+k][i]
__global__ __launch_bounds__(256) void k_reduce16(const float* __restrict__ part, int nparts, size_t E, float* __restrict__ out) {
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i >= E) return;
  const int r = blockIdx.y;
  const int k0 = r * 16, k1 = min(k0 + 16, nparts);
  float s = 0.f;
#pragma unroll 16
  for (int k = k0; k < k1; ++k) s += part[(size_t)k * E + i];
  out[(size_t)r * E + i] = s;
}

// ---------------------------------------------------------------- loss / optimiser / masks
__global__ __launch_bounds__(256) void k_bce(const float* __restrict__ z, const float* __restrict__ t, const float* __restrict__ w,
                                             float pos_weight, size_t n, float inv_n, double* loss_sum, float* __restrict__ dz,
                                             float* __restrict__ loss_px) {
  __shared__ double s_tmp[4];
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = z[i], y = t[i];
    const float wt = w ? w[i] : 1.f;
    const float lw = fmaf(pos_weight - 1.f, y, 1.f);
    const float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f);      // softplus(-x)
    const float l = (1.f - y) * x + lw * sp;
    if (loss_px) loss_px[i] = l;
    acc += (double)(l * wt);
    if (dz) {
      const float sg = 1.f / (1.f + expf(-x));
      dz[i] = wt * (lw * sg - pos_weight * y) * inv_n;
    }
  }
  double v[1] = {acc};
  block_sum_d<1>(v, s_tmp);
  if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum, v[0]);
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                              float wd, float bc1, float bc2s, float gscale, const float* __restrict__ hp) {
  if (hp) { lr = hp[0]; bc1 = hp[1]; bc2s = hp[2]; }
  const float step = lr / bc1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    // torch: exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = pi - step * (mi / denom);
  }
}

__global__ void k_adam_prepare(long long* step, const float* lr_dev, float b1, float b2, float* hp) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long t = step[0] + 1;
    step[0] = t;
    hp[0] = lr_dev[0];
    hp[1] = (float)(1.0 - pow((double)b1, (double)t));
    hp[2] = (float)sqrt(1.0 - pow((double)b2, (double)t));
  }
}

__global__ __launch_bounds__(256) void k_threshold(const float* __restrict__ z, const float* __restrict__ tgt, int ge0,
                                                   float* __restrict__ pred, long long* __restrict__ pb,
                                                   long long* __restrict__ diff, unsigned long long* tile_count, int HW) {
  __shared__ double s_tmp[4];
  const int n = blockIdx.y;
  const size_t base = (size_t)n * HW;
  const int start = blockIdx.x * 4096, end = min(start + 4096, HW);
  int cnt = 0;
  for (int i = start + threadIdx.x; i < end; i += 256) {
    const float x = z[base + i];
    const float sg = 1.f / (1.f + expf(-x));
    const int b = ge0 ? (x >= 0.f) : (sg > 0.5f);
    if (pred) pred[base + i] = sg;
    if (pb) pb[base + i] = b;
    if (diff) diff[base + i] = 2 * b + ((long long)tgt[base + i] == 1 ? 1 : 0);
    cnt += b;
  }
  double v[1] = {(double)cnt};
  block_sum_d<1>(v, s_tmp);
  if (threadIdx.x == 0 && tile_count) atomicAdd(&tile_count[n], (unsigned long long)(v[0] + 0.5));
}

__global__ void k_pred_cls(const long long* __restrict__ cnt, long long* __restrict__ cls, int N, double thr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) cls[i] = ((double)cnt[i] > thr) ? 1 : 0;
}

inline int nblocks(size_t n, int cap = 2048) {
  size_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

// ------------------------------------------------------------------------------------------
size_t sc_reduce_scratch_floats(int nparts, size_t E) {
  size_t tot = 0;
  int np = nparts;
  while (np > 1) { np = (np + 15) / 16; tot += (size_t)np * E; }
  return tot;
}

int sc_reduce_rows_partial(const float* part, int nparts, size_t E, float* scratch, const float** rows_out, int* nrows_out,
                           int max_rows, hipStream_t st) {
  const float* cur = part;
  int np = nparts;
  while (np > max_rows) {    // the caller's final kernel sums up to max_rows rows itself
    const int nn = (np + 15) / 16;
    hipLaunchKernelGGL(k_reduce16, dim3((unsigned)((E + 255) / 256), nn), dim3(256), 0, st, cur, np, E, scratch);
    SC_LAUNCH_OK("sc_reduce16");
    cur = scratch; scratch += (size_t)nn * E; np = nn;
  }
  *rows_out = cur; *nrows_out = np;
  return SC_OK;
}

int sc_reduce_rows(const float* part, int nparts, size_t E, float* scratch, float* out, hipStream_t st) {
  const float* cur; int np;
  int rc = sc_reduce_rows_partial(part, nparts, E, scratch, &cur, &np, 16, st);
  if (rc != SC_OK) return rc;
  hipLaunchKernelGGL(k_reduce16, dim3((unsigned)((E + 255) / 256), 1), dim3(256), 0, st, cur, np, E, out);
  SC_LAUNCH_OK("sc_reduce16(final)");
  return SC_OK;
}

extern "C" int sc_stat_rows(int kind, int N, int H, int W) {
  switch (kind) {
    case SC_STAT_CONV3: return N * ((W + 31) / 32) * ((H + 3) / 4);
    case SC_STAT_CONV1: return N * ((H * W + 127) / 128);
    case SC_STAT_CONV1K: return N * ((H * W + 31) / 32);
    case SC_STAT_DW: return N * ((W + sc_dw_tile_w(W) - 1) / sc_dw_tile_w(W)) * ((H + sc_dw_tile_h(W) - 1) / sc_dw_tile_h(W));
    case SC_STAT_STEM: return N * ((W + 31) / 32) * ((H + 7) / 8);
    case SC_STAT_BNBWD: return N * ((H * W + 4095) / 4096);
    case SC_STAT_PW3: return (int)(((long)N * H * W + 31) / 32);
    default: return -1;
  }
}

extern "C" int sc_bn_finalize(const float* stats, int nrows, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, int training,
                              float* cst_fwd, int C, double* scratch, float* act_bound, sc_stream stream) {
  SC_REQUIRE(C > 0 && cst_fwd && gamma && beta && running_mean && running_var, "sc_bn_finalize: null argument");
  SC_REQUIRE(!training || (stats && count > 0 && nrows > 0), "sc_bn_finalize: training needs stats rows and count");
  if (training && nrows >= 4096 && scratch && 2 * C <= 1024) {
    hipLaunchKernelGGL(k_bn_rows_prereduce, dim3(BN_PRE_S), dim3(1024), 0, (hipStream_t)stream, stats, nrows, C, scratch);
    hipLaunchKernelGGL((k_bn_finalize<4, double>), dim3(C), dim3(256), 0, (hipStream_t)stream, (const double*)scratch, BN_PRE_S, count, gamma,
                       beta, running_mean, running_var, momentum, eps, training, cst_fwd, C, act_bound);
  } else if (training && nrows >= 4096)
    hipLaunchKernelGGL((k_bn_finalize<16>), dim3(C), dim3(1024), 0, (hipStream_t)stream, stats, nrows, count, gamma, beta,
                       running_mean, running_var, momentum, eps, training, cst_fwd, C, act_bound);
  else
    hipLaunchKernelGGL((k_bn_finalize<4>), dim3(C), dim3(256), 0, (hipStream_t)stream, stats, nrows, count, gamma, beta,
                       running_mean, running_var, momentum, eps, training, cst_fwd, C, act_bound);
  SC_LAUNCH_OK("sc_bn_finalize");
  return SC_OK;
}

extern "C" int sc_bn_bwd_reduce(const float* g, const float* y, const float* cst_fwd, int act, double* sums, int N, int C,
                                int HW, float* absmax, float* act_absmax, sc_stream stream) {
  SC_REQUIRE(g && y && cst_fwd && sums && N > 0 && C > 0 && HW > 0, "sc_bn_bwd_reduce: bad argument");
  dim3 grid((HW + 4095) / 4096, C, N);
  hipLaunchKernelGGL(k_bn_bwd_reduce, grid, dim3(256), 0, (hipStream_t)stream, g, y, cst_fwd, act, sums, C, HW, absmax, act_absmax);
  SC_LAUNCH_OK("sc_bn_bwd_reduce");
  return SC_OK;
}

extern "C" int sc_bn_bwd_small(const float* g, const float* y, const float* cst_fwd, int act, int N, int C, int HW,
                               float* dgamma, float* dbeta, float* cst_bwd, float* absmax, float* act_absmax, sc_stream stream) {
  SC_REQUIRE(g && y && cst_fwd && cst_bwd && N > 0 && C > 0 && HW > 0, "sc_bn_bwd_small: bad argument");
  hipLaunchKernelGGL(k_bn_bwd_small, dim3(C), dim3(256), 0, (hipStream_t)stream, g, y, cst_fwd, act, N, C, HW,
                     (double)N * HW, dgamma, dbeta, cst_bwd, absmax, act_absmax);
  SC_LAUNCH_OK("sc_bn_bwd_small");
  return SC_OK;
}

extern "C" int sc_bn_bwd_finalize(const double* sums, int nrows, double count, const float* cst_fwd, float* dgamma,
                                  float* dbeta, float* cst_bwd, int C, sc_stream stream) {
  SC_REQUIRE(sums && cst_fwd && cst_bwd && C > 0 && count > 0 && nrows > 0, "sc_bn_bwd_finalize: bad argument");
  hipLaunchKernelGGL(k_bn_bwd_finalize<double>, dim3(C), dim3(256), 0, (hipStream_t)stream, sums, nrows, count, cst_fwd,
                     dgamma, dbeta, cst_bwd, C);
  SC_LAUNCH_OK("sc_bn_bwd_finalize");
  return SC_OK;
}

extern "C" int sc_bn_bwd_finalize_rows32(const float* rows, int nrows, double count, const float* cst_fwd, float* dgamma,
                                         float* dbeta, float* cst_bwd, int C, double* scratch, sc_stream stream) {
  SC_REQUIRE(rows && cst_fwd && cst_bwd && C > 0 && count > 0 && nrows > 0, "sc_bn_bwd_finalize_rows32: bad argument");
  if (nrows >= 4096 && scratch && 2 * C <= 1024) {       // many rows: the coalesced pre-reduction of sc_bn_finalize first
    hipLaunchKernelGGL(k_bn_rows_prereduce, dim3(BN_PRE_S), dim3(1024), 0, (hipStream_t)stream, rows, nrows, C, scratch);
    hipLaunchKernelGGL(k_bn_bwd_finalize<double>, dim3(C), dim3(256), 0, (hipStream_t)stream, (const double*)scratch, BN_PRE_S, count, cst_fwd,
                       dgamma, dbeta, cst_bwd, C);
  } else
  hipLaunchKernelGGL(k_bn_bwd_finalize<float>, dim3(C), dim3(256), 0, (hipStream_t)stream, rows, nrows, count, cst_fwd,
                     dgamma, dbeta, cst_bwd, C);
  SC_LAUNCH_OK("sc_bn_bwd_finalize_rows32");
  return SC_OK;
}

// Cross-stream ordering on ONE device without the system-scope fence of a default event.  torch's Stream.wait_stream records a
// default-flag event: when it transitions to recorded the runtime performs a system-scope release -- a write-back / invalidate of the
// eight L2s -- and the next kernels of BOTH queues start 13-25 us later (tools/phase_gaps.py on a rocprofv3 trace: 0.3-0.4 ms of
// device-idle time per training step behind the ~50 fork points of the weight-gradient stream).  Kernels of the same device only need
// the device-scope release every dispatch ends with, so the events here are created with hipEventDisableSystemFence.
extern "C" int sc_stream_wait_stream(sc_stream waiter, sc_stream signaller) {
  // (an event may be re-recorded while an earlier wait on it is still pending: a wait captures the record that preceded it.  One ring
  //  per device, created on first use there: an event belongs to the device that was current when it was created)
  constexpr int NEV = 64, MAXDEV = 16;
  static hipEvent_t rings[MAXDEV][NEV];
  static std::once_flag once[MAXDEV];
  static bool oks[MAXDEV] = {};
  static std::atomic<unsigned> next{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) {
    sc_set_error("sc_stream_wait_stream: device index outside [0, %d)", MAXDEV);
    return SC_ERR_LAUNCH;
  }
  std::call_once(once[dev], [dev] {
    bool ok = true;
    for (int i = 0; i < NEV; ++i) ok = ok && hipEventCreateWithFlags(&rings[dev][i], hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
    oks[dev] = ok;
  });
  if (!oks[dev]) {
    sc_set_error("sc_stream_wait_stream: hipEventCreateWithFlags failed");
    return SC_ERR_LAUNCH;
  }
  hipEvent_t* const ring = rings[dev];
  hipEvent_t e = ring[next.fetch_add(1) % NEV];
  if (hipEventRecord(e, (hipStream_t)signaller) != hipSuccess || hipStreamWaitEvent((hipStream_t)waiter, e, 0) != hipSuccess) {
    sc_set_error("sc_stream_wait_stream: %s", hipGetErrorString(hipGetLastError()));
    return SC_ERR_LAUNCH;
  }
  return SC_OK;
}

extern "C" int sc_add_srcs_absmax(const sc_src* a, const sc_src* b, float* out, int N, int C, int HW, float* absmax,
                                  sc_stream stream) {
  SC_REQUIRE(a && a->C == C && (!b || b->C == C), "sc_add_srcs: channel mismatch");
  SC_REQUIRE(a->up == 0 && (!b || b->up == 0), "sc_add_srcs: upsampled sources unsupported");
  SC_REQUIRE(out || absmax, "sc_add_srcs: neither an output nor an absmax record");
  dim3 grid((HW + 2047) / 2048, C, N);
  uintptr_t al = (uintptr_t)a->x | (uintptr_t)out;
  if (a->mode == SC_SRC_BNBWD) al |= (uintptr_t)a->aux;
  if (b) { al |= (uintptr_t)b->x; if (b->mode == SC_SRC_BNBWD) al |= (uintptr_t)b->aux; }
  if (HW % 4 == 0 && (al & 15) == 0)
    hipLaunchKernelGGL(k_add_srcs<true>, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*a), b ? to_srcd(*b) : empty_srcd(),
                       b ? 1 : 0, out, C, HW, absmax);
  else
    hipLaunchKernelGGL(k_add_srcs<false>, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*a), b ? to_srcd(*b) : empty_srcd(),
                       b ? 1 : 0, out, C, HW, absmax);
  SC_LAUNCH_OK("sc_add_srcs");
  return SC_OK;
}

extern "C" int sc_add_srcs(const sc_src* a, const sc_src* b, float* out, int N, int C, int HW, sc_stream stream) {
  return sc_add_srcs_absmax(a, b, out, N, C, HW, nullptr, stream);
}

extern "C" int sc_apply_src(const sc_src* a, float* out, int N, int C, int HW, sc_stream stream) {
  return sc_add_srcs(a, nullptr, out, N, C, HW, stream);
}

extern "C" int sc_maxpool2x2(const sc_src* in, float* out, int N, int C, int Hout, int Wout, sc_stream stream) {
  SC_REQUIRE(in && in->x && out && N > 0 && C > 0 && Hout > 0 && Wout > 0 && in->up == 0, "sc_maxpool2x2: bad argument");
  SC_REQUIRE(in->mode != SC_SRC_BNBWD, "sc_maxpool2x2: forward sources only");
  dim3 grid((Hout * Wout + 1023) / 1024, C, N);
  hipLaunchKernelGGL(k_maxpool2x2, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), out, C, Hout, Wout);
  SC_LAUNCH_OK("sc_maxpool2x2");
  return SC_OK;
}

extern "C" int sc_upsample_bilinear2x(const sc_src* in, float* out, int N, int C, int Hin, int Win, sc_stream stream) {
  SC_REQUIRE(in && in->x && out && N > 0 && C > 0 && Hin > 0 && Win > 0 && in->up == 0, "sc_upsample_bilinear2x: bad argument");
  SC_REQUIRE(in->mode != SC_SRC_BNBWD, "sc_upsample_bilinear2x: forward sources only");
  dim3 grid((4 * Hin * Win + 1023) / 1024, C, N);
  hipLaunchKernelGGL(k_upsample_bilinear2x, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), out, C, Hin, Win);
  SC_LAUNCH_OK("sc_upsample_bilinear2x");
  return SC_OK;
}

extern "C" int sc_maxpool2x2_bwd(const sc_src* in, const float* gpool, float* gin, int accum, int N, int C, int Hout, int Wout,
                                 sc_stream stream) {
  SC_REQUIRE(in && in->x && gpool && gin && N > 0 && C > 0 && Hout > 0 && Wout > 0 && in->up == 0, "sc_maxpool2x2_bwd: bad argument");
  SC_REQUIRE(in->mode != SC_SRC_BNBWD, "sc_maxpool2x2_bwd: forward sources only");
  dim3 grid((Hout * Wout + 1023) / 1024, C, N);
  hipLaunchKernelGGL(k_maxpool2x2_bwd, grid, dim3(256), 0, (hipStream_t)stream, to_srcd(*in), gpool, gin, accum, C, Hout, Wout);
  SC_LAUNCH_OK("sc_maxpool2x2_bwd");
  return SC_OK;
}

extern "C" int sc_upsample_bilinear2x_bwd(const float* gout, float* gin, int N, int C, int Hin, int Win, sc_stream stream) {
  SC_REQUIRE(gout && gin && N > 0 && C > 0 && Hin > 0 && Win > 0, "sc_upsample_bilinear2x_bwd: bad argument");
  dim3 grid((Hin * Win + 1023) / 1024, C, N);
  hipLaunchKernelGGL(k_upsample_bilinear2x_bwd, grid, dim3(256), 0, (hipStream_t)stream, gout, gin, C, Hin, Win);
  SC_LAUNCH_OK("sc_upsample_bilinear2x_bwd");
  return SC_OK;
}

extern "C" int sc_downsum2x2(const float* in, float* out, int accum, int N, int C, int Hout, int Wout, sc_stream stream) {
  SC_REQUIRE(in && out && N > 0 && C > 0 && Hout > 0 && Wout > 0, "sc_downsum2x2: bad argument");
  const size_t total = (size_t)N * C * Hout * Wout;
  hipLaunchKernelGGL(k_downsum2x2, dim3(nblocks(total, 4096)), dim3(256), 0, (hipStream_t)stream, in, out, accum, Hout, Wout, total);
  SC_LAUNCH_OK("sc_downsum2x2");
  return SC_OK;
}

extern "C" int sc_fill_f64(double* p, double v, size_t n, sc_stream stream) {
  if (n == 0) return SC_OK;
  hipLaunchKernelGGL(k_fill_f64, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, v, n);
  SC_LAUNCH_OK("sc_fill_f64");
  return SC_OK;
}

extern "C" int sc_bce_logits_weighted(const float* logits, const float* target, const float* weight, float pos_weight,
                                      size_t n, double* loss_sum, float* dlogits, float* loss_px, sc_stream stream) {
  SC_REQUIRE(logits && target && n > 0, "sc_bce_logits_weighted: bad argument");
  hipLaunchKernelGGL(k_bce, dim3(nblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, logits, target, weight, pos_weight, n,
                     (float)(1.0 / (double)n), loss_sum, dlogits, loss_px);
  SC_LAUNCH_OK("sc_bce_logits_weighted");
  return SC_OK;
}

extern "C" int sc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                            float bias_correction2_sqrt, float grad_scale, const float* hp_dev, sc_stream stream) {
  SC_REQUIRE(param && grad && exp_avg && exp_avg_sq, "sc_adam_step: null argument");
  if (n == 0) return SC_OK;
  hipLaunchKernelGGL(k_adam, dim3(nblocks(n, 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                     lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt, grad_scale, hp_dev);
  SC_LAUNCH_OK("sc_adam_step");
  return SC_OK;
}

extern "C" int sc_adam_prepare(int64_t* step, const float* lr_dev, float beta1, float beta2, float* hp_dev, sc_stream stream) {
  SC_REQUIRE(step && lr_dev && hp_dev, "sc_adam_prepare: null argument");
  hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)step, lr_dev, beta1, beta2, hp_dev);
  SC_LAUNCH_OK("sc_adam_prepare");
  return SC_OK;
}

extern "C" int sc_threshold_masks(const float* logits, const float* target, int ge0, float* prediction, int64_t* pred_binary,
                                  int64_t* differences, int64_t* tile_count, int N, int HW, sc_stream stream) {
  SC_REQUIRE(logits && N > 0 && HW > 0, "sc_threshold_masks: bad argument");
  SC_REQUIRE(!differences || target, "sc_threshold_masks: differences need a target");
  dim3 grid((HW + 4095) / 4096, N);
  hipLaunchKernelGGL(k_threshold, grid, dim3(256), 0, (hipStream_t)stream, logits, target, ge0, prediction,
                     (long long*)pred_binary, (long long*)differences, (unsigned long long*)tile_count, HW);
  SC_LAUNCH_OK("sc_threshold_masks");
  return SC_OK;
}

extern "C" int sc_pred_classification(const int64_t* tile_count, int64_t* cls, int N, int H, int W, sc_stream stream) {
  SC_REQUIRE(tile_count && cls && N > 0, "sc_pred_classification: bad argument");
  // model_module.py:210-212: n_pixels = (10 * H * W) / 64**2 (float); sum > n_pixels
  const double thr = (10.0 * (double)H * (double)W) / 4096.0;
  hipLaunchKernelGGL(k_pred_cls, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const long long*)tile_count,
                     (long long*)cls, N, thr);
  SC_LAUNCH_OK("sc_pred_classification");
  return SC_OK;
}
