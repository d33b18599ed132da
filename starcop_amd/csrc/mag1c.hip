// mag1c: robust / albedo-corrected reweighted-L1 matched filter, one work-group per pixel group.
//
// Reference arithmetic: starcop/models/mag1c.py  rmf :284-348, acrwl1mf :177-280 (constants :55-57),
// group semantics: func_by_groups :116-174 and starcop/models/mag1c_emit.py:58-84.
//
// What is different from the reference's op sequence (same mathematics, see DESIGN.md "mag1c"):
//   * the reference recomputes   C_k = (modx-mu)^T (modx-mu) / N   (2*P*S^2 flop, 31x per group) with
//     modx_p = x_p - w_p*tau.  Because modx differs from x by a rank-1 term per pixel,
//         N*C_k = C_0 - v tau^T - tau v^T + q tau tau^T,   C_0 = sum (x_p-xbar)(x_p-xbar)^T  (once, fp64 MFMA)
//         v = sum_p w_p (x_p - xbar),  q = sum_p (w_p - wbar)^2,  mu_k = xbar - wbar*tau
//     so every iteration streams X twice (per-pixel dot products; v = X w) instead of re-multiplying it.
//   * every statistic, the Cholesky factorisation, both triangular solves and the per-pixel filter are
//     evaluated in fp64 for fp32 and fp64 radiances alike (the reference filters fp32 data in fp32).
// Layout: pixels of a group are packed band-major, x[s*Ppad + p]: a wave reads 64 consecutive pixels of one
// band (coalesced), per-pixel spectral reductions are register loops, per-band reductions are wave shuffles.
#include "sc_common.h"

namespace {

typedef double doublex4 __attribute__((ext_vector_type(4)));

struct Mag1cP {
  const void* x;
  const long long* xoff; const int* P; const int* Ppad; const long long* poff;
  const unsigned char* statmask;
  int G, S;
  long long npix;
  const double* templ;
  int num_iter;
  double alpha, kscale;
  int albedo_override, zero_override, sparse_override, apply_scaling;
  double* workC;     // [G][S*S]  C_0 per group
  double* mfw; double* Rw; double* wv;   // [npix] per-pixel state
  void* mf_out; void* alb_out;
  int* status;
  double* energy;    // [G][max(num_iter,0)+1] residual terms (compute_energy), or null
  double* logdet;    // [G] P/2 * log(1 / prod diag chol C) of the first covariance, or null
  // DIRECT mode of the resident tile kernel (sc_mag1c_args.cube): gather from the pixel-major cube, scatter to image order
  const float* cube; int S_total, band0; const long long* pix; void* sc_mf; void* sc_alb; int sc_f64;
};

#ifdef STARCOP_MAG1C_PROF
__device__ long long g_prof[32];
#define PROF(k) do { if (tid == 0 && g == 0) { const long long t_ = wall_clock64(); g_prof[k] += t_ - tprev; tprev = t_; } } while (0)
#else
#define PROF(k) do { } while (0)
#endif

constexpr int MAXS = 128;
constexpr int VEC = 128;

template <int NW>
__device__ __forceinline__ double block_sum_n(double v, double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) t += red[w];
  return t;
}

template <typename T, int NT, bool ENERGY = false>
__global__ __launch_bounds__(NT, NT == 512 ? 6 : 4) void k_mag1c(const Mag1cP p) {      // 512 threads: 3 groups per CU
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int g = blockIdx.x;
  const int S = p.S, LDC = S | 1;                         // odd pitch: conflict-free row-per-lane ds_read_b64
  const int S16 = (S + 15) & ~15;
  double* Cm = reinterpret_cast<double*>(smem);           // [S][LDC] working covariance / Cholesky factor
  double* vec = Cm + (size_t)S * LDC;
  double* xbar = vec, *tmpl = vec + VEC, *tau = vec + 2 * VEC, *mu = vec + 3 * VEC, *tnew = vec + 4 * VEC;
  double* cit = vec + 5 * VEC, *vv = vec + 6 * VEC, *col = vec + 7 * VEC;
  double* red = vec + 8 * VEC;       // [64]: [0,32) block sums, [32,64) broadcast scalars
  double* stg = red + 64;            // [S16][17] staging for the scatter matrix

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = p.P[g], pitch = p.Ppad[g];
  if (P <= 0) return;                                     // skipped group (sc_mag1c_layout_columns: too few valid pixels)
  const T* X = reinterpret_cast<const T*>(p.x) + p.xoff[g];
  const long long po = p.poff[g];
  const unsigned char* mk = p.statmask ? p.statmask + po : nullptr;
  double* mfw = p.mfw + po; double* Rw = p.Rw + po; double* wv = p.wv + po;
  double* C0 = p.workC + (size_t)g * S * S;
  const double N = (double)P;

#ifdef STARCOP_MAG1C_PROF
  long long tprev = wall_clock64();
#endif
  // ---------------- phase A: band means over the statistics pixels
  double nstat;
  {
    double c = 0.0;
    for (int q = tid; q < P; q += NT) c += (mk == nullptr || mk[q]) ? 1.0 : 0.0;
    nstat = block_sum_n<NW>(c, red);
  }
  for (int s = wave; s < S; s += NW) {
    double a = 0.0;
    for (int q = lane; q < P; q += 64 * 8) {
      T xr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xr[u] = X[(size_t)s * pitch + min(q + 64 * u, P - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int qq = q + 64 * u; if (qq < P && (mk == nullptr || mk[qq])) a += (double)xr[u]; }
    }
    a = wave_sum_d(a);
    if (lane == 0) xbar[s] = a / nstat;
  }
  for (int s = tid; s < S; s += NT) tmpl[s] = p.templ[s];
  __syncthreads();

  PROF(16);
  // ---------------- phase B: C_0 = sum_p (x_p - xbar)(x_p - xbar)^T  on v_mfma_f64_16x16x4_f64
  {
    const int nb = S16 >> 4;
    const int nblk = nb * (nb + 1) / 2;
    constexpr int NB = (36 + NW - 1) / NW;
    doublex4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = (doublex4){0.0, 0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < P; c0 += 16) {
      __syncthreads();
      for (int i = tid; i < S16 * 16; i += NT) {
        const int s = i >> 4, k = i & 15, q = c0 + k;
        double v = 0.0;
        if (s < S && q < P && (mk == nullptr || mk[q])) v = (double)X[(size_t)s * pitch + q] - xbar[s];
        stg[s * 17 + k] = v;
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int blk = wave + NW * b;
        if (blk < nblk) {
          // upper-triangular block index -> (bi <= bj)
          int bi = 0, rem = blk;
          while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
          const int bj = bi + rem;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const double a = stg[(bi * 16 + (lane & 15)) * 17 + kk * 4 + (lane >> 4)];
            const double bb = stg[(bj * 16 + (lane & 15)) * 17 + kk * 4 + (lane >> 4)];
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[b], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int blk = wave + NW * b;
      if (blk < nblk) {
        int bi = 0, rem = blk;
        while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
        const int bj = bi + rem;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = bi * 16 + (lane >> 4) + 4 * r, j = bj * 16 + (lane & 15);
          if (i < S && j < S) { C0[(size_t)i * S + j] = acc[b][r]; C0[(size_t)j * S + i] = acc[b][r]; }
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();

  PROF(17);
  // compute_energy (see k_mag1c_tile<.., ENERGY>): d = mean over all pixels - mean over the statistics pixels, kept in stg (free now)
  double* dvec = stg;
  if constexpr (ENERGY) {
    for (int s = wave; s < S; s += NW) {
      double a = 0.0;
      if (mk != nullptr) for (int q = lane; q < P; q += 64) a += (double)X[(size_t)s * pitch + q];
      a = wave_sum_d(a);
      if (lane == 0) dvec[s] = mk != nullptr ? a / N - xbar[s] : 0.0;
    }
    __syncthreads();
  }
  // ---------------- phase C: rmf (it == 0) then the reweighted-L1 iterations
  double sw = 0.0, sww = 0.0;
  bool notpd = false;
  const int last = p.num_iter < 0 ? 0 : p.num_iter;
  for (int it = 0; it <= last; ++it) {
    // (1) mean / target / covariance of the target-removed data
    double wbar = 0.0, q = 0.0;
    if (it > 0) { wbar = sw / nstat; q = sww - nstat * wbar * wbar; }
    for (int s = tid; s < S; s += NT) {
      const double m = (it > 0) ? xbar[s] - wbar * tau[s] : xbar[s];
      mu[s] = m;
      tnew[s] = tmpl[s] * m;
    }
    __syncthreads();
    const double oma = 1.0 - p.alpha;
    for (int i0 = tid; i0 < S * S; i0 += NT * 8) {
      double c0v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c0v[u] = C0[min(i0 + NT * u, S * S - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + NT * u;
        if (i < S * S) {
          const int r = i / S, c = i - r * S;
          if (c <= r) {
            double v = c0v[u];
            if (it > 0) v += -vv[r] * tau[c] - tau[r] * vv[c] + q * tau[r] * tau[c];
            v /= N;
            if (c != r) v *= oma;           // lerp towards the diagonal: C + alpha*(diag(C) - C)
            Cm[r * LDC + c] = v;
          }
        }
      }
    }
    __syncthreads();
    PROF(18);
    // (2) Cholesky, lower, in place.  S <= 64: left-looking inside wave 0 (lane = row; no block barriers -- with 31
    //     factorisations per group the 2*S barriers of the block version were the largest item of an EMIT iteration);
    //     larger S: right-looking over the whole block, two barriers per column.
    if (S <= 64) {
      if (wave == 0) {
        const int i = lane < S ? lane : S - 1;
        for (int j = 0; j < S; ++j) {
          double sacc = Cm[i * LDC + j];
          for (int kx = 0; kx < j; ++kx) sacc = fma(-Cm[i * LDC + kx], Cm[j * LDC + kx], sacc);
          const double djj = __shfl(sacc, j, 64);
          if (!(djj > 0.0)) notpd = true;
          const double d = sqrt(djj);
          if (lane >= j && lane < S) Cm[lane * LDC + j] = (lane == j) ? d : sacc / d;
        }
      }
      __syncthreads();
    } else {
      for (int j = 0; j < S; ++j) {
        const double djj = Cm[j * LDC + j];
        if (!(djj > 0.0)) notpd = true;
        const double d = sqrt(djj);
        for (int i = j + tid; i < S; i += NT) col[i] = (i == j) ? d : Cm[i * LDC + j] / d;
        __syncthreads();
        const int ti = tid >> 4, tk = tid & 15;
        for (int i = j + 1 + ti; i < S; i += NT / 16) {
          const double ci = col[i];
          for (int kx = j + 1 + tk; kx <= i; kx += 16) Cm[i * LDC + kx] -= ci * col[kx];
        }
        for (int i = j + tid; i < S; i += NT) Cm[i * LDC + j] = col[i];
        __syncthreads();
      }
    }
    PROF(19);
    // (3) cit = C^{-1} tnew : forward and backward substitution by wave 0 (each lane owns rows lane, lane+64)
    if (wave == 0) {
      double b0 = lane < S ? tnew[lane] : 0.0;
      double b1 = lane + 64 < S ? tnew[lane + 64] : 0.0;
      for (int j = 0; j < S; ++j) {                       // L y = b
        const double bj = __shfl(j < 64 ? b0 : b1, j & 63, 64);
        const double yj = bj / Cm[j * LDC + j];
        if (lane == (j & 63)) { if (j < 64) b0 = yj; else b1 = yj; }
        if (lane > j && lane < S) b0 -= Cm[lane * LDC + j] * yj;
        if (lane + 64 > j && lane + 64 < S) b1 -= Cm[(lane + 64) * LDC + j] * yj;
      }
      for (int j = S - 1; j >= 0; --j) {                  // L^T z = y
        const double bj = __shfl(j < 64 ? b0 : b1, j & 63, 64);
        const double zj = bj / Cm[j * LDC + j];
        if (lane == (j & 63)) { if (j < 64) b0 = zj; else b1 = zj; }
        if (lane < j) b0 -= Cm[j * LDC + lane] * zj;
        if (lane + 64 < j) b1 -= Cm[j * LDC + lane + 64] * zj;
      }
      if (lane < S) cit[lane] = b0;
      if (lane + 64 < S) cit[lane + 64] = b1;
      // (4) normaliser = target . C^{-1} target ;  mu . cit ; mu . mu
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      if (lane < S) { a0 += tnew[lane] * b0; a1 += mu[lane] * b0; a2 += mu[lane] * mu[lane]; }
      if (lane + 64 < S) { a0 += tnew[lane + 64] * b1; a1 += mu[lane + 64] * b1; a2 += mu[lane + 64] * mu[lane + 64]; }
      a0 = wave_sum_d(a0); a1 = wave_sum_d(a1); a2 = wave_sum_d(a2);
      if (lane == 0) { red[32] = a0; red[33] = a1; red[34] = a2; }
      if constexpr (ENERGY) {
        // s^T C_k^{-1} s with s = P (d + wbar tau): the same two substitutions for a second right-hand side; first iteration: the
        // log-determinant term P/2 log(1 / prod diag L)
        const double r0 = lane < S ? dvec[lane] + wbar * tau[lane] : 0.0;
        const double r1 = lane + 64 < S ? dvec[lane + 64] + wbar * tau[lane + 64] : 0.0;
        double c0 = r0, c1 = r1;
        for (int j = 0; j < S; ++j) {
          const double bj = __shfl(j < 64 ? c0 : c1, j & 63, 64);
          const double yj = bj / Cm[j * LDC + j];
          if (lane == (j & 63)) { if (j < 64) c0 = yj; else c1 = yj; }
          if (lane > j && lane < S) c0 -= Cm[lane * LDC + j] * yj;
          if (lane + 64 > j && lane + 64 < S) c1 -= Cm[(lane + 64) * LDC + j] * yj;
        }
        for (int j = S - 1; j >= 0; --j) {
          const double bj = __shfl(j < 64 ? c0 : c1, j & 63, 64);
          const double zj = bj / Cm[j * LDC + j];
          if (lane == (j & 63)) { if (j < 64) c0 = zj; else c1 = zj; }
          if (lane < j) c0 -= Cm[j * LDC + lane] * zj;
          if (lane + 64 < j) c1 -= Cm[j * LDC + lane + 64] * zj;
        }
        double e3 = r0 * c0 + r1 * c1, lg = 0.0;
        if (it == 0) { if (lane < S) lg += log(Cm[lane * LDC + lane]); if (lane + 64 < S) lg += log(Cm[(lane + 64) * LDC + lane + 64]); }
        e3 = wave_sum_d(e3); lg = wave_sum_d(lg);
        if (lane == 0) { p.energy[(size_t)g * (last + 1) + it] = N * N * e3; if (it == 0) p.logdet[g] = -0.5 * N * lg; }
      }
    }
    __syncthreads();
    PROF(20);
    double norm = red[32];
    const double mucit = red[33], mumu = red[34];
    if (it > 0 && norm < 1.0) norm = 1.0;               // normalizer.clamp_(min=1) (mag1c.py:264-266)
    // (5) per-pixel filter
    double lsw = 0.0, lsww = 0.0;
    for (int q0 = tid; q0 < P; q0 += NT) {
      double dot = 0.0, dmu = 0.0;
      const bool need_mu = (it == 0) && !p.albedo_override;
      for (int s0 = 0; s0 < S; s0 += 16) {            // 16 independent loads in flight per pixel, then the FMAs
        T xr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) xr[u] = X[(size_t)min(s0 + u, S - 1) * pitch + q0];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (s0 + u < S) {
            dot = fma((double)xr[u], cit[s0 + u], dot);
            if (need_mu) dmu = fma((double)xr[u], mu[s0 + u], dmu);
          }
        }
      }
      const double score = dot - mucit;
      double R, mf;
      if (it == 0) {
        R = p.albedo_override ? 1.0 : dmu / mumu;
        mf = score / (R * norm);
        if (!p.zero_override) mf = fmax(mf, 0.0);
      } else {
        R = Rw[q0];
        const double reg = p.sparse_override ? 0.0 : 1.0 / (R * (mfw[q0] + 1e-9));
        mf = fmax((score - reg) / (R * norm), 0.0);
      }
      mfw[q0] = mf;
      if (it == 0) Rw[q0] = R;
      const double w = (mk == nullptr || mk[q0]) ? p.kscale * R * mf : 0.0;
      wv[q0] = w;
      lsw += w; lsww += w * w;
    }
    PROF(21);
    if (it == last) break;
    sw = block_sum_n<NW>(lsw, red);
    sww = block_sum_n<NW>(lsww, red + 16);
    __threadfence_block();
    __syncthreads();
    PROF(22);
    // (6) v = X w - xbar * sum(w);  tau <- current target
    for (int s = wave; s < S; s += NW) {
      double a = 0.0;
      for (int q0 = lane; q0 < P; q0 += 64 * 8) {    // 16 independent loads in flight per lane
        T xr[8]; double wr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int qq = min(q0 + 64 * u, P - 1); xr[u] = X[(size_t)s * pitch + qq]; wr[u] = wv[qq]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (q0 + 64 * u < P) a = fma((double)xr[u], wr[u], a);
      }
      a = wave_sum_d(a);
      if (lane == 0) { vv[s] = a - xbar[s] * sw; tau[s] = tnew[s]; }
    }
    __syncthreads();
    PROF(23);
  }
  // ---------------- outputs
  const double scale = (p.num_iter >= 0 || p.apply_scaling) ? 1e5 : 1.0;
  T* mo = reinterpret_cast<T*>(p.mf_out) + po;
  T* ao = reinterpret_cast<T*>(p.alb_out) + po;
  for (int q0 = tid; q0 < P; q0 += NT) {
    mo[q0] = (T)(mfw[q0] * scale);
    ao[q0] = (T)Rw[q0];
  }
  if (tid == 0) p.status[g] = notpd ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// alpha == 0 (the AVIRIS-NG driver, process_aviris.py:210): no shrinkage, so C_k differs from C_0/N by an exact rank-2
// term and C_k^{-1} t follows from ONE factorisation per group by the Woodbury identity:
//   B0 = C_0^{-1} = W/N (W = (C_0/N)^{-1}, formed once: Cholesky -> triangular inverse -> X^T X),  U = [v tau],
//   M = [[0,-1],[-1,q]]:  (C_0 + U M U^T)^{-1} b = B0 b - B0 U (M^{-1} + U^T B0 U)^{-1} U^T B0 b
// Per iteration: two mat-vecs (W v, W t_new; W tau is last iteration's W t_new), a 2x2 solve, two streaming passes.
// NT threads per group (1024: the 125 KB matrix in LDS allows one work-group per CU, so the work-group itself has to fill it)
template <typename T, int NT>
__global__ __launch_bounds__(NT) void k_mag1c_fast(const Mag1cP p) {
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int g = blockIdx.x;
  const int S = p.S, LDC = S | 1;                           // odd pitch: conflict-free row-per-lane ds_read_b64
  const int S16 = (S + 15) & ~15;
  double* Cm = reinterpret_cast<double*>(smem);             // [S][LDC]: A -> L (lower) + X=L^{-1} (upper, transposed) -> W
  double* vec = Cm + (size_t)S * LDC;
  double* xbar = vec, *tmpl = vec + VEC, *tau = vec + 2 * VEC, *mu = vec + 3 * VEC, *tnew = vec + 4 * VEC;
  double* cit = vec + 5 * VEC, *vv = vec + 6 * VEC, *col = vec + 7 * VEC;
  double* p1 = vec + 8 * VEC, *p2 = vec + 9 * VEC, *p3 = vec + 10 * VEC;
  double* red = vec + 11 * VEC;      // [64]: [0,32) block sums, [32,64) broadcast scalars
  double* stg = red + 64;            // [S16][17]; after phase B reused as scratch (4 x 128 partial mat-vec sums)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = p.P[g], pitch = p.Ppad[g];
  if (P <= 0) return;                                     // skipped group (sc_mag1c_layout_columns: too few valid pixels)
  const T* X = reinterpret_cast<const T*>(p.x) + p.xoff[g];
  const long long po = p.poff[g];
  const unsigned char* mk = p.statmask ? p.statmask + po : nullptr;
  double* mfw = p.mfw + po; double* Rw = p.Rw + po; double* wv = p.wv + po;
  double* C0 = p.workC + (size_t)g * S * S;
  const double N = (double)P;

  // ---------------- phase A: band means
  double nstat;
  {
    double c = 0.0;
    for (int q = tid; q < P; q += NT) c += (mk == nullptr || mk[q]) ? 1.0 : 0.0;
    nstat = block_sum_n<NW>(c, red);
  }
  for (int s = wave; s < S; s += NW) {
    double a = 0.0;
    for (int q = lane; q < P; q += 64 * 8) {
      T xr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xr[u] = X[(size_t)s * pitch + min(q + 64 * u, P - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int qq = q + 64 * u; if (qq < P && (mk == nullptr || mk[qq])) a += (double)xr[u]; }
    }
    a = wave_sum_d(a);
    if (lane == 0) xbar[s] = a / nstat;
  }
  for (int s = tid; s < S; s += NT) tmpl[s] = p.templ[s];
  __syncthreads();

  // ---------------- phase B: C_0 on the fp64 MFMA, written as A = C_0/N straight into LDS (both triangles)
  {
    const int nb = S16 >> 4;
    const int nblk = nb * (nb + 1) / 2;
    constexpr int NB = (36 + NW - 1) / NW;       // 16x16 blocks of the upper triangle (<= 36 for S <= 128) per wave
    doublex4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = (doublex4){0.0, 0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < P; c0 += 16) {
      __syncthreads();
      for (int i = tid; i < S16 * 16; i += NT) {
        const int s = i >> 4, k = i & 15, q = c0 + k;
        double v = 0.0;
        if (s < S && q < P && (mk == nullptr || mk[q])) v = (double)X[(size_t)s * pitch + q] - xbar[s];
        stg[s * 17 + k] = v;
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int blk = wave + NW * b;
        if (blk < nblk) {
          int bi = 0, rem = blk;
          while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
          const int bj = bi + rem;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const double a = stg[(bi * 16 + (lane & 15)) * 17 + kk * 4 + (lane >> 4)];
            const double bb = stg[(bj * 16 + (lane & 15)) * 17 + kk * 4 + (lane >> 4)];
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[b], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int blk = wave + NW * b;
      if (blk < nblk) {
        int bi = 0, rem = blk;
        while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
        const int bj = bi + rem;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = bi * 16 + (lane >> 4) + 4 * r, j = bj * 16 + (lane & 15);
          if (i < S && j < S) { Cm[i * LDC + j] = acc[b][r] / N; Cm[j * LDC + i] = acc[b][r] / N; }
        }
      }
    }
  }
  __syncthreads();
  // ---------------- Cholesky of A (lower, in place), once per group
  bool notpd = false;
  for (int j = 0; j < S; ++j) {
    const double djj = Cm[j * LDC + j];
    if (!(djj > 0.0)) notpd = true;
    const double d = sqrt(djj);
    for (int i = j + tid; i < S; i += NT) col[i] = (i == j) ? d : Cm[i * LDC + j] / d;
    __syncthreads();
    const int ti = tid >> 4, tk = tid & 15;
    for (int i = j + 1 + ti; i < S; i += NT / 16) {
      const double ci = col[i];
      for (int k = j + 1 + tk; k <= i; k += 16) Cm[i * LDC + k] -= ci * col[k];
    }
    for (int i = j + tid; i < S; i += NT) Cm[i * LDC + j] = col[i];
    __syncthreads();
  }
  // ---------------- X = L^{-1}: thread j builds column j; X[i][j] (i > j) lives at Cm[j][i] (upper triangle), 1/L[j][j] in col[]
  if (tid < S) col[tid] = 1.0 / Cm[tid * LDC + tid];
  __syncthreads();
  {
    // column j is built by the TPC lanes tid = j*TPC .. j*TPC+TPC-1 (adjacent lanes of one wave): they split the inner
    // product over k and combine with DPP-free xor shuffles; rows i are sequential (forward substitution)
    constexpr int TPC = NT / 128;                            // 8 at NT = 1024, 2 at NT = 256
    const int j = tid / TPC, part = tid % TPC;
    if (j < S) {
      for (int i = j + 1; i < S; ++i) {
        double a = (part == 0) ? Cm[i * LDC + j] * col[j] : 0.0;            // L[i][j] * X[j][j]
        for (int k = j + 1 + part; k < i; k += TPC) a = fma(Cm[i * LDC + k], Cm[j * LDC + k], a);
#pragma unroll
        for (int o = TPC >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (part == 0) Cm[j * LDC + i] = -a * col[i];
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
  // ---------------- W = X^T X -> global scratch (the C_0 slot), then back into LDS as a full symmetric matrix
  for (int e = tid; e < S * S; e += NT) {
    const int a = e / S, b = e - a * S;
    if (b > a) continue;
    // sum_{i >= a} X[i][a] X[i][b]   (a >= b);  X[i][c] = (i == c) ? col[c] : Cm[c][i]
    double acc = (a == b) ? col[a] * col[a] : col[a] * Cm[b * LDC + a];
    for (int i = a + 1; i < S; ++i) acc = fma(Cm[a * LDC + i], Cm[b * LDC + i], acc);
    C0[(size_t)a * S + b] = acc; C0[(size_t)b * S + a] = acc;
  }
  __threadfence_block();
  __syncthreads();
  for (int e = tid; e < S * S; e += NT) { const int a = e / S, b = e - a * S; Cm[a * LDC + b] = C0[e]; }
  for (int s = tid; s < S; s += NT) { p2[s] = 0.0; vv[s] = 0.0; tau[s] = 0.0; }
  __syncthreads();
  // compute_energy (see k_mag1c_tile<.., ENERGY>): d = mean over all pixels - mean over the statistics pixels; d.W d and the
  // log-determinant term once; d ends up in the `col` slot (1 / diag L, needed for that term first, is dead afterwards)
  if (p.energy) {
    for (int s = wave; s < S; s += NW) {
      double a = 0.0;
      if (mk != nullptr) for (int q0 = lane; q0 < P; q0 += 64) a += (double)X[(size_t)s * pitch + q0];
      a = wave_sum_d(a);
      if (lane == 0) cit[s] = mk != nullptr ? a / N - xbar[s] : 0.0;
    }
    __syncthreads();
    double dw = 0.0, lg = 0.0;
    for (int r = tid; r < S; r += NT) {
      double a = 0.0;
      for (int c = 0; c < S; ++c) a = fma(Cm[r * LDC + c], cit[c], a);
      dw += cit[r] * a; lg += log(col[r]);
    }
    dw = block_sum_n<NW>(dw, red);
    lg = block_sum_n<NW>(lg, red + 16);
    if (tid == 0) { red[50] = dw; p.logdet[g] = 0.5 * N * lg; }
    __syncthreads();
    for (int s = tid; s < S; s += NT) col[s] = cit[s];
    __syncthreads();
  }

  // ---------------- rmf (it == 0) then the reweighted-L1 iterations
  double sw = 0.0, sww = 0.0;
  const int last = p.num_iter < 0 ? 0 : p.num_iter;
  for (int it = 0; it <= last; ++it) {
    double wbar = 0.0, q = 0.0;
    if (it > 0) { wbar = sw / nstat; q = sww - nstat * wbar * wbar; }
    for (int s = tid; s < S; s += NT) {
      const double m = (it > 0) ? xbar[s] - wbar * tau[s] : xbar[s];
      mu[s] = m;
      tnew[s] = tmpl[s] * m;
    }
    __syncthreads();
    // p1 = W v, p3 = W t_new: row r = tid & 127, vector = (tid >> 7) & 1, the NT/256 column parts of a row are summed via LDS
    {
      constexpr int NP = NT / 256;
      const int r = tid & 127, which = (tid >> 7) & 1, part = tid >> 8;
      const double* u = which ? tnew : vv;
      double a0 = 0.0, a1 = 0.0;
      if (r < S) {
        int c = 2 * part;
        for (; c + 1 < S; c += 2 * NP) { a0 = fma(Cm[r * LDC + c], u[c], a0); a1 = fma(Cm[r * LDC + c + 1], u[c + 1], a1); }
        if (c < S) a0 = fma(Cm[r * LDC + c], u[c], a0);
      }
      if (NP == 1) {
        if (r < S) (which ? p3 : p1)[r] = a0 + a1;
      } else {
        stg[(part * 2 + which) * 128 + r] = a0 + a1;
        __syncthreads();
        if (tid < 256 && r < S) {
          double t = 0.0;
#pragma unroll
          for (int q2 = 0; q2 < NP; ++q2) t += stg[(q2 * 2 + which) * 128 + r];
          (which ? p3 : p1)[r] = t;
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
      double d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // v.p1, v.p2, tau.p2, p1.b, p2.b, b.p3, mu.p1, mu.p2, mu.p3
      for (int s = lane; s < S; s += 64) {
        d[0] = fma(vv[s], p1[s], d[0]); d[1] = fma(vv[s], p2[s], d[1]); d[2] = fma(tau[s], p2[s], d[2]);
        d[3] = fma(p1[s], tnew[s], d[3]); d[4] = fma(p2[s], tnew[s], d[4]); d[5] = fma(tnew[s], p3[s], d[5]);
        d[6] = fma(mu[s], p1[s], d[6]); d[7] = fma(mu[s], p2[s], d[7]); d[8] = fma(mu[s], p3[s], d[8]);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) d[k] = wave_sum_d(d[k]);
      double mm = 0.0;
      for (int s = lane; s < S; s += 64) mm = fma(mu[s], mu[s], mm);
      mm = wave_sum_d(mm);
      double y1 = 0.0, y2 = 0.0;
      if (it > 0) {
        // G = M^{-1} + U^T B0 U,  M^{-1} = [[-q,-1],[-1,0]],  B0 = W/N ;  G y = U^T B0 b
        const double g11 = -q + d[0] / N, g12 = -1.0 + d[1] / N, g22 = d[2] / N;
        const double z1 = d[3] / N, z2 = d[4] / N;
        const double det = g11 * g22 - g12 * g12;
        y1 = (z1 * g22 - z2 * g12) / det;
        y2 = (g11 * z2 - g12 * z1) / det;
      }
      if (lane == 0) {
        red[32] = d[5] - y1 * d[3] - y2 * d[4];           // normaliser  t . C^{-1} t
        red[33] = d[8] - y1 * d[6] - y2 * d[7];           // mu . C^{-1} t
        red[34] = mm; red[35] = y1; red[36] = y2;
      }
      if (p.energy) {
        // s^T C_k^{-1} s = N [ s^T B0 s - b^T G^{-1} b ],  s = P (d + wbar tau),  b = U^T B0 s
        double dp1 = 0.0, dp2 = 0.0;
        for (int s = lane; s < S; s += 64) { dp1 = fma(col[s], p1[s], dp1); dp2 = fma(col[s], p2[s], dp2); }
        dp1 = wave_sum_d(dp1); dp2 = wave_sum_d(dp2);
        const double sBs = N * (red[50] + 2.0 * wbar * dp2 + wbar * wbar * d[2]);
        double bGb = 0.0;
        if (it > 0) {
          const double g11 = -q + d[0] / N, g12 = -1.0 + d[1] / N, g22 = d[2] / N;
          const double b1 = dp1 + wbar * d[1], b2 = dp2 + wbar * d[2];
          bGb = (b1 * b1 * g22 - 2.0 * b1 * b2 * g12 + b2 * b2 * g11) / (g11 * g22 - g12 * g12);
        }
        if (lane == 0) p.energy[(size_t)g * (last + 1) + it] = N * (sBs - bGb);
      }
    }
    __syncthreads();
    {
      const double y1 = red[35], y2 = red[36];
      for (int s = tid; s < S; s += NT) cit[s] = p3[s] - y1 * p1[s] - y2 * p2[s];
    }
    __syncthreads();
    double norm = red[32];
    const double mucit = red[33], mumu = red[34];
    if (!(norm == norm)) notpd = true;
    if (it > 0 && norm < 1.0) norm = 1.0;
    double lsw = 0.0, lsww = 0.0;
    for (int q0 = tid; q0 < P; q0 += NT) {
      double dot = 0.0, dmu = 0.0;
      const bool need_mu = (it == 0) && !p.albedo_override;
      for (int s0 = 0; s0 < S; s0 += 16) {            // 16 independent loads in flight per pixel, then the FMAs
        T xr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) xr[u] = X[(size_t)min(s0 + u, S - 1) * pitch + q0];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (s0 + u < S) {
            dot = fma((double)xr[u], cit[s0 + u], dot);
            if (need_mu) dmu = fma((double)xr[u], mu[s0 + u], dmu);
          }
        }
      }
      const double score = dot - mucit;
      double R, mf;
      if (it == 0) {
        R = p.albedo_override ? 1.0 : dmu / mumu;
        mf = score / (R * norm);
        if (!p.zero_override) mf = fmax(mf, 0.0);
      } else {
        R = Rw[q0];
        const double reg = p.sparse_override ? 0.0 : 1.0 / (R * (mfw[q0] + 1e-9));
        mf = fmax((score - reg) / (R * norm), 0.0);
      }
      mfw[q0] = mf;
      if (it == 0) Rw[q0] = R;
      const double w = (mk == nullptr || mk[q0]) ? p.kscale * R * mf : 0.0;
      wv[q0] = w;
      lsw += w; lsww += w * w;
    }
    if (it == last) break;
    sw = block_sum_n<NW>(lsw, red);
    sww = block_sum_n<NW>(lsww, red + 16);
    __threadfence_block();
    __syncthreads();
    for (int s = wave; s < S; s += NW) {
      double a = 0.0;
      for (int q0 = lane; q0 < P; q0 += 64 * 8) {    // 16 independent loads in flight per lane
        T xr[8]; double wr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int qq = min(q0 + 64 * u, P - 1); xr[u] = X[(size_t)s * pitch + qq]; wr[u] = wv[qq]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (q0 + 64 * u < P) a = fma((double)xr[u], wr[u], a);
      }
      a = wave_sum_d(a);
      if (lane == 0) { vv[s] = a - xbar[s] * sw; tau[s] = tnew[s]; p2[s] = p3[s]; }
    }
    __syncthreads();
  }
  const double scale = (p.num_iter >= 0 || p.apply_scaling) ? 1e5 : 1.0;
  T* mo = reinterpret_cast<T*>(p.mf_out) + po;
  T* ao = reinterpret_cast<T*>(p.alb_out) + po;
  for (int q0 = tid; q0 < P; q0 += NT) {
    mo[q0] = (T)(mfw[q0] * scale);
    ao[q0] = (T)Rw[q0];
  }
  if (tid == 0) p.status[g] = notpd ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// k_mag1c_tile<JB>: every fp32 group (any P, S <= 16*JB, alpha = 0 or not) with the radiances held in a REGISTER TILE.
// 512 threads; thread (pixel group pg = wave*4 + lane/16, band lane bg = lane%16) keeps the 16 x JB tile
//   x[chunk*512 + pg*16 + i][bg + 16*j],   i < 16, j < JB          (JB = 8: 128 registers, S <= 128;  JB = 4: 64 registers, S <= 64:
//   half the conversions / products per pixel row for the same reduce-scatter)
// of one 512-pixel chunk of the group.  A group of P <= 512 pixels (JB = 8) is loaded ONCE and stays on chip for the whole kernel
// (P x S x 4 B = 256 KB at 512 x 125 does not fit the LDS beside the S x S matrix, but it fits the registers of the work-group);
// larger groups stream their chunks through the same tile -- once per iteration, not twice: the weight of a pixel needs only that
// pixel's own filter output, so v = X^T w accumulates in the pass that computes the outputs.  From the tile:
//   * band means: in-lane sums, lane permutes over the wave's 4 pixel groups, the 8 waves through LDS;
//   * C_0 on the fp64 MFMA with BOTH operands straight from registers: lane (bg, pg%4) of the tile is exactly lane (row bg, k) of a
//     16x16x4 operand of band block j, so  C_0[bi][bj] += x(i, bi) x(i, bj)^T  for i = 0..15 needs no staging, no barrier and no
//     memory traffic; every wave accumulates its pixels, NACC blocks per pass, and the eight partial blocks meet in LDS in a
//     fixed order (deterministic);
//   * per iteration: the per-pixel dots are JB in-lane products per pixel + a reduce-scatter over the 16 band lanes, v = X^T w is
//     16 x JB in-lane products + the same cross-lane / cross-wave reduction as the means.
// The S x S system is solved in 16 x 16 blocks on the fp64 MFMA (spd_inverse_blocked): alpha = 0 inverts C_0/N once and follows
// the rank-2 changes by the Woodbury identity (see k_mag1c_fast); alpha != 0 (the EMIT driver: the shrinkage is not low rank)
// rebuilds and inverts C_k every iteration -- 18 us instead of the 88 us of a factorisation + two solves by a single wave.
constexpr int RNT = 512, RNW = 8;

template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// the block pairs (bi <= bj) of the upper triangle of an n x n block matrix, row-major
constexpr int tri_pair_bi(int k, int n) { int bi = 0; while (k >= n - bi) { k -= n - bi; ++bi; } return bi; }
constexpr int tri_pair_bj(int k, int n) { int bi = 0; while (k >= n - bi) { k -= n - bi; ++bi; } return bi + k; }

// one row of the register tile through an empty volatile asm: its float -> double conversions stay where the row is used (the asm
// statements keep their order, and the sums a row feeds pass through one as well) instead of 128 conversions hoisted into 256
// registers -- the difference between no spill and 40-odd, each of which is a memory round trip inside the iteration
template <int JB>
__device__ __forceinline__ void rowbar(float (&r)[JB]) {
  if constexpr (JB == 8) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
  else asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
}
template <int JB>
__device__ __forceinline__ void accbar(double (&a)[JB]) {
  if constexpr (JB == 8) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
  else asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
}

template <int CTRL>
__device__ __forceinline__ double res_comb(bool sel, double lo, double hi) {
  return (sel ? hi : lo) + dpp_mov_d<CTRL>(sel ? lo : hi);
}
template <int JB>
__device__ __forceinline__ double tile_row_dot(float (&xt)[16][JB], const double (&cj)[JB], int i) {
  double t = 0.0;
  rowbar<JB>(xt[i]);
#pragma unroll
  for (int j = 0; j < JB; ++j) t = fma((double)xt[i][j], cj[j], t);
  asm volatile("" : "+v"(t));
  return t;
}
// (rows in ascending order: lane bit 0 first -- quad xor 1, quad xor 2, lane xor 4, row rotation by 8 -- so that the four-row pieces
// a tile is loaded in are consumed one after the other, which is what lets the streaming kernels refill a piece in place as soon as
// the X^T w pass is done with it)
template <int JB>
__device__ __forceinline__ double tile_pixel_dots(float (&xt)[16][JB], const double (&cj)[JB], int bz) {
  const bool b3 = bz & 8, b2 = bz & 4, b1 = bz & 2, b0 = bz & 1;
  double r8[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    double r4[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      double r2[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int i = 8 * h + 4 * a + 2 * g;
        const double d0 = tile_row_dot<JB>(xt, cj, i), d1 = tile_row_dot<JB>(xt, cj, i + 1);
        r2[g] = res_comb<0xB1>(b0, d0, d1);
      }
      r4[a] = res_comb<0x4E>(b1, r2[0], r2[1]);
    }
    r8[h] = (b2 ? r4[1] : r4[0]) + __shfl_xor(b2 ? r4[0] : r4[1], 4, 64);
  }
  return res_comb<0x128>(b3, r8[0], r8[1]);
}

// per-band totals of per-lane partial sums aj[j] (band bg + 16*j): over the wave's four pixel groups by lane permutes, over the eight
// waves through LDS (stg: 8 x 128).  Thread tz < 128 returns the total of band tz.  Contains one barrier.
template <int JB>
__device__ __forceinline__ double tile_band_total(double (&aj)[JB], double* stg, int tz) {
  const int lz = tz & 63;
#pragma unroll
  for (int j = 0; j < JB; ++j) { aj[j] += __shfl_xor(aj[j], 16, 64); aj[j] += __shfl_xor(aj[j], 32, 64); }
  if (lz < 16) {
#pragma unroll
    for (int j = 0; j < JB; ++j) stg[(tz >> 6) * 128 + lz + 16 * j] = aj[j];
  }
  __syncthreads();
  double t = 0.0;
  if (tz < 16 * JB) {
#pragma unroll
    for (int w = 0; w < RNW; ++w) t += stg[w * 128 + tz];
  }
  return t;
}

// one 512-pixel chunk into the tile: 4*JB float4 loads per thread, all in flight at once.  Returns the statistics mask of the lane's
// 16 pixels (bit i: pixel q0 + i exists and counts): every lane looks at ITS pixel q0 + bg, the 16-lane row votes.
template <int JB>
__device__ __forceinline__ unsigned tile_load(float (&xt)[16][JB], const float* X, int pitch, int S, int P, const unsigned char* mk,
                                              int q0, int bg, int lane) {
  const bool pgok = q0 < pitch;                             // (the pack kernel wrote zeros into [P, pitch), pitch % 64 == 0)
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int sj = bg + 16 * j;
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pgok && sj < S) {
      const float* src = X + (size_t)sj * pitch + q0;
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(src + 4 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { xt[4 * u][j] = a[u].x; xt[4 * u + 1][j] = a[u].y; xt[4 * u + 2][j] = a[u].z; xt[4 * u + 3][j] = a[u].w; }
  }
  const bool mine_ok = q0 + bg < P && (mk == nullptr || mk[q0 + bg]);
  return (unsigned)(__ballot(mine_ok) >> (lane & 48)) & 0xffffu;
}

// DIRECT mode: the same tile from the pixel-major cube -- pixel q0 + i is row pg[q0 + i] of the cube, its bands bg + 16 j one 64-byte
// piece per 16 lanes; the 16 pixel indices of the group in one load, all 16 * JB element loads in flight together.  Read ONCE per group:
// the resident kernel keeps the tile.  (tools/prof_mag1c_phases.py, configs[2]: this load 8.5-9 us per group against 5.7 us from the
// packed copy; a pixel-wise variant -- 256 contiguous bytes of ONE pixel per wave instruction, SGPR base, then a 4 x 4 exchange between
// the row groups by ds_bpermute -- took 22 us: the exchange costs more than the longer runs gain.)
template <int JB>
__device__ __forceinline__ unsigned tile_load_direct(float (&xt)[16][JB], const float* __restrict__ cube, int S_total, int band0,
                                                     const long long* __restrict__ pg, int S, int P, const unsigned char* mk, int q0, int bg,
                                                     int lane) {
  const long long myidx = (q0 + bg < P) ? pg[q0 + bg] : -1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long long idx = __shfl(myidx, (lane & 48) | i, 64);
    const float* row = cube + (size_t)(idx < 0 ? 0 : idx) * S_total + band0;
#pragma unroll
    for (int j = 0; j < JB; ++j) xt[i][j] = (idx >= 0 && bg + 16 * j < S) ? row[bg + 16 * j] : 0.f;
  }
  const bool mine_ok = q0 + bg < P && (mk == nullptr || mk[q0 + bg]);
  return (unsigned)(__ballot(mine_ok) >> (lane & 48)) & 0xffffu;
}

// ---- 16 x 16 block algebra on the fp64 MFMA (v_mfma_f64_16x16x4: lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15];
// D[(l >> 4) + 4*r][l & 15] in register r).  Blocks live in LDS with arbitrary row / column strides, so a transposed operand is a
// swap of two arguments.
template <bool NEG = false>
__device__ __forceinline__ void blk_mma(doublex4& acc, const double* a, int ar, int ak, const double* b, int bk, int bc, int lane) {
  const int r = lane & 15, kq = lane >> 4;
  double av[4], bv[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { av[kk] = a[r * ar + (4 * kk + kq) * ak]; bv[kk] = b[(4 * kk + kq) * bk + r * bc]; }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -av[kk] : av[kk], bv[kk], acc, 0, 0, 0);
}
template <bool NEG = false>
__device__ __forceinline__ void blk_store(double* d, int dr, int dc, const doublex4& acc, int lane) {
  const int c = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) d[(kq + 4 * r) * dr + c * dc] = NEG ? -acc[r] : acc[r];
}
__device__ __forceinline__ doublex4 blk_load(const double* d, int dr, int dc, int lane) {
  const int c = lane & 15, kq = lane >> 4;
  doublex4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = d[(kq + 4 * r) * dr + c * dc];
  return acc;
}
__device__ __forceinline__ double readlane_d(double v, int l) {           // l uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// Cholesky factor L of one 16 x 16 diagonal block AND X = L^{-1} in one sweep, by one wave: lane r < 16 keeps row r of the block and of
// X in registers; step j takes the pivot and the column of L from the lanes that own them (v_readlane), no LDS, no barrier.
// Only X is stored (Dk [16][17], zero above the diagonal): the panel solve, the inverse and W all use X_kk, nothing uses L_kk again.
__device__ __forceinline__ bool res_diag_factor(const double* blk, int LD, double* Dk, int lane) {
  const int r = lane & 15, q = lane >> 4;
  // every 16-lane group keeps the whole row r of the block (the factor's sweep is replicated: its column of L is then at hand in
  // every group at no cost); of X the group q keeps the columns q, q+4, q+8, q+12 only -- a quarter of the longest part of the sweep
  double a[16], x[4];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = blk[r * LD + c];
#pragma unroll
  for (int cq = 0; cq < 4; ++cq) x[cq] = (4 * cq + q == r) ? 1.0 : 0.0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double d = readlane_d(a[j], j);
    if (!(d > 0.0)) bad = true;
    double rd = __builtin_amdgcn_rsq(d);                 // v_rsq_f64 + two Newton steps (the sqrt / divide pair is a ~250-cycle chain per pivot)
    rd = rd * fma(-0.5 * d * rd, rd, 1.5);
    rd = rd * fma(-0.5 * d * rd, rd, 1.5);
    const double lr = a[j] * rd;                        // L[r][j] (r >= j)
#pragma unroll
    for (int c = j + 1; c < 16; ++c) a[c] = fma(-lr, readlane_d(lr, c), a[c]);
    // [L | I] -> [I | X]: row j scaled (by its own lanes), then taken out of the rows below -- with per-step factors (1 or rd, lr or
    // 0) instead of per-element selects.  Columns beyond j hold zeros in row j: they take part without effect.
    const double sc = (r == j) ? rd : 1.0, lm = (r > j) ? lr : 0.0;
    const int src = (lane & 48) | j;                    // row j of this lane group
#pragma unroll
    for (int cq = 0; cq <= (j >> 2); ++cq) {
      x[cq] *= sc;
      x[cq] = fma(-lm, __shfl(x[cq], src, 64), x[cq]);
    }
  }
#pragma unroll
  for (int cq = 0; cq < 4; ++cq) Dk[r * 17 + 4 * cq + q] = (4 * cq + q <= r) ? x[cq] : 0.0;
  return bad;
}


// A (SPD, S16 x S16 in LDS, lower triangle + diagonal valid, identity beyond S) -> A^{-1} as a full symmetric matrix, in place, all
// of it in 16 x 16 blocks on the fp64 MFMA.  Dx: [nb][16][17] scratch.  flag: set to 1 when a pivot is not positive.  All threads
// of the work-group call it; ends with a barrier.  (The unblocked forms it replaces: 125 steps of two barriers for the factor, 124
// dependent dot products for the inverse, 62 LDS reads per element of the product -- 0.3 of the 0.67 ms of a 125-band group.)
// Phase (1) alone: the blocked Cholesky factor L (strict lower blocks of Cm; the diagonal blocks are NOT written back) and the inverses
// X_kk = L_kk^{-1} of its diagonal blocks (Dx [nb][16][17]).  flag: set to 1 when a pivot is not positive.  Ends with a barrier.
__device__ __forceinline__ void spd_cholesky_blocked(double* Cm, int LD, int nb, double* Dx, double* flag, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    // (1) right-looking Cholesky.  Per block column k: panel L_ik = A_ik X_kk^T, trailing A_ij -= L_ik L_jk^T; wave 0 takes the next
    // diagonal block first and factors it while the others finish the update: two barriers per block column.
    if (wave == 0 && res_diag_factor(Cm, LD, Dx, lane) && lane == 0) *flag = 1.0;
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
      for (int i = k + 1 + wave; i < nb; i += RNW) {
        double* blk = Cm + (size_t)(16 * i) * LD + 16 * k;
        doublex4 acc = (doublex4){0.0, 0.0, 0.0, 0.0};
        blk_mma(acc, blk, LD, 1, Dx + k * 272, 1, 17, lane);
        blk_store(blk, LD, 1, acc, lane);
      }
      __syncthreads();
      if (k + 1 == nb) break;
      const int n = nb - k - 1, cnt = n * (n + 1) / 2;
      for (int t0 = (wave == 0) ? 0 : wave; t0 < ((wave == 0) ? 1 : cnt); t0 += RNW - 1) {
        int t = t0, ii = 0;
        while (t >= ii + 1) { t -= ii + 1; ++ii; }
        const int i = k + 1 + ii, j = k + 1 + t;
        double* blk = Cm + (size_t)(16 * i) * LD + 16 * j;
        doublex4 acc = blk_load(blk, LD, 1, lane);
        blk_mma<true>(acc, Cm + (size_t)(16 * i) * LD + 16 * k, LD, 1, Cm + (size_t)(16 * j) * LD + 16 * k, 1, LD, lane);
        blk_store(blk, LD, 1, acc, lane);
      }
      if (wave == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (res_diag_factor(Cm + (size_t)(16 * (k + 1)) * LD + 16 * (k + 1), LD, Dx + (k + 1) * 272, lane) && lane == 0) *flag = 1.0;
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ void spd_inverse_blocked(double* Cm, int LD, int nb, double* Dx, double* flag, int tid) {
  spd_cholesky_blocked(Cm, LD, nb, Dx, flag, tid);
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S16 = nb * 16;
  {
    // (2) X = L^{-1}.  First Y_im = X_ii L_im in place (all blocks at once), then block column j by wave j alone:
    //   X_ij = -sum_{m = j .. i-1} Y_im X_mj   (i > j),  stored TRANSPOSED in the free upper block (j, i) -- the form W reads.
    {
      const int cnt = nb * (nb - 1) / 2;
      for (int t0 = wave; t0 < cnt; t0 += RNW) {
        int t = t0, ii = 0;
        while (t >= ii + 1) { t -= ii + 1; ++ii; }
        const int i = ii + 1, m = t;
        double* blk = Cm + (size_t)(16 * i) * LD + 16 * m;
        doublex4 acc = (doublex4){0.0, 0.0, 0.0, 0.0};
        blk_mma(acc, Dx + i * 272, 17, 1, blk, LD, 1, lane);
        blk_store(blk, LD, 1, acc, lane);
      }
    }
    __syncthreads();
    for (int j = wave; j < nb; j += RNW) {
      for (int i = j + 1; i < nb; ++i) {
        doublex4 acc = (doublex4){0.0, 0.0, 0.0, 0.0};
        blk_mma(acc, Cm + (size_t)(16 * i) * LD + 16 * j, LD, 1, Dx + j * 272, 17, 1, lane);
        for (int m = j + 1; m < i; ++m)
          blk_mma(acc, Cm + (size_t)(16 * i) * LD + 16 * m, LD, 1, Cm + (size_t)(16 * j) * LD + 16 * m, 1, LD, lane);
        blk_store<true>(Cm + (size_t)(16 * j) * LD + 16 * i, 1, LD, acc, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    
    // (3) W_ab = sum_{i >= a} X_ia^T X_ib  (a >= b) into the lower blocks (L is dead), then mirrored
    {
      const int cnt = nb * (nb + 1) / 2;
      for (int t0 = wave; t0 < cnt; t0 += RNW) {
        int t = t0, a = 0;
        while (t >= a + 1) { t -= a + 1; ++a; }
        const int b = t;
        doublex4 acc = (doublex4){0.0, 0.0, 0.0, 0.0};
        if (a == b) blk_mma(acc, Dx + a * 272, 1, 17, Dx + a * 272, 17, 1, lane);
        else blk_mma(acc, Dx + a * 272, 1, 17, Cm + (size_t)(16 * b) * LD + 16 * a, 1, LD, lane);
        for (int i = a + 1; i < nb; ++i)
          blk_mma(acc, Cm + (size_t)(16 * a) * LD + 16 * i, LD, 1, Cm + (size_t)(16 * b) * LD + 16 * i, 1, LD, lane);
        blk_store(Cm + (size_t)(16 * a) * LD + 16 * b, LD, 1, acc, lane);
      }
    }
    __syncthreads();
    for (int e = tid; e < S16 * S16; e += RNT) {
      const int a = e / S16, b = e - a * S16;
      if (b > a && (b >> 4) != (a >> 4)) Cm[a * LD + b] = Cm[b * LD + a];
    }
  }
  __syncthreads();
}


// z = A^{-1} t from the blocked factor (spd_cholesky_blocked), by ONE wave and without barriers: block forward substitution
// y_k = X_kk (t_k - sum_{m<k} L_km y_m), block back substitution z_k = X_kk^T (y_k - sum_{m>k} L_mk^T z_m).  Lane (r = l & 15, q = l >> 4)
// takes the columns q, q+4, q+8, q+12 of every 16-wide block of row r; the four parts meet by lane permutes.  z (in / out: y, then z)
// is an LDS vector of 16*nb doubles; t is only read.  This is all the alpha != 0 path needs of C_k^{-1} (the explicit inverse --
// two more block phases, a mirror pass and a mat-vec over the whole work-group -- cost 9 us of its 79 us per iteration).
__device__ __forceinline__ void spd_solve_wave(const double* Cm, int LD, int nb, const double* Dx, const double* t, double* z, int lane) {
  const int r = lane & 15, q = lane >> 4;
  for (int k = 0; k < nb; ++k) {
    double acc = 0.0;
    for (int m = 0; m < k; ++m) {
      const double* lrow = Cm + (size_t)(16 * k + r) * LD + 16 * m;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = fma(lrow[q + 4 * j], z[16 * m + q + 4 * j], acc);
    }
    acc += __shfl_xor(acc, 16, 64); acc += __shfl_xor(acc, 32, 64);
    const double sr = t[16 * k + r] - acc;
    double y = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = q + 4 * j; y = fma(Dx[k * 272 + r * 17 + c], __shfl(sr, c, 64), y); }
    y += __shfl_xor(y, 16, 64); y += __shfl_xor(y, 32, 64);
    if (q == 0) z[16 * k + r] = y;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  for (int k = nb - 1; k >= 0; --k) {
    double acc = 0.0;
    for (int m = k + 1; m < nb; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int c = q + 4 * j; acc = fma(Cm[(size_t)(16 * m + c) * LD + 16 * k + r], z[16 * m + c], acc); }
    }
    acc += __shfl_xor(acc, 16, 64); acc += __shfl_xor(acc, 32, 64);
    const double sr = z[16 * k + r] - acc;
    __builtin_amdgcn_wave_barrier();
    double y = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = q + 4 * j; y = fma(Dx[k * 272 + c * 17 + r], __shfl(sr, c, 64), y); }
    y += __shfl_xor(y, 16, 64); y += __shfl_xor(y, 32, 64);
    if (q == 0) z[16 * k + r] = y;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int JB> struct TileCfg {
  static constexpr int NACC = JB == 8 ? 4 : 2;              // covariance blocks accumulated per pass (8 registers each)
  static constexpr int NBUF = JB == 8 ? 2 : 1;              // staging buffers (JB = 4: 62 KB of LDS per group)
  static constexpr int NPAIR = JB * (JB + 1) / 2;
  static constexpr int NPASS = (NPAIR + NACC - 1) / NACC;
  static constexpr int STAGE = NBUF * RNW * NACC * 256;     // doubles
};

// one MFMA of the covariance: block pair K (compile-time: the tile and xb are indexed by constants, i.e. stay in registers)
template <int JB, int K>
__device__ __forceinline__ void tile_cov_mfma(float (&row)[JB], const double (&xb)[JB], bool bit, doublex4& acc) {
  constexpr int NP = JB * (JB + 1) / 2, k = K < NP ? K : NP - 1;
  constexpr int bi = tri_pair_bi(k, JB), bj = tri_pair_bj(k, JB);
  const double xa = bit ? (double)row[bi] - xb[bi] : 0.0;
  const double xc = bit ? (double)row[bj] - xb[bj] : 0.0;
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xc, acc, 0, 0, 0);
}

// one covariance pass: blocks (pairs) NACC*PASS .. of C_0 over ALL chunks of the group, then the eight partial blocks through LDS
template <int JB, int PASS>
__device__ __forceinline__ void tile_cov_pass(float (&xt)[16][JB], unsigned& mbits, bool resident, int nchunk, const float* X, int pitch,
                                              int P, const unsigned char* mk, const double (&xb)[JB], int nb, double* stage,
                                              double* __restrict__ C0, int S, double invN, int tid) {
  using Cfg = TileCfg<JB>;
  constexpr int NACC = Cfg::NACC;
  if constexpr (PASS < Cfg::NPASS) {
    const int lane = tid & 63, wave = tid >> 6;
    // (a pass is skipped as a whole when S needs none of its blocks; blocks beyond S inside a pass multiply zeros: no branch per MFMA)
    constexpr int K0 = PASS * NACC, K1 = (K0 + NACC - 1 < Cfg::NPAIR) ? K0 + NACC - 1 : Cfg::NPAIR - 1;
    if (tri_pair_bi(K0, JB) >= nb || (tri_pair_bi(K0, JB) == tri_pair_bi(K1, JB) && tri_pair_bj(K0, JB) >= nb)) return;
    doublex4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = (doublex4){0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < nchunk; ++c) {
      if (!resident) mbits = tile_load<JB>(xt, X, pitch, S, P, mk, c * RNT + (tid >> 4) * 16, tid & 15, lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        rowbar<JB>(xt[i]);
        const bool bit = (mbits >> i) & 1u;
        tile_cov_mfma<JB, K0 + 0>(xt[i], xb, bit, acc[0]);
        tile_cov_mfma<JB, K0 + 1>(xt[i], xb, bit, acc[1]);
        if constexpr (NACC > 2) {
          tile_cov_mfma<JB, K0 + 2>(xt[i], xb, bit, acc[2]);
          tile_cov_mfma<JB, K0 + 3>(xt[i], xb, bit, acc[NACC > 2 ? 3 : 0]);
        }
      }
    }
    double* mine = stage + (size_t)(((PASS % Cfg::NBUF) * RNW + wave) * NACC) * 256;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[a * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[a][r];
    __syncthreads();
    const double* buf = stage + (size_t)((PASS % Cfg::NBUF) * RNW * NACC) * 256;
#pragma unroll
    for (int h = 0; h < NACC * 256 / RNT; ++h) {
      const int e = tid + h * RNT, a = e >> 8, idx = e & 255;
      int k = K0 + a, bi = 0;
      if (k < Cfg::NPAIR) {
        while (k >= JB - bi) { k -= JB - bi; ++bi; }
        const int gi = bi * 16 + (idx >> 4), gj = (bi + k) * 16 + (idx & 15);
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < RNW; ++w) t += buf[(w * NACC + a) * 256 + idx];
        if (gi < S && gj < S) { C0[(size_t)gi * S + gj] = t * invN; C0[(size_t)gj * S + gi] = t * invN; }
      }
    }
    if constexpr (Cfg::NBUF == 1) __syncthreads();
  }
}

// RES: the group's only chunk stays in the tile (JB = 8, P <= 512); launched as a pair with the streaming instantiation, each group
// is taken by exactly one of the two, decided on the device from P[g].  SHRINK: alpha != 0.
// One work-group per CU (up to 256 registers) for JB = 4 as well: at 128 registers (two groups per CU) the streaming loop kept ~30 values
// in scratch and a group took 2.5 ms instead of 1.56 ms alone on its CU -- the EMIT granule (621 groups: three rounds of one or two
// rounds of two) takes 4.9 ms either way, a shard of it (column_range, fewer groups than CUs) only the faster form.
// ENERGY: compute_energy of the reference (mag1c.py:270-275, 337-343) -- per iteration the SUM of all entries of the P x P matrix
// (x-mu) C_k^{-1} (x-mu)^T, which is s^T C_k^{-1} s with s = sum_p (x_p - mu_k) = P (d + wbar tau), d = mean over all pixels - mean over
// the statistics pixels (zero without a mask): nothing P x P is formed.  alpha = 0: from the dot products the iteration has anyway
// (+ d.p1, d.p2, and W d once);  alpha != 0: one more substitution per iteration.  Plus, once, the log-determinant term of rmf from
// the diagonal of the first factor.  Separate instantiations (streaming form only): the default kernels carry none of it.
template <int JB, bool RES, bool SHRINK, bool ENERGY = false>
__global__ __launch_bounds__(RNT, 2) void k_mag1c_tile(const Mag1cP p) {
  using Cfg = TileCfg<JB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int g = blockIdx.x;
  const int S = p.S, S16 = (S + 15) & ~15, LD = S16 | 1;    // the matrix padded to whole 16 x 16 blocks (identity beyond S), odd pitch
  const int nb = S16 >> 4;
  double* vec = reinterpret_cast<double*>(smem);
  double* xbar = vec, *tmpl = vec + VEC, *tau = vec + 2 * VEC, *mu = vec + 3 * VEC, *tnew = vec + 4 * VEC;
  double* vv = vec + 6 * VEC;
  double* p1 = vec + 8 * VEC, *p2 = vec + 9 * VEC, *p3 = vec + 10 * VEC;
  double* red = vec + 11 * VEC;      // [64]: [0,32) wave sums, [32,49) the dot products of an iteration, [60] not-PD flag, [61] nstat, [62] 1/N
  double* Cm = red + 64;             // [S16][LD]: A -> L (lower) + L^{-1} blocks (upper, transposed) -> W;  during the covariance: staging
  const size_t matsz = (size_t)S16 * LD;
  double* stg = Cm + (matsz > (size_t)Cfg::STAGE ? matsz : (size_t)Cfg::STAGE);   // [2176]: mat-vec partials | band sums of the 8 waves | X_kk
  double* Dx = stg;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool direct = RES && p.cube != nullptr;           // (uniform)
  const int P = p.P[g], pitch = direct ? 0 : p.Ppad[g];
  if (P <= 0) return;                                     // skipped group (sc_mag1c_layout_columns: too few valid pixels)
  if (direct && P > RNT) { if (threadIdx.x == 0) p.status[g] = 2; return; }      // DIRECT mode takes resident groups only
  const float* X = direct ? nullptr : reinterpret_cast<const float*>(p.x) + p.xoff[g];
  const long long po = p.poff[g];
  const unsigned char* mk = p.statmask ? p.statmask + po : nullptr;
  double* mfw = p.mfw + po; double* Rw = p.Rw + po;
  double* C0 = p.workC + (size_t)g * S * S;
  const double N = (double)P;
  constexpr bool shrink = SHRINK, resident = RES;
  static_assert(!RES || JB == 8, "the resident form exists for JB = 8 only");
  const int nchunk = (P + RNT - 1) / RNT;
  static_assert(!ENERGY || !RES, "the energy instantiations are streaming ones");
  if (!ENERGY && JB == 8 && (nchunk == 1) != RES) return;  // the other kernel of the pair takes this group (ENERGY: launched alone)
#ifdef STARCOP_MAG1C_PROF
  long long tprev = wall_clock64();
#endif

  const int bg = lane & 15;
  float xt[16][JB];                                         // [pixel q0 + i][band bg + 16*j]
  unsigned mbits = 0;
  if constexpr (RES) {
    if (direct) mbits = tile_load_direct<JB>(xt, p.cube, p.S_total, p.band0, p.pix + po, S, P, mk, (tid >> 4) * 16, bg, lane);
    else mbits = tile_load<JB>(xt, X, pitch, S, P, mk, (tid >> 4) * 16, bg, lane);
  }
  for (int s = tid; s < S; s += RNT) tmpl[s] = p.templ[s];
  if (tid == 0) { red[60] = 0.0; red[62] = 1.0 / N; }
  PROF(7);

  // ---------------- band means
  double nstat;
  {
    double aj[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) aj[j] = 0.0;
    double cnt = 0.0;
    for (int c = 0; c < nchunk; ++c) {
      if (!resident) mbits = tile_load<JB>(xt, X, pitch, S, P, mk, c * RNT + (tid >> 4) * 16, bg, lane);
      if (bg == 0) cnt += (double)__popc(mbits);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        rowbar<JB>(xt[i]);
        const double m = ((mbits >> i) & 1u) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < JB; ++j) aj[j] = fma((double)xt[i][j], m, aj[j]);
        accbar<JB>(aj);
      }
    }
    nstat = block_sum_n<RNW>(cnt, red);
    const double t = tile_band_total<JB>(aj, stg, tid);
    if (tid < S) xbar[tid] = t / nstat;
    if (tid == 0) red[61] = nstat;
  }
  __syncthreads();
  double* dvec = vec + 7 * VEC, *wdv = vec + 5 * VEC;     // (ENERGY) d = mean(all) - mean(statistics pixels);  W d | the second solution
  if constexpr (ENERGY) {
    double aj[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) aj[j] = 0.0;
    if (mk != nullptr) {
      for (int c = 0; c < nchunk; ++c) {
        mbits = tile_load<JB>(xt, X, pitch, S, P, nullptr, c * RNT + (tid >> 4) * 16, bg, lane);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          rowbar<JB>(xt[i]);
#pragma unroll
          for (int j = 0; j < JB; ++j) aj[j] += (double)xt[i][j];       // (rows beyond P hold zeros)
          accbar<JB>(aj);
        }
      }
    }
    const double t = tile_band_total<JB>(aj, stg, tid);
    if (tid < VEC) dvec[tid] = (mk != nullptr && tid < S) ? t / N - xbar[tid] : 0.0;
    __syncthreads();
  }
  PROF(8);

  // ---------------- C_0 / N -> global scratch (the LDS matrix region is the staging area meanwhile)
  {
    double xb[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) xb[j] = (bg + 16 * j < S) ? xbar[bg + 16 * j] : 0.0;
    const double invN = 1.0 / N;
#define TILE_COV(K) tile_cov_pass<JB, K>(xt, mbits, resident, nchunk, X, pitch, P, mk, xb, nb, Cm, C0, S, invN, tid)
    TILE_COV(0); TILE_COV(1); TILE_COV(2); TILE_COV(3); TILE_COV(4); TILE_COV(5); TILE_COV(6); TILE_COV(7); TILE_COV(8);
#undef TILE_COV
    static_assert(Cfg::NPASS <= 9, "tile_cov_pass calls");
  }
  __threadfence_block();
  __syncthreads();
  PROF(9);
  if constexpr (!shrink) {
    // alpha == 0: W = (C_0 / N)^{-1} once
    for (int e = tid; e < S16 * S16; e += RNT) {
      const int a = e / S16, b = e - a * S16;
      Cm[a * LD + b] = (a < S && b < S) ? C0[(size_t)a * S + b] : (a == b ? 1.0 : 0.0);
    }
    __syncthreads();
    spd_inverse_blocked(Cm, LD, nb, Dx, red + 60, tid);
    if constexpr (ENERGY) {
      // P/2 log(1 / prod diag L) = P/2 sum log diag(L^{-1}) (the diagonal blocks of L^{-1} are still in Dx);  W d and d.W d
      double lg = 0.0, dw = 0.0;
      if (tid < S16) lg = log(Dx[(tid >> 4) * 272 + (tid & 15) * 17 + (tid & 15)]);
      if (tid < S) {
        double a = 0.0;
        for (int c = 0; c < S; ++c) a = fma(Cm[tid * LD + c], dvec[c], a);
        wdv[tid] = a; dw = dvec[tid] * a;
      }
      lg = block_sum_n<RNW>(lg, red);
      dw = block_sum_n<RNW>(dw, red + 16);
      if (tid == 0) { p.logdet[g] = 0.5 * N * lg; red[50] = dw; }
      __syncthreads();
    }
  }
  PROF(10);

  // ---------------- rmf (it == 0) then the reweighted-L1 iterations; lane (pg, bg) keeps the state of pixel q0 + bg.
  double R_sel = 1.0, Rinv_sel = 1.0, mf_sel = 0.0;
  double wbar = 0.0, q = 0.0;
  const int last = p.num_iter < 0 ? 0 : p.num_iter;
  const double scale = (p.num_iter >= 0 || p.apply_scaling) ? 1e5 : 1.0;
  if (tid < S) { const double m = xbar[tid]; p1[tid] = 0.0; p2[tid] = 0.0; vv[tid] = 0.0; tau[tid] = 0.0; mu[tid] = m; tnew[tid] = tmpl[tid] * m; }
  __syncthreads();
  PROF(12);
  for (int it = 0; it <= last; ++it) {
    // the lane-dependent addresses of the loop body are rebuilt every iteration from an opaque copy of the thread index: hoisted
    // out of the loop they would be held in registers (or spilled) beside those of the tile
    int tz = tid;
    asm volatile("" : "+v"(tz));
    const int lz = tz & 63, bz = tz & 15;
    if constexpr (shrink) {
      // alpha != 0: C_k = (C_0 - v tau^T - tau v^T + q tau tau^T) / N, off-diagonal shrunk by (1 - alpha)  (mag1c.py:246-250), then
      // W = C_k^{-1}.  (C_0/N is read back from the group's global scratch: 19 KB at 49 bands, L2-resident.)
      const double invN = red[62], oma = 1.0 - p.alpha;
      // thread (row r0 = tz / 16 + 32*k, column c = tz % 16 + 16*cb): no integer division, the reads of a row's blocks in flight together
      for (int r = tz >> 4; r < S16; r += RNT / 16) {
        double c0v[JB];
#pragma unroll
        for (int cb = 0; cb < JB; ++cb) {
          const int c = (tz & 15) + 16 * cb;
          c0v[cb] = (c <= r && r < S) ? C0[(size_t)r * S + c] : 0.0;
        }
        const double vr = r < S ? vv[r] : 0.0, tr = r < S ? tau[r] : 0.0;
#pragma unroll
        for (int cb = 0; cb < JB; ++cb) {
          const int c = (tz & 15) + 16 * cb;
          if (c <= r) {
            double v = (r == c) ? 1.0 : 0.0;
            if (r < S) {
              v = c0v[cb];
              if (it > 0) v += (-vr * tau[c] - tr * vv[c] + q * tr * tau[c]) * invN;
              if (c != r) v *= oma;
            }
            Cm[r * LD + c] = v;
          }
        }
      }
      __syncthreads();
      PROF(4);
      spd_cholesky_blocked(Cm, LD, nb, Dx, red + 60, tz);
      PROF(5);
      // C_k^{-1} t and the three dot products the filter needs of it, by wave 0 (no 2 x 2 solve here: y1 = y2 = 0)
      if (wave == 0) {
        for (int s2 = lz; s2 < S16; s2 += 64) p1[s2] = s2 < S ? tnew[s2] : 0.0;          // (p1: unused in this path) t padded to whole blocks
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        spd_solve_wave(Cm, LD, nb, Dx, p1, p3, lz);
        double e0 = 0.0, e1 = 0.0, e2 = 0.0;
        for (int s2 = lz; s2 < S; s2 += 64) { const double pv = p3[s2], mv = mu[s2]; e0 = fma(tnew[s2], pv, e0); e1 = fma(mv, pv, e1); e2 = fma(mv, mv, e2); }
        e0 = wave_sum_d(e0); e1 = wave_sum_d(e1); e2 = wave_sum_d(e2);
        if (lz == 0) { red[38] = e0; red[41] = 0.0; red[39] = e1; red[42] = 0.0; red[48] = e2; }
        if constexpr (ENERGY) {
          // s^T C_k^{-1} s,  s = P (d + wbar tau): one more substitution;  first iteration: the log-determinant term
          for (int s2 = lz; s2 < S16; s2 += 64) p1[s2] = s2 < S ? dvec[s2] + wbar * tau[s2] : 0.0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          spd_solve_wave(Cm, LD, nb, Dx, p1, wdv, lz);
          double e3 = 0.0, lg = 0.0;
          for (int s2 = lz; s2 < S; s2 += 64) e3 = fma(p1[s2], wdv[s2], e3);
          if (it == 0) for (int s2 = lz; s2 < S16; s2 += 64) lg += log(Dx[(s2 >> 4) * 272 + (s2 & 15) * 17 + (s2 & 15)]);
          e3 = wave_sum_d(e3); lg = wave_sum_d(lg);
          if (lz == 0) { p.energy[(size_t)g * (last + 1) + it] = N * N * e3; if (it == 0) p.logdet[g] = 0.5 * N * lg; }
        }
      }
      __syncthreads();
    }
    if constexpr (!shrink) {
    // p1 = W v and p3 = W t_new in one pass over W: wave w takes the columns 16w .. 16w+15 for ALL rows (lane l: rows l and l + 64),
    // so its 16 + 16 vector elements are wave-uniform: one LDS read, then scalar operands from v_readlane -- the pass reads W once
    // and nothing else (125 KB per iteration; rows beyond S read padding or stale LDS, their sums are never used)
    {
      const int c0 = wave * 16;
      double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
      if (c0 < S16) {
        double uv = 0.0, ut = 0.0;
        if (lz < 16 && c0 + lz < S) { uv = vv[c0 + lz]; ut = tnew[c0 + lz]; }
        const double* w0 = Cm + lz * LD + c0;
        const double* w1 = w0 + 64 * LD;
        const bool two = S16 > 64;
#pragma unroll
        for (int h = 0; h < 16; h += 8) {                  // (8 + 8 reads in flight: 16 + 16 would not fit beside the tile)
          double m0[8], m1[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) { m0[i] = w0[h + i]; m1[i] = two ? w1[h + i] : 0.0; }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const double sv = readlane_d(uv, h + i), st = readlane_d(ut, h + i);
            a0 = fma(m0[i], sv, a0); a1 = fma(m0[i], st, a1);
            b0 = fma(m1[i], sv, b0); b1 = fma(m1[i], st, b1);
          }
        }
      }
      double* mine = stg + (tz >> 6) * 256;
      mine[lz] = a0; mine[lz + 64] = b0; mine[128 + lz] = a1; mine[128 + lz + 64] = b1;
    }
    __syncthreads();
    PROF(0);
    // waves 0-3: the sums over the 8 column blocks -> p1 (waves 0, 1), p3 (waves 2, 3) and, while the element is in a register, its
    // terms of the dot products with v, t, mu;  waves 4-7: the dot products that do not involve p1 / p3
    //   red[32 + 3*wave ..]: (v.p1, p1.t, mu.p1) x 2 waves, (t.p3, mu.p3, -) x 2 waves;  red[44..48]: v.p2  tau.p2  p2.t  mu.p2  mu.mu
    if (wave < 4) {
      const int which = tz >> 7, r = tz & 127;
      double pv = 0.0;
#pragma unroll
      for (int w = 0; w < RNW; ++w) pv += stg[w * 256 + which * 128 + r];
      double e0 = 0.0, e1 = 0.0, e2 = 0.0;
      if (r < S) {
        (which ? p3 : p1)[r] = pv;
        const double tv = tnew[r], mv = mu[r];
        if (which) { e0 = tv * pv; e1 = mv * pv; } else { e0 = vv[r] * pv; e1 = tv * pv; e2 = mv * pv; }
      }
      e0 = wave_sum_d(e0); e1 = wave_sum_d(e1); e2 = wave_sum_d(e2);
      if (lz == 0) { red[32 + 3 * wave] = e0; red[33 + 3 * wave] = e1; red[34 + 3 * wave] = e2; }
      if constexpr (ENERGY) {
        if (wave < 2) {                                   // d . p1 (rows of waves 0 and 1)
          double e3 = (r < S) ? dvec[r] * pv : 0.0;
          e3 = wave_sum_d(e3);
          if (lz == 0) red[51 + wave] = e3;
        }
      }
    } else {
      // vec slots: tau 2, mu 3, t 4, v 6, p2 9
      const int sa = wave == 4 ? 6 : wave == 5 ? 2 : wave == 6 ? 9 : 3, sb = wave == 6 ? 4 : 9;
      const double* A = vec + sa * VEC, *B = vec + sb * VEC;
      double d = 0.0, d2 = 0.0;
      for (int s = lz; s < S; s += 64) { d = fma(A[s], B[s], d); if (wave == 7) d2 = fma(A[s], A[s], d2); }
      d = wave_sum_d(d);
      if (wave == 7) d2 = wave_sum_d(d2);
      if (lz == 0) { red[40 + wave] = d; if (wave == 7) red[48] = d2; }
      if constexpr (ENERGY) {
        if (wave == 5) {                                  // d . p2
          double e3 = 0.0;
          for (int s = lz; s < S; s += 64) e3 = fma(dvec[s], p2[s], e3);
          e3 = wave_sum_d(e3);
          if (lz == 0) red[53] = e3;
        }
      }
    }
    __syncthreads();
    }
    PROF(1);
    const double dvp1 = red[32] + red[35], dp1t = red[33] + red[36], dmup1 = red[34] + red[37];
    const double dtp3 = red[38] + red[41], dmup3 = red[39] + red[42];
    const double dvp2 = red[44], dtaup2 = red[45], dp2t = red[46], dmup2 = red[47];
    double y1 = 0.0, y2 = 0.0;
    if (it > 0 && !shrink) {
      // G = M^{-1} + U^T B0 U,  M^{-1} = [[-q,-1],[-1,0]],  B0 = W/N ;  G y = U^T B0 b
      const double invN = red[62];
      const double g11 = -q + dvp1 * invN, g12 = -1.0 + dvp2 * invN, g22 = dtaup2 * invN;
      const double z1 = dp1t * invN, z2 = dp2t * invN;
      const double idet = 1.0 / (g11 * g22 - g12 * g12);
      y1 = (z1 * g22 - z2 * g12) * idet;
      y2 = (g11 * z2 - g12 * z1) * idet;
    }
    // (alpha != 0: only dtp3, dmup3 and mu.mu were written -- the other slots are stale LDS and must not enter even as 0 x value)
    if constexpr (ENERGY && !shrink) {
      if (tz == 0) {
        // s^T C_k^{-1} s = N [ s^T B0 s - b^T G^{-1} b ],  s = P (d + wbar tau),  b = U^T B0 s   (B0 = W / N, U = [v tau], G as above)
        const double dwd = red[50], dp1 = red[51] + red[52], dp2 = red[53];
        const double sBs = N * (dwd + 2.0 * wbar * dp2 + wbar * wbar * dtaup2);
        double bGb = 0.0;
        if (it > 0) {
          const double invN = red[62];
          const double g11 = -q + dvp1 * invN, g12 = -1.0 + dvp2 * invN, g22 = dtaup2 * invN;
          const double b1 = dp1 + wbar * dvp2, b2 = dp2 + wbar * dtaup2;
          bGb = (b1 * b1 * g22 - 2.0 * b1 * b2 * g12 + b2 * b2 * g11) / (g11 * g22 - g12 * g12);
        }
        p.energy[(size_t)g * (last + 1) + it] = N * (sBs - bGb);
      }
    }
    double norm = shrink ? dtp3 : dtp3 - y1 * dp1t - y2 * dp2t;                     // normaliser  t . C^{-1} t
    const double mucit = shrink ? dmup3 : dmup3 - y1 * dmup1 - y2 * dmup2;           // mu . C^{-1} t
    const double mumu = red[48];
    if (it > 0 && norm < 1.0) norm = 1.0;               // normalizer.clamp_(min=1) (mag1c.py:264-266)
    const double inorm = 1.0 / norm;
    // ONE pass over the group's chunks: per-pixel filter (C^{-1} t for the lane's JB bands, 16 x JB products, a reduce-scatter per
    // pixel row), the weight of the pixel, and its share of v = X^T w, sum w, sum w^2
    {
      const bool need_mu = (it == 0) && !p.albedo_override;
      double cj[JB];
#pragma unroll
      for (int j = 0; j < JB; ++j) { const int sj = bz + 16 * j; cj[j] = sj < S ? (shrink ? p3[sj] : p3[sj] - y1 * p1[sj] - y2 * p2[sj]) : 0.0; }
      double aj[JB];
#pragma unroll
      for (int j = 0; j < JB; ++j) aj[j] = 0.0;
      double s1 = 0.0, s2 = 0.0;
      // streaming groups: the tile of chunk c + 1 is requested IN PLACE, four rows (one float4 per band) at a time, as soon as the
      // X^T w pass is done with those rows of chunk c -- a whole tile in flight behind the arithmetic with no second register set
      // (the first chunk of an iteration is loaded here: held across the factorisation it would cost that phase 64 registers)
      double R_ld = 1.0, mf_ld = 0.0;
      if (!resident) {
        mbits = tile_load<JB>(xt, X, pitch, S, P, mk, (tz >> 4) * 16, bz, lz);
        if (it > 0 && (tz >> 4) * 16 + bz < P) { R_ld = Rw[(tz >> 4) * 16 + bz]; mf_ld = mfw[(tz >> 4) * 16 + bz]; }
      }
      for (int c = 0; c < nchunk; ++c) {
        const int q0c = c * RNT + (tz >> 4) * 16, r_q = q0c + bz;
        const bool refill = !resident && c + 1 < nchunk;
        auto request_rows = [&](int u) {               // rows 4u .. 4u+3 of the next chunk
          const int q0n = q0c + RNT;
          const bool pgok = q0n < pitch;
#pragma unroll
          for (int j = 0; j < JB; ++j) {
            const int sj = bz + 16 * j;
            float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pgok && sj < S) a4 = *reinterpret_cast<const float4*>(X + (size_t)sj * pitch + q0n + 4 * u);
            xt[4 * u][j] = a4.x; xt[4 * u + 1][j] = a4.y; xt[4 * u + 2][j] = a4.z; xt[4 * u + 3][j] = a4.w;
          }
        };
        const double dsel = tile_pixel_dots<JB>(xt, cj, bz);
        double dmu = 0.0;
        if (need_mu) {
          double cm[JB];
#pragma unroll
          for (int j = 0; j < JB; ++j) { const int sj = bz + 16 * j; cm[j] = sj < S ? mu[sj] : 0.0; }
          dmu = tile_pixel_dots<JB>(xt, cm, bz);
        }
        double w_sel = 0.0;
        if (r_q < P) {
          const double score = dsel - mucit;
          double R = R_sel, Rinv = Rinv_sel, mfp = mf_sel, mf;
          if (it == 0) {
            R = p.albedo_override ? 1.0 : dmu / mumu;
            Rinv = 1.0 / R;
            mf = score * Rinv * inorm;
            if (!p.zero_override) mf = fmax(mf, 0.0);
            if (!resident) Rw[r_q] = R;
          } else {
            if (!resident) { R = R_ld; Rinv = 1.0 / R; mfp = mf_ld; }
            const double reg = p.sparse_override ? 0.0 : Rinv / (mfp + 1e-9);
            mf = fmax((score - reg) * Rinv * inorm, 0.0);
          }
          if (resident) { R_sel = R; Rinv_sel = Rinv; mf_sel = mf; } else mfw[r_q] = mf;
          w_sel = ((mbits >> bz) & 1u) ? p.kscale * R * mf : 0.0;
          if (it == last) {
            if (RES && p.cube != nullptr) {               // DIRECT mode: straight to image order
              const long long ix = p.pix[po + r_q];
              // (f64 outputs: rounded through float first, exactly what the packed path's float store + sc_scatter_n widening gives)
              if (p.sc_f64) { reinterpret_cast<double*>(p.sc_mf)[ix] = (double)(float)(mf * scale); reinterpret_cast<double*>(p.sc_alb)[ix] = (double)(float)R; }
              else { reinterpret_cast<float*>(p.sc_mf)[ix] = (float)(mf * scale); reinterpret_cast<float*>(p.sc_alb)[ix] = (float)R; }
            } else {
              reinterpret_cast<float*>(p.mf_out)[po + r_q] = (float)(mf * scale);
              reinterpret_cast<float*>(p.alb_out)[po + r_q] = (float)R;
            }
          }
        }
        if (it != last) { s1 += w_sel; s2 += w_sel * w_sel; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (it != last) {
#pragma unroll
            for (int i = 4 * u; i < 4 * u + 4; ++i) {
              int lq = lz & 48;                            // (opaque per row: the 16 permute addresses are not worth 16 registers --
              asm volatile("" : "+v"(lq));               //  hoisted out of the chunk loop they were spilled and reloaded once per row)
              const double wi = __shfl(w_sel, lq | i, 64);
              rowbar<JB>(xt[i]);
#pragma unroll
              for (int j = 0; j < JB; ++j) aj[j] = fma((double)xt[i][j], wi, aj[j]);
              accbar<JB>(aj);
            }
          }
          if (refill) request_rows(u);
        }
        if (refill) {
          const int r_qn = r_q + RNT;
          const bool mine_ok = r_qn < P && (mk == nullptr || mk[r_qn]);
          mbits = (unsigned)(__ballot(mine_ok) >> (lz & 48)) & 0xffffu;
          if (it > 0 && r_qn < P) { R_ld = Rw[r_qn]; mf_ld = mfw[r_qn]; }
        }
      }
      PROF(2);
      if (it == last) break;
      // v = X^T w - xbar * sum(w);  the sums of w and w^2 ride on the same barrier
      s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
      if (lz == 0) { red[tz >> 6] = s1; red[16 + (tz >> 6)] = s2; }
      const double t = tile_band_total<JB>(aj, stg, tz);
      double sw = 0.0, sww = 0.0;
#pragma unroll
      for (int w = 0; w < RNW; ++w) { sw += red[w]; sww += red[16 + w]; }
      const double nst = red[61];                        // (kept in LDS: one register pair less across the whole kernel)
      wbar = sw / nst; q = sww - nst * wbar * wbar;
      if (tz < S) {
        // v, and the next iteration's tau <- t, W tau <- W t, mu = xbar - wbar tau, t = template * mu
        const double tn = tnew[tz], m = xbar[tz] - wbar * tn;
        vv[tz] = t - xbar[tz] * sw; tau[tz] = tn; p2[tz] = p3[tz]; mu[tz] = m; tnew[tz] = tmpl[tz] * m;
      }
    }
    __syncthreads();
    PROF(3);
  }
  if (tid == 0) p.status[g] = red[60] != 0.0 ? 1 : 0;
}

template <int JB>
size_t mag1c_tile_lds_bytes(int S) {
  const size_t S16 = (size_t)((S + 15) & ~15), mat = S16 * (S16 | 1), stage = (size_t)TileCfg<JB>::STAGE;
  return (11 * VEC + 64 + (mat > stage ? mat : stage) + 8 * 272) * sizeof(double);
}

#ifdef STARCOP_MAG1C_PROF
}  // namespace
extern "C" int sc_debug_mag1c_prof(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * 32); }
namespace {
#endif

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_mag1c_pack(const TI* __restrict__ cube, int S_total, int band0, int S,
                                                    const long long* __restrict__ pix, const long long* __restrict__ xoff,
                                                    const int* __restrict__ Ppad, const long long* __restrict__ poff,
                                                    const int* __restrict__ P, TO* __restrict__ xp) {
  // tile transpose through LDS: 64 pixels x up to 128 bands per step.  Read side: a WAVE per pixel (its index is one scalar load,
  // its S bands one or two coalesced row reads of the pixel-major cube; no per-element index load or division); write side: 64
  // consecutive pixels of a band per wave.  The padding [P, Ppad) of every band row is written as zeros, so the packed buffer
  // needs no memset.
  __shared__ float s_t[64][MAXS + 1];
  const int g = blockIdx.x;
  const int np = P[g], pitch = Ppad[g];
  TO* out = xp + xoff[g];
  const long long* pg = pix + poff[g];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q0 = blockIdx.y * 64; q0 < pitch; q0 += gridDim.y * 64) {
    __syncthreads();
    // the wave's 16 pixel indices in one load (lane j -> pixel wave + 4j), then all 32 row reads in flight before the first LDS store:
    // two memory round trips per tile instead of 32 dependent ones
    const int kq = q0 + wave + 4 * (lane & 15);
    const long long myidx = (lane < 16 && kq < np) ? pg[kq] : -1;
    float v[16][2];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const long long idx = __shfl(myidx, j, 64);
      const TI* row = cube + (size_t)(idx < 0 ? 0 : idx) * S_total + band0;
      v[j][0] = (idx >= 0 && lane < S) ? (float)row[lane] : 0.f;
      v[j][1] = (idx >= 0 && lane + 64 < S) ? (float)row[lane + 64] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = wave + 4 * j;
      if (lane < S) s_t[k][lane] = v[j][0];
      if (lane + 64 < S) s_t[k][lane + 64] = v[j][1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * S; i += 256) {
      const int s = i >> 6, k = i & 63;
      out[(size_t)s * pitch + q0 + k] = (q0 + k < np) ? (TO)s_t[k][s] : (TO)0;
    }
  }
}

// fp64 cubes keep full precision: direct (uncoalesced-read) gather
__global__ __launch_bounds__(256) void k_mag1c_pack_f64(const double* __restrict__ cube, int S_total, int band0, int S,
                                                        const long long* __restrict__ pix, const long long* __restrict__ xoff,
                                                        const int* __restrict__ Ppad, const long long* __restrict__ poff,
                                                        const int* __restrict__ P, double* __restrict__ xp) {
  const int g = blockIdx.x;
  const int np = P[g], pitch = Ppad[g];
  double* out = xp + xoff[g];
  const long long* pg = pix + poff[g];
  for (int i = blockIdx.y * 256 + threadIdx.x; i < pitch * S; i += gridDim.y * 256) {
    const int s = i / pitch, k = i - s * pitch;
    out[(size_t)s * pitch + k] = k < np ? cube[(size_t)pg[k] * S_total + band0 + s] : 0.0;      // padding written: no memset of the buffer
  }
}

template <typename TI, typename TO>
__global__ void k_scatter(const TI* __restrict__ val, const long long* __restrict__ pix, size_t n, TO* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[pix[i]] = (TO)val[i];
}

// valid[p] = all(cube[p][band0 .. band0+S) > nodata)  (func_by_groups' default mask, mag1c.py:140-142; NE: != fill, mag1c_emit.py:60-66:
// pixels with any band equal to the fill value are left out) in one pass over the pixel-major cube.  A block owns 256 consecutive
// pixels = one contiguous run of the cube and reads it as 16-byte vectors, eight per thread in flight; the pixel of an element comes
// from a multiply-shift (block-local element index e < 256 S_total and S_total <= 2^16 (host-checked), so e * (ceil(2^40 / S_total) - 2^40 / S_total) < 2^40 / S_total
// and floor(e * ceil(2^40 / S_total) / 2^40) is the exact quotient), a failing
// element clears its pixel's flag in LDS (benign race: every writer stores 0).  (History: per element with a 64-bit division 79 us on
// the 512 x 512 x 125 tile = 1.7 TB/s; a wave per pixel 67 us -- its 500-byte rows straddle lines and leave lanes idle on narrow cubes.)
template <typename T, bool NE>
__global__ __launch_bounds__(256) void k_valid_mask(const T* __restrict__ cube, int S_total, int band0, int S, double ref,
                                                    unsigned long long magic, long long npix, unsigned char* __restrict__ valid) {
  constexpr int VL = 16 / sizeof(T);              // elements per vector
  __shared__ int s_ok[256];
  const long long p0 = (long long)blockIdx.x * 256;
  const int np = (int)((npix - p0) < 256 ? (npix - p0) : 256);
  s_ok[threadIdx.x] = 1;
  __syncthreads();
  const T* base = cube + p0 * S_total;            // 256 * S_total elements per block: 16-byte aligned whenever the cube is
  const int n = np * S_total;
  auto check = [&](int e, T v) {
    const int q = (int)(((unsigned long long)(unsigned)e * magic) >> 40), b = e - q * S_total;
    const bool bad = NE ? ((double)v == ref) : !((double)v > ref);
    if (b >= band0 && b < band0 + S && bad) s_ok[q] = 0;
  };
  const int nv = (reinterpret_cast<uintptr_t>(base) & 15) == 0 ? n / VL : 0;      // (a cube that is not 16-byte aligned: element by element)
  typedef T vecT __attribute__((ext_vector_type(VL)));
  for (int i0 = threadIdx.x; i0 < nv; i0 += 256 * 8) {
    vecT v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u; v[u] = reinterpret_cast<const vecT*>(base)[i < nv ? i : nv - 1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u;
      if (i < nv) {
#pragma unroll
        for (int k = 0; k < VL; ++k) check(i * VL + k, v[u][k]);
      }
    }
  }
  for (int e2 = nv * VL + threadIdx.x; e2 < n; e2 += 256) check(e2, base[e2]);
  __syncthreads();
  if ((int)threadIdx.x < np) valid[p0 + threadIdx.x] = (unsigned char)s_ok[threadIdx.x];
}

// ---- packed layout of column-structured groups, built on the device without a sort -----------------------------------------
// group g = image columns [gcol[g], gcol[g+1]); its pixels are the valid ones in row-major order.
__device__ __forceinline__ int lc_row_count(const unsigned char* __restrict__ valid, int r, int cols, int c0, int c1) {
  int n = 0;
  for (int c = c0; c < c1; ++c) n += valid[(size_t)r * cols + c] ? 1 : 0;
  return n;
}
// inclusive scan of one int per thread over a 256-thread block; returns the block total through `total`
__device__ __forceinline__ int lc_block_scan(int v, int* s_w, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
  if (lane == 63) s_w[wave] = v;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_w[w];
  total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  __syncthreads();
  return v + base;
}
__global__ __launch_bounds__(256) void k_lc_count(const unsigned char* __restrict__ valid, int rows, int cols,
                                                  const int* __restrict__ gcol, int min_keep, int* __restrict__ P) {
  __shared__ int s_w[4];
  const int g = blockIdx.x, c0 = gcol[g], c1 = gcol[g + 1];
  int n = 0;
  for (int r = threadIdx.x; r < rows; r += 256) n += lc_row_count(valid, r, cols, c0, c1);
  int total;
  (void)lc_block_scan(n, s_w, total);
  if (threadIdx.x == 0) P[g] = total > min_keep ? total : 0;
}
__global__ __launch_bounds__(256) void k_lc_scan(const int* __restrict__ P, int G, int S, int* __restrict__ Ppad,
                                                 long long* __restrict__ poff, long long* __restrict__ xoff,
                                                 long long* __restrict__ totals) {
  __shared__ int s_w[4];
  long long carry_p = 0, carry_x = 0;
  for (int g0 = 0; g0 < G; g0 += 256) {
    const int g = g0 + threadIdx.x;
    const int pv = g < G ? P[g] : 0;
    const int pp = (pv + 63) & ~63;
    int tp, tq;
    const int ip = lc_block_scan(pv, s_w, tp);
    const int iq = lc_block_scan(pp, s_w, tq);
    if (g < G) {
      Ppad[g] = pp;
      poff[g] = carry_p + (ip - pv);
      xoff[g] = (carry_x + (iq - pp)) * (long long)S;
    }
    carry_p += tp; carry_x += tq;
  }
  if (threadIdx.x == 0) { totals[0] = carry_p; totals[1] = carry_x * (long long)S; }
}
__global__ __launch_bounds__(256) void k_lc_index(const unsigned char* __restrict__ valid, int rows, int cols,
                                                  const int* __restrict__ gcol, const int* __restrict__ P,
                                                  const long long* __restrict__ poff, long long* __restrict__ pix) {
  __shared__ int s_w[4];
  const int g = blockIdx.x;
  if (P[g] <= 0) return;
  const int c0 = gcol[g], c1 = gcol[g + 1];
  long long* out = pix + poff[g];
  int carry = 0;
  for (int r0 = 0; r0 < rows; r0 += 256) {
    const int r = r0 + threadIdx.x;
    const int n = r < rows ? lc_row_count(valid, r, cols, c0, c1) : 0;
    int total;
    int pos = carry + lc_block_scan(n, s_w, total) - n;
    if (r < rows)
      for (int c = c0; c < c1; ++c)
        if (valid[(size_t)r * cols + c]) out[pos++] = (long long)r * cols + c;
    carry += total;
  }
}
// ---- layout of ARBITRARY integer groups by a stable counting sort (the orthorectified AVIRIS-NG case: groups = |GLT sample index|,
// process_aviris.py:211-217, bounded by the detector width).  Group g = id value g; pixels of a group in ascending pixel index
// (the order of the reference's boolean indexing and of a stable sort).  Blocks of LI_BLK consecutive pixels, one wave each.
constexpr int LI_BLK = 1024;
// cnt[b][id] = valid pixels with that id in block b (a block is one wave walking its pixels in order: plain increments)
__global__ __launch_bounds__(64) void k_li_count(const unsigned char* __restrict__ valid, const int* __restrict__ ids, long long npix,
                                                 int nids, int* __restrict__ cnt) {
  const long long p0 = (long long)blockIdx.x * LI_BLK;
  int* row = cnt + (size_t)blockIdx.x * nids;
  for (int c = 0; c < LI_BLK; c += 64) {
    const long long p = p0 + c + threadIdx.x;
    const bool ok = p < npix && valid[p];
    const int id = ok ? ids[p] : -1;
    if (id >= 0 && id < nids) atomicAdd(row + id, 1);         // one wave owns the row: order-independent integer sums
  }
}
// per id: exclusive scan of cnt over the blocks (in place) and the group's total
__global__ __launch_bounds__(256) void k_li_colscan(int* __restrict__ cnt, int nblk, int nids, int min_keep, int* __restrict__ P) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= nids) return;
  int run = 0;
  for (int b = 0; b < nblk; ++b) {
    const int v = cnt[(size_t)b * nids + id];
    cnt[(size_t)b * nids + id] = run;
    run += v;
  }
  P[id] = run > min_keep ? run : 0;
}
// pix[poff[id] + (pixels of id in earlier blocks) + (rank inside the block)] = pixel
__global__ __launch_bounds__(64) void k_li_index(const unsigned char* __restrict__ valid, const int* __restrict__ ids, long long npix,
                                                 int nids, int* __restrict__ cnt, const int* __restrict__ P,
                                                 const long long* __restrict__ poff, long long* __restrict__ pix) {
  const long long p0 = (long long)blockIdx.x * LI_BLK;
  int* row = cnt + (size_t)blockIdx.x * nids;                  // running position of every id inside this block's range
  const int lane = threadIdx.x;
  for (int c = 0; c < LI_BLK; c += 64) {
    const long long p = p0 + c + lane;
    const bool ok = p < npix && valid[p];
    int id = ok ? ids[p] : -1;
    if (id >= nids) id = -1;
    // rank among the lower lanes of this chunk with the same id, and how many lanes of the chunk share it
    int rank = 0, same = 0;
    for (int j = 0; j < 64; ++j) {
      const int idj = __shfl(id, j, 64);
      if (idj == id) { same++; if (j < lane) rank++; }
    }
    if (id >= 0) {
      const int base = row[id];
      if (P[id] > 0) pix[poff[id] + base + rank] = p;
      if (rank == same - 1) row[id] = base + same;             // the last lane of the id advances the running position
    }
  }
}
template <typename TI, typename TO>
__global__ void k_scatter_n(const TI* __restrict__ val, const long long* __restrict__ pix, const long long* __restrict__ n_dev,
                            TO* __restrict__ out) {
  const size_t n = (size_t)*n_dev;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[pix[i]] = (TO)val[i];
}

size_t mag1c_lds_bytes(int S) {
  const int S16 = (S + 15) & ~15;
  const size_t stg = (size_t)S16 * 17 > 8 * 128 ? (size_t)S16 * 17 : 8 * 128;
  return ((size_t)S * (S + 1) + 11 * VEC + 64 + stg) * sizeof(double);
}

// the dynamic-LDS limit of a kernel: set once per instantiation (the pointer identifies it); not a stream operation
template <typename K>
hipError_t mag1c_lds_attr(K kern, size_t lds) {
  static const void* done[32];
  static int ndone = 0;
  const void* f = reinterpret_cast<const void*>(kern);
  for (int i = 0; i < ndone; ++i) if (done[i] == f) return hipSuccess;
  (void)lds;
  const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess && ndone < 32) done[ndone++] = f;
  return e;
}

}  // namespace

extern "C" size_t sc_mag1c_workspace_doubles(int G, int S, int64_t npix) {
  return (size_t)G * S * S + 3 * (size_t)npix;
}

extern "C" int sc_mag1c_groups(const sc_mag1c_args* a, sc_stream stream) {
  SC_REQUIRE(a != nullptr, "sc_mag1c_groups: null args");
  SC_REQUIRE(a->S >= 1 && a->S <= MAXS, "sc_mag1c_groups: number of bands must be in [1,%d] (got %d)", MAXS, a->S);
  SC_REQUIRE(a->G >= 0 && a->npix >= 0, "sc_mag1c_groups: bad group / pixel count");
  const bool direct = a->cube != nullptr;
  if (direct) {
    SC_REQUIRE(a->P && a->poff && a->templ && a->work && a->status && a->pix_index && a->scatter_mf && a->scatter_alb,
               "sc_mag1c_groups: DIRECT mode: null pointer argument");
    SC_REQUIRE(!a->x_is_f64 && a->S > 64 && a->energy == nullptr && a->band0 >= 0 && a->band0 + a->S <= a->S_total,
               "sc_mag1c_groups: DIRECT mode takes fp32 cubes, 65..128 bands inside the cube's bands, no compute_energy");
  } else
  SC_REQUIRE(a->x && a->xoff && a->P && a->Ppad && a->poff && a->templ && a->work && a->mf_out && a->albedo_out && a->status,
             "sc_mag1c_groups: null pointer argument");
  if (a->G == 0) return SC_OK;
  Mag1cP p;
  p.cube = a->cube; p.S_total = a->S_total; p.band0 = a->band0; p.pix = (const long long*)a->pix_index;
  p.sc_mf = a->scatter_mf; p.sc_alb = a->scatter_alb; p.sc_f64 = a->scatter_is_f64;
  p.x = a->x; p.xoff = (const long long*)a->xoff; p.P = a->P; p.Ppad = a->Ppad; p.poff = (const long long*)a->poff;
  p.statmask = a->statmask; p.G = a->G; p.S = a->S; p.npix = a->npix; p.templ = a->templ; p.num_iter = a->num_iter;
  p.alpha = a->alpha; p.kscale = a->cov_update_scaling;
  p.albedo_override = a->albedo_override; p.zero_override = a->zero_override; p.sparse_override = a->sparse_override;
  p.apply_scaling = a->apply_scaling;
  p.workC = a->work;
  p.mfw = a->work + (size_t)a->G * a->S * a->S; p.Rw = p.mfw + a->npix; p.wv = p.Rw + a->npix;
  p.mf_out = a->mf_out; p.alb_out = a->albedo_out; p.status = a->status;
  p.energy = a->energy; p.logdet = a->logdet;
  SC_REQUIRE((a->energy == nullptr) == (a->logdet == nullptr), "sc_mag1c_groups: energy and logdet go together");
  size_t lds = mag1c_lds_bytes(a->S);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  const bool fast = a->alpha == 0.0;     // no shrinkage: one factorisation per group + Woodbury updates
  if (!a->x_is_f64) {
    // fp32 radiances (both drivers of the reference): the register-tile kernel, one group per CU.  Up to 64 bands: chunks streamed;
    // more: a pair of launches (groups of <= 512 pixels stay in registers, larger ones stream), see k_mag1c_tile
#define SC_TILE_GO(...)                                                                                                        \
    do {                                                                                                                       \
      if (e == hipSuccess) e = mag1c_lds_attr(&k_mag1c_tile<__VA_ARGS__>, lds);                                                  \
      if (e == hipSuccess) hipLaunchKernelGGL((k_mag1c_tile<__VA_ARGS__>), dim3(a->G), dim3(RNT), lds, st, p);                  \
    } while (0)
    e = hipSuccess;
    if (a->energy) {                 // compute_energy: the streaming instantiations take every group
      if (a->S <= 64) { lds = mag1c_tile_lds_bytes<4>(a->S); if (fast) SC_TILE_GO(4, false, false, true); else SC_TILE_GO(4, false, true, true); }
      else { lds = mag1c_tile_lds_bytes<8>(a->S); if (fast) SC_TILE_GO(8, false, false, true); else SC_TILE_GO(8, false, true, true); }
    } else if (a->S <= 64) {
      lds = mag1c_tile_lds_bytes<4>(a->S);
      if (fast) SC_TILE_GO(4, false, false); else SC_TILE_GO(4, false, true);
    } else {
      lds = mag1c_tile_lds_bytes<8>(a->S);
      // (DIRECT mode: every group is resident by the caller's promise -- a larger one gets status 2 -- so the streaming launch is dropped)
      if (fast) { SC_TILE_GO(8, true, false); if (!direct) SC_TILE_GO(8, false, false); }
      else { SC_TILE_GO(8, true, true); if (!direct) SC_TILE_GO(8, false, true); }
    }
#undef SC_TILE_GO
  } else if (fast) {
    e = mag1c_lds_attr(&k_mag1c_fast<double, 1024>, lds);
    if (e == hipSuccess) hipLaunchKernelGGL((k_mag1c_fast<double, 1024>), dim3(a->G), dim3(1024), lds, st, p);
  } else {
    // fp64 radiances, alpha != 0: refactorisation every iteration.  Few bands: 512 threads (3 groups per CU);
    // many bands: the matrix fills the LDS, one group of 1024 threads per CU
#define SC_MAG1C_GO(T_, NT_, EN_)                                                                                                 \
    do {                                                                                                                       \
      e = mag1c_lds_attr(&k_mag1c<T_, NT_, EN_>, lds);                                                                          \
      if (e == hipSuccess) hipLaunchKernelGGL((k_mag1c<T_, NT_, EN_>), dim3(a->G), dim3(NT_), lds, st, p);                     \
    } while (0)
    if (a->energy) { if (a->S <= 64) SC_MAG1C_GO(double, 512, true); else SC_MAG1C_GO(double, 1024, true); }
    else { if (a->S <= 64) SC_MAG1C_GO(double, 512, false); else SC_MAG1C_GO(double, 1024, false); }
#undef SC_MAG1C_GO
  }
  if (e != hipSuccess) { sc_set_error("sc_mag1c_groups: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return SC_ERR_LAUNCH; }
  SC_LAUNCH_OK("sc_mag1c_groups");
  return SC_OK;
}

extern "C" int sc_mag1c_pack(const void* cube, int cube_is_f64, int S_total, int band0, int S, const int64_t* pix_index,
                             const int64_t* xoff, const int32_t* Ppad, const int64_t* poff, const int32_t* P, int G,
                             void* xpacked, int out_is_f64, sc_stream stream) {
  SC_REQUIRE(cube && pix_index && xoff && Ppad && poff && P && xpacked, "sc_mag1c_pack: null pointer argument");
  SC_REQUIRE(S >= 1 && S <= MAXS && band0 >= 0 && band0 + S <= S_total, "sc_mag1c_pack: bad band range");
  if (G == 0) return SC_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(G, 8);
  const long long* px = (const long long*)pix_index; const long long* xo = (const long long*)xoff; const long long* po = (const long long*)poff;
  if (cube_is_f64) {
    SC_REQUIRE(out_is_f64, "sc_mag1c_pack: fp64 cube needs an fp64 packed buffer");
    hipLaunchKernelGGL(k_mag1c_pack_f64, grid, dim3(256), 0, st, (const double*)cube, S_total, band0, S, px, xo, Ppad, po, P, (double*)xpacked);
  } else if (out_is_f64) {
    hipLaunchKernelGGL((k_mag1c_pack<float, double>), grid, dim3(256), 0, st, (const float*)cube, S_total, band0, S, px, xo, Ppad, po, P, (double*)xpacked);
  } else {
    hipLaunchKernelGGL((k_mag1c_pack<float, float>), grid, dim3(256), 0, st, (const float*)cube, S_total, band0, S, px, xo, Ppad, po, P, (float*)xpacked);
  }
  SC_LAUNCH_OK("sc_mag1c_pack");
  return SC_OK;
}

extern "C" int sc_scatter(const void* val, int val_is_f64, const int64_t* pix_index, size_t n, void* out, int out_is_f64,
                          sc_stream stream) {
  SC_REQUIRE(val && pix_index && out, "sc_scatter: null pointer argument");
  if (n == 0) return SC_OK;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  const long long* px = (const long long*)pix_index;
  if (val_is_f64 && out_is_f64) hipLaunchKernelGGL((k_scatter<double, double>), dim3(blocks), dim3(256), 0, st, (const double*)val, px, n, (double*)out);
  else if (val_is_f64) hipLaunchKernelGGL((k_scatter<double, float>), dim3(blocks), dim3(256), 0, st, (const double*)val, px, n, (float*)out);
  else if (out_is_f64) hipLaunchKernelGGL((k_scatter<float, double>), dim3(blocks), dim3(256), 0, st, (const float*)val, px, n, (double*)out);
  else hipLaunchKernelGGL((k_scatter<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)val, px, n, (float*)out);
  SC_LAUNCH_OK("sc_scatter");
  return SC_OK;
}

extern "C" int sc_valid_mask(const void* cube, int cube_is_f64, int S_total, int band0, int S, double nodata, int64_t npix,
                             unsigned char* valid, sc_stream stream) {
  SC_REQUIRE(cube && valid && S_total > 0 && band0 >= 0 && S > 0 && band0 + S <= S_total && npix >= 0, "sc_valid_mask: bad argument");
  SC_REQUIRE(S_total <= (1 << 16), "sc_valid_mask: at most 65536 bands per pixel");      // (e < 256 S_total: the multiply-shift quotient is exact only while S_total^2 * 256 <= 2^40)
  if (npix == 0) return SC_OK;
  const unsigned blocks = (unsigned)((npix + 255) / 256);
  const unsigned long long magic = ((1ull << 40) + S_total - 1) / S_total;
  if (cube_is_f64) hipLaunchKernelGGL((k_valid_mask<double, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)cube, S_total, band0, S, nodata, magic, (long long)npix, valid);
  else hipLaunchKernelGGL((k_valid_mask<float, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)cube, S_total, band0, S, nodata, magic, (long long)npix, valid);
  SC_LAUNCH_OK("sc_valid_mask");
  return SC_OK;
}

extern "C" int sc_valid_mask_ne(const void* cube, int cube_is_f64, int S_total, int band0, int S, double fill, int64_t npix,
                                unsigned char* valid, sc_stream stream) {
  SC_REQUIRE(cube && valid && S_total > 0 && band0 >= 0 && S > 0 && band0 + S <= S_total && npix >= 0, "sc_valid_mask_ne: bad argument");
  SC_REQUIRE(S_total <= (1 << 16), "sc_valid_mask_ne: at most 65536 bands per pixel");
  if (npix == 0) return SC_OK;
  const unsigned blocks = (unsigned)((npix + 255) / 256);
  const unsigned long long magic = ((1ull << 40) + S_total - 1) / S_total;
  if (cube_is_f64) hipLaunchKernelGGL((k_valid_mask<double, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)cube, S_total, band0, S, fill, magic, (long long)npix, valid);
  else hipLaunchKernelGGL((k_valid_mask<float, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)cube, S_total, band0, S, fill, magic, (long long)npix, valid);
  SC_LAUNCH_OK("sc_valid_mask_ne");
  return SC_OK;
}

extern "C" int sc_mag1c_layout_columns(const unsigned char* valid, int rows, int cols, const int32_t* gcol, int G, int S,
                                       int min_keep, int32_t* P, int32_t* Ppad, int64_t* poff, int64_t* xoff,
                                       int64_t* pix_index, int64_t* totals, sc_stream stream) {
  SC_REQUIRE(valid && gcol && P && Ppad && poff && xoff && pix_index && totals, "sc_mag1c_layout_columns: null pointer argument");
  SC_REQUIRE(rows > 0 && cols > 0 && G >= 0 && S >= 1 && S <= MAXS, "sc_mag1c_layout_columns: bad shape");
  if (G == 0) return SC_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_lc_count, dim3(G), dim3(256), 0, st, valid, rows, cols, gcol, min_keep, P);
  hipLaunchKernelGGL(k_lc_scan, dim3(1), dim3(256), 0, st, (const int*)P, G, S, Ppad, (long long*)poff, (long long*)xoff, (long long*)totals);
  hipLaunchKernelGGL(k_lc_index, dim3(G), dim3(256), 0, st, valid, rows, cols, gcol, (const int*)P, (const long long*)poff, (long long*)pix_index);
  SC_LAUNCH_OK("sc_mag1c_layout_columns");
  return SC_OK;
}

extern "C" size_t sc_mag1c_layout_ids_workspace_ints(int64_t npix, int nids) {
  return (size_t)((npix + LI_BLK - 1) / LI_BLK) * (size_t)nids;
}

extern "C" int sc_mag1c_layout_ids(const unsigned char* valid, const int32_t* ids, int64_t npix, int nids, int S, int min_keep,
                                   int32_t* P, int32_t* Ppad, int64_t* poff, int64_t* xoff, int64_t* pix_index, int64_t* totals,
                                   int32_t* work, sc_stream stream) {
  SC_REQUIRE(valid && ids && P && Ppad && poff && xoff && pix_index && totals && work, "sc_mag1c_layout_ids: null pointer argument");
  SC_REQUIRE(npix > 0 && nids > 0 && S > 0, "sc_mag1c_layout_ids: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)((npix + LI_BLK - 1) / LI_BLK);
  if (hipMemsetAsync(work, 0, (size_t)nblk * nids * sizeof(int), st) != hipSuccess) { sc_set_error("sc_mag1c_layout_ids: memset failed"); return SC_ERR_LAUNCH; }
  hipLaunchKernelGGL(k_li_count, dim3(nblk), dim3(64), 0, st, valid, ids, (long long)npix, nids, work);
  hipLaunchKernelGGL(k_li_colscan, dim3((nids + 255) / 256), dim3(256), 0, st, work, nblk, nids, min_keep, P);
  hipLaunchKernelGGL(k_lc_scan, dim3(1), dim3(256), 0, st, (const int*)P, nids, S, Ppad, (long long*)poff, (long long*)xoff, (long long*)totals);
  hipLaunchKernelGGL(k_li_index, dim3(nblk), dim3(64), 0, st, valid, ids, (long long)npix, nids, work, (const int*)P,
                     (const long long*)poff, (long long*)pix_index);
  SC_LAUNCH_OK("sc_mag1c_layout_ids");
  return SC_OK;
}

extern "C" int sc_scatter_n(const void* val, int val_is_f64, const int64_t* pix_index, const int64_t* n_dev, size_t n_max,
                            void* out, int out_is_f64, sc_stream stream) {
  SC_REQUIRE(val && pix_index && out && n_dev, "sc_scatter_n: null pointer argument");
  if (n_max == 0) return SC_OK;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((n_max + 255) / 256 > 2048 ? 2048 : (n_max + 255) / 256);
  const long long* px = (const long long*)pix_index; const long long* nd = (const long long*)n_dev;
  if (val_is_f64 && out_is_f64) hipLaunchKernelGGL((k_scatter_n<double, double>), dim3(blocks), dim3(256), 0, st, (const double*)val, px, nd, (double*)out);
  else if (val_is_f64) hipLaunchKernelGGL((k_scatter_n<double, float>), dim3(blocks), dim3(256), 0, st, (const double*)val, px, nd, (float*)out);
  else if (out_is_f64) hipLaunchKernelGGL((k_scatter_n<float, double>), dim3(blocks), dim3(256), 0, st, (const float*)val, px, nd, (double*)out);
  else hipLaunchKernelGGL((k_scatter_n<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)val, px, nd, (float*)out);
  SC_LAUNCH_OK("sc_scatter_n");
  return SC_OK;
}
