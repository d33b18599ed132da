// Shared device/host helpers for libstarcop_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "starcop_hip.h"

void sc_set_error(const char* fmt, ...);

#define SC_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      sc_set_error(__VA_ARGS__);              \
      return SC_ERR_ARG;                      \
    }                                         \
  } while (0)

#define SC_LAUNCH_OK(name)                                                        \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      sc_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return SC_ERR_LAUNCH;                                                       \
    }                                                                             \
  } while (0)

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// device-side view of sc_src (passed by value inside kernel-argument structs)
struct SrcD {
  const float* x;
  const float* aux;
  const float* cst;
  int C, mode, act, up;
};

static inline SrcD to_srcd(const sc_src& s) {
  SrcD d;
  d.x = s.x; d.aux = s.aux; d.cst = s.cst; d.C = s.C; d.mode = s.mode; d.act = s.act; d.up = s.up;
  return d;
}
static inline SrcD empty_srcd() {
  SrcD d;
  d.x = nullptr; d.aux = nullptr; d.cst = nullptr; d.C = 0; d.mode = 0; d.act = 0; d.up = 0;
  return d;
}

__device__ __forceinline__ float sc_actf(float v, int act) {
  if (act == SC_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SC_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}

// "normalise on load": see enum sc_src_mode in starcop_hip.h
__device__ __forceinline__ float sc_prologue(int mode, int act, float x, float aux, float4 c0, float c4) {
  if (mode == SC_SRC_RAW) return x;
  if (mode == SC_SRC_AFFINE) return sc_actf(fmaf(x, c0.x, c0.y), act);
  if (mode == SC_SRC_BNBWD) {
    const float yh = fmaf(aux, c0.x, c0.y);
    bool pass = true;
    if (act == SC_ACT_RELU) pass = yh > 0.f;
    else if (act == SC_ACT_RELU6) pass = (yh > 0.f) && (yh < 6.f);
    const float gm = pass ? x : 0.f;
    return fmaf(gm, c0.z, fmaf(aux, c0.w, c4));
  }
  // SC_SRC_NORM: clamp((x - off) / fac, lo, hi)   (true division, as the reference does)
  const float v = (x - c0.x) / c0.y;
  return fminf(fmaxf(v, c0.z), c0.w);
}

// ---- branch-free prologues for the hot kernels (mode is uniform per source; NORM is only used by the stem) ----
// act limits: none -> (-inf, +inf), relu -> (0, +inf), relu6 -> (0, 6)
__device__ __forceinline__ float sc_act_lo(int act) { return act == SC_ACT_NONE ? -__builtin_inff() : 0.f; }
__device__ __forceinline__ float sc_act_hi(int act) { return act == SC_ACT_RELU6 ? 6.f : __builtin_inff(); }
// SC_SRC_RAW is the affine form with scale 1, shift 0, no limits
__device__ __forceinline__ float sc_pro_affine(float x, float sc, float sh, float lo, float hi) {
  return fminf(fmaxf(fmaf(x, sc, sh), lo), hi);
}
__device__ __forceinline__ float sc_pro_bnbwd(float g, float y, float sc, float sh, float A, float B, float D, float lo, float hi) {
  const float yh = fmaf(y, sc, sh);
  const float gm = (yh > lo && yh < hi) ? g : 0.f;
  return fmaf(gm, A, fmaf(y, B, D));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 32 lanes of each half-wave with DPP adds (one VALU instruction each, no LDS crossbar traffic):
// quad_perm, quad_perm, row_half_mirror, row_mirror -> 16-lane row sums in every lane; row_bcast15 into rows 1 and 3
// -> lanes 16..31 hold the sum of lanes 0..31, lanes 48..63 the sum of lanes 32..63.  Read the result at (lane&31)==16.
// (bound_ctrl for the full-mask permutations: every lane has a valid source there, and with it the compiler folds the DPP move into
// the add -- v_add_f32_dpp -- for all of them instead of for the quad permutations only)
#define SC_DPP_ADD(v, ctrl, rmask) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, (rmask) == 0xF))
__device__ __forceinline__ float half_sum32(float v) {
  SC_DPP_ADD(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
  SC_DPP_ADD(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
  SC_DPP_ADD(v, 0x141, 0xF);   // row_half_mirror
  SC_DPP_ADD(v, 0x140, 0xF);   // row_mirror
  SC_DPP_ADD(v, 0x142, 0xA);   // row_bcast15 -> rows 1, 3
  return v;
}
constexpr int SC_HALF_SUM_LANE = 16;
// the first four steps only: every lane holds the sum of its 16-lane row.  Callers that keep one partial per ROW (two per
// half-wave) save the row_bcast step, the one DPP step the compiler cannot fold into its add (row mask => v_mov_b32_dpp + add).
__device__ __forceinline__ float row_sum16(float v) {
  SC_DPP_ADD(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
  SC_DPP_ADD(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
  SC_DPP_ADD(v, 0x141, 0xF);   // row_half_mirror
  SC_DPP_ADD(v, 0x140, 0xF);   // row_mirror
  return v;
}

// sums nparts rows of E floats (part[k][i]) into out[i]; uses `scratch` (>= sc_reduce_scratch_floats)
// for the intermediate levels.  Defined in elementwise.hip.
size_t sc_reduce_scratch_floats(int nparts, size_t E);
int sc_reduce_rows(const float* part, int nparts, size_t E, float* scratch, float* out, hipStream_t st);
// like sc_reduce_rows but leaves <= max_rows rows for a caller-side final kernel: returns the pointer/row count
int sc_reduce_rows_partial(const float* part, int nparts, size_t E, float* scratch, const float** rows_out,
                           int* nrows_out, int max_rows, hipStream_t st);

// weight-gradient epilogue shared by the MFMA wgrad kernels (conv_mfma.hip): sums part[nparts][taps][CoP][CiP]
// (reduction scratch of sc_reduce_scratch_floats(nparts, taps*CoP*CiP) floats follows the partials) into dw (OIHW)
int sc_wgrad_finish(float* part, int nparts, int taps, int Cout, int Cin, int CoP, int CiP, float* dw, hipStream_t st);

// device table [C][SC_CST] of identity constants (scale 1, rest 0) for RAW sources; NULL if unavailable (conv_mfma.hip)
const float* sc_identity_cst_table(int C);

// row of this work-group in a [rows][C][2] partial-statistics buffer (grid = (tiles, channel-tiles, N))
// depthwise tile shape (outputs per block) by plane width; shared by the launchers and sc_stat_rows
inline int sc_dw_tile_w(int W) { return W > 32 ? 64 : (W > 16 ? 32 : 16); }
inline int sc_dw_tile_h(int W) { return W > 16 ? 32 : 16; }
__device__ __forceinline__ size_t stat_row() { return (size_t)blockIdx.z * gridDim.x + blockIdx.x; }
