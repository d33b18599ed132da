// A whole MobileNetV2 inverted-residual block in ONE launch -- inference (eval-mode BatchNorm) only.
//   expand 1x1 (Cin -> hidden) -> BN + ReLU6 -> depthwise 3x3 (stride 1 | 2, pad 1) -> BN + ReLU6 -> project 1x1 (hidden -> Cout)
// (torchvision InvertedResidual.conv as smp.Unet('mobilenet_v2') runs it: /root/reference/starcop/models/model_module.py:244-251; the
// notebook / validation inference path, model_module.py:90-98 under torch.no_grad()).  With running statistics the BatchNorm constants
// are known before the launch, so nothing forces the 6x-expanded tensors through HBM: a work-group owns an output tile, stages the
// block input (tile + halo, every channel) once into LDS in MFMA-operand form, and walks the hidden channels in chunks of 32:
//   (b) e[32][in px]  = W_e[32 x Cin] * x            v_mfma_f32_32x32x16_bf16, three exact bf16 terms per operand, six products
//       -> BN + ReLU6, ZERO outside the image (the depthwise conv pads its activated input) -> LDS fp32
//   (c) d[32][out px] = 3x3 stencil of e              VALU, thread = (pixel, 8 channels) -> BN + ReLU6 -> three bf16 terms -> LDS
//   (d) p[Cout][out px] += W_p[Cout x 32] * d         same MFMA; accumulators persist over the chunks
// and finally stores the RAW projection output (its BatchNorm is applied by the consumer on load, as everywhere in this network).
// The filter operands are the SC_PACK_PW3 packs the pointwise kernels use (lane -> matrix row/col l&31, k = 8*(l>>5)..+7 serves as
// the A operand unchanged).  Replaces three convolution launches (+ the e / d round trips: 2 x 25-38 MB per block at 32^2) by one.
#include "sc_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) unsigned int uintx4;

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
__device__ __forceinline__ void split8(const float (&v)[8], uintx4 (&t)[3]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned t0, t1, t2;
    split3x2(v[2 * q], v[2 * q + 1], t0, t1, t2);
    t[0][q] = t0; t[1][q] = t1; t[2][q] = t2;
  }
}
__device__ __forceinline__ floatx16 mfma_bf16(const uintx4& a, const uintx4& b, const floatx16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ floatx16 mfma6(const uintx4 (&a)[3], const uintx4 (&b)[3], floatx16 c) {
  c = mfma_bf16(a[1], b[1], c);
  c = mfma_bf16(a[2], b[0], c);
  c = mfma_bf16(a[0], b[2], c);
  c = mfma_bf16(a[1], b[0], c);
  c = mfma_bf16(a[0], b[1], c);
  c = mfma_bf16(a[0], b[0], c);
  return c;
}

struct IrP {
  SrcD x;                      // block input: RAW (residual sum) or AFFINE (BatchNorm'd projection of the previous block)
  const uintx4* we;            // SC_PACK_PW3 pack of the expansion filter  (M = hidden, K = Cin)
  const uintx4* wp;            // SC_PACK_PW3 pack of the projection filter (M = Cout,   K = hidden)
  const float* wd;             // depthwise filter [hidden][9]
  const float* cst_e;          // [hidden][SC_CST] eval constants (scale, shift) of the expansion's BatchNorm
  const float* cst_d;          // ... of the depthwise conv's BatchNorm
  float* out;                  // raw projection output [N][Cout][Ho][Wo]
  int N, Cin, Hd, Cout, H, W, Ho, Wo;
  int nks_e, nchunk, nks_p;    // Cin/16 (rounded up), hidden/32 (rounded up), hidden/16 (rounded up)
  int tiles_x, tiles_y;
};

constexpr int IR_MAXP = 5;     // (cout block, pixel block) pairs per wave: Cout <= 320 at 64 output pixels

// S: stride.  Output tile: 8 x 8 (stride 1), 4 x 8 (stride 2); input tile (TH*S + 2) x (TW*S + 2)
template <int S>
__global__ __launch_bounds__(256) void k_ir_eval(const IrP p) {
  constexpr int TH = S == 1 ? 8 : 4, TW = 8;
  constexpr int OUT_PX = TH * TW;                 // 64 | 32
  constexpr int OUT_PXB = OUT_PX / 32;            // 2 | 1
  constexpr int IH = TH * S + (S == 1 ? 2 : 1), IW = TW * S + (S == 1 ? 2 : 1);      // 10 x 10 | 9 x 17 (rows 2*oy .. 2*oy + 2)
  constexpr int IN_PX = IH * IW;                  // 100 | 153
  constexpr int IN_PXB = (IN_PX + 31) / 32;       // 4 | 5
  constexpr int IN_PAD = IN_PXB * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  // layout: s_x [3][KG][IN_PAD] uintx4 | s_e [32][IN_PAD] float | s_d [3][4][OUT_PX] uintx4 | s_c [2][32][16] float
  const int KG = (p.Cin + 7) / 8;
  uintx4* s_x = reinterpret_cast<uintx4*>(s_raw);
  float* s_e = reinterpret_cast<float*>(s_x + (size_t)3 * KG * IN_PAD);
  uintx4* s_d = reinterpret_cast<uintx4*>(s_e + 32 * IN_PAD);
  float* s_c = reinterpret_cast<float*>(s_d + 3 * 4 * OUT_PX);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int tile = blockIdx.x;
  const int n = tile / (p.tiles_x * p.tiles_y), tt = tile - n * p.tiles_x * p.tiles_y;
  const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const int H = p.H, W = p.W, Cin = p.Cin, Hd = p.Hd, Cout = p.Cout;
  const size_t HW = (size_t)H * W;

  // ---------------- phase 0: the block input, tile + halo, every channel -> three bf16 terms in B-operand form
  {
    const float lo = sc_act_lo(p.x.act), hi = sc_act_hi(p.x.act);
    const bool aff = p.x.mode != SC_SRC_RAW;
    for (int item = tid; item < KG * IN_PAD; item += 256) {
      const int kg = item / IN_PAD, px = item - kg * IN_PAD;
      const int iy = px / IW, ix = px - iy * IW;
      const int y = iy0 + iy, x = ix0 + ix;
      const bool ok = px < IN_PX && y >= 0 && y < H && x >= 0 && x < W;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = kg * 8 + j;
        float t = 0.f;
        if (ok && c < Cin) {
          t = p.x.x[((size_t)n * Cin + c) * HW + (size_t)y * W + x];
          if (aff) t = sc_pro_affine(t, p.x.cst[(size_t)c * SC_CST], p.x.cst[(size_t)c * SC_CST + 1], lo, hi);
        }
        v[j] = t;
      }
      uintx4 t3[3];
      split8(v, t3);
#pragma unroll
      for (int t = 0; t < 3; ++t) s_x[((size_t)t * KG + kg) * IN_PAD + px] = t3[t];
    }
  }
  // chunk constants: [32][16] floats = scale_e, shift_e, scale_d, shift_d, 9 depthwise taps (double-buffered)
  auto load_consts = [&](int chunk, int buf) {
    for (int i = tid; i < 32 * 13; i += 256) {
      const int cl = i / 13, f = i - cl * 13;
      const int c = chunk * 32 + cl;
      float v = 0.f;
      if (c < Hd) {
        if (f < 2) v = p.cst_e[(size_t)c * SC_CST + f];
        else if (f < 4) v = p.cst_d[(size_t)c * SC_CST + (f - 2)];
        else v = p.wd[(size_t)c * 9 + (f - 4)];
      }
      s_c[(buf * 32 + cl) * 16 + f] = v;
    }
  };
  load_consts(0, 0);

  // project pairs of this wave: pair q = wave + 4*ql -> (cout block q / OUT_PXB, pixel block q % OUT_PXB)
  const int npairs = ((Cout + 31) / 32) * OUT_PXB;
  floatx16 accp[IR_MAXP];
#pragma unroll
  for (int q = 0; q < IR_MAXP; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[q][r] = 0.f;
  __syncthreads();

  for (int chunk = 0; chunk < p.nchunk; ++chunk) {
    const int buf = chunk & 1;
    const float* sc = s_c + buf * 32 * 16;
    // ---------------- (b) expansion of this chunk's 32 hidden channels over the input tile
    for (int pxb = wave; pxb < IN_PXB; pxb += 4) {
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const uintx4* wa = p.we + ((size_t)chunk * p.nks_e * 3) * 64 + lane;
      for (int ks = 0; ks < p.nks_e; ++ks) {
        uintx4 a[3], b[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) a[t] = wa[((size_t)ks * 3 + t) * 64];
        const int kg = 2 * ks + lhi;
#pragma unroll
        for (int t = 0; t < 3; ++t) b[t] = kg < KG ? s_x[((size_t)t * KG + kg) * IN_PAD + pxb * 32 + l31] : (uintx4){0u, 0u, 0u, 0u};
        acc = mfma6(a, b, acc);
      }
      // acc[i] = e[hidden 8*(i/4) + 4*lhi + (i%4)][pixel pxb*32 + l31]
      const int px = pxb * 32 + l31;
      const int iy = px / IW, ix = px - iy * IW;
      const int y = iy0 + iy, x = ix0 + ix;
      const bool inside = px < IN_PX && y >= 0 && y < H && x >= 0 && x < W;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int hl = 8 * (i >> 2) + 4 * lhi + (i & 3);
        const float v = fminf(fmaxf(fmaf(acc[i], sc[hl * 16], sc[hl * 16 + 1]), 0.f), 6.f);
        s_e[hl * IN_PAD + px] = inside ? v : 0.f;
      }
    }
    __syncthreads();
    // ---------------- (c) depthwise 3x3 + BN + ReLU6 -> three bf16 terms in B-operand form; next chunk's constants
    if (tid < OUT_PX * 4) {
      const int px = tid % OUT_PX, kg = tid / OUT_PX;
      const int oy = px / TW, ox = px - oy * TW;
      const int base = (oy * S) * IW + ox * S;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cl = kg * 8 + j;
        const float* e = s_e + cl * IN_PAD + base;
        const float* wk = sc + cl * 16 + 4;
        float a = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) a = fmaf(wk[ky * 3 + kx], e[ky * IW + kx], a);
        v[j] = fminf(fmaxf(fmaf(a, sc[cl * 16 + 2], sc[cl * 16 + 3]), 0.f), 6.f);
      }
      uintx4 t3[3];
      split8(v, t3);
#pragma unroll
      for (int t = 0; t < 3; ++t) s_d[(t * 4 + kg) * OUT_PX + px] = t3[t];
    }
    if (chunk + 1 < p.nchunk) load_consts(chunk + 1, buf ^ 1);
    __syncthreads();
    // ---------------- (d) projection: accumulate this chunk's contribution
#pragma unroll
    for (int ql = 0; ql < IR_MAXP; ++ql) {
      const int q = wave + 4 * ql;
      if (q < npairs) {
        const int cob = q / OUT_PXB, pxb = q - cob * OUT_PXB;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int ks = 2 * chunk + k2;
          if (ks < p.nks_p) {
            const uintx4* wa = p.wp + (((size_t)cob * p.nks_p + ks) * 3) * 64 + lane;
            uintx4 a[3], b[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = wa[(size_t)t * 64];
#pragma unroll
            for (int t = 0; t < 3; ++t) b[t] = s_d[(t * 4 + 2 * k2 + lhi) * OUT_PX + pxb * 32 + l31];
            accp[ql] = mfma6(a, b, accp[ql]);
          }
        }
      }
    }
    // (the barrier after (b) of the next chunk orders these s_d reads before the next (c) overwrites them)
  }
  // ---------------- raw projection output: accp[ql][i] = p[cout cob*32 + 8*(i/4) + 4*lhi + (i%4)][pixel pxb*32 + l31]
  const size_t HWo = (size_t)p.Ho * p.Wo;
#pragma unroll
  for (int ql = 0; ql < IR_MAXP; ++ql) {
    const int q = wave + 4 * ql;
    if (q >= npairs) continue;
    const int cob = q / OUT_PXB, pxb = q - cob * OUT_PXB;
    const int px = pxb * 32 + l31;
    const int oy = oy0 + px / TW, ox = ox0 + px % TW;
    if (oy >= p.Ho || ox >= p.Wo) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = cob * 32 + 8 * (i >> 2) + 4 * lhi + (i & 3);
      if (co < Cout) p.out[((size_t)n * Cout + co) * HWo + (size_t)oy * p.Wo + ox] = accp[ql][i];
    }
  }
}

size_t ir_lds_bytes(int Cin, int stride) {
  const int TH = stride == 1 ? 8 : 4, TW = 8;
  const int halo = stride == 1 ? 2 : 1;
  const int in_px = (TH * stride + halo) * (TW * stride + halo), in_pad = (in_px + 31) / 32 * 32, out_px = TH * TW;
  const int KG = (Cin + 7) / 8;
  return (size_t)3 * KG * in_pad * 16 + (size_t)32 * in_pad * 4 + (size_t)3 * 4 * out_px * 16 + (size_t)2 * 32 * 16 * 4;
}

}  // namespace

extern "C" int sc_ir_block_eval_supported(int Cin, int hidden, int Cout, int stride) {
  if (stride != 1 && stride != 2) return 0;
  if (Cin < 1 || hidden < 1 || Cout < 1) return 0;
  const int pairs = ((Cout + 31) / 32) * (stride == 1 ? 2 : 1);
  return ir_lds_bytes(Cin, stride) <= 160 * 1024 && pairs <= 4 * IR_MAXP;
}

extern "C" int sc_ir_block_eval(const sc_src* x, const float* wpk_expand, const float* wpk_project, const float* w_dw,
                                const float* cst_expand, const float* cst_dw, float* out, int N, int Cin, int hidden, int Cout,
                                int H, int W, int stride, sc_stream stream) {
  SC_REQUIRE(x && x->x && wpk_expand && wpk_project && w_dw && cst_expand && cst_dw && out, "sc_ir_block_eval: null argument");
  SC_REQUIRE(x->C == Cin && x->up == 0 && (x->mode == SC_SRC_RAW || (x->mode == SC_SRC_AFFINE && x->cst)), "sc_ir_block_eval: the block input must be a RAW or AFFINE source of Cin channels");
  SC_REQUIRE(N > 0 && H > 0 && W > 0, "sc_ir_block_eval: bad shape");
  SC_REQUIRE(sc_ir_block_eval_supported(Cin, hidden, Cout, stride), "sc_ir_block_eval: unsupported block (Cin %d, hidden %d, Cout %d, stride %d)", Cin, hidden, Cout, stride);
  SC_REQUIRE((((uintptr_t)wpk_expand) | ((uintptr_t)wpk_project)) % 16 == 0, "sc_ir_block_eval: packed filters must be 16-byte aligned");
  IrP p;
  p.x = to_srcd(*x);
  p.we = reinterpret_cast<const uintx4*>(wpk_expand); p.wp = reinterpret_cast<const uintx4*>(wpk_project);
  p.wd = w_dw; p.cst_e = cst_expand; p.cst_d = cst_dw; p.out = out;
  p.N = N; p.Cin = Cin; p.Hd = hidden; p.Cout = Cout; p.H = H; p.W = W;
  p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
  p.nks_e = (Cin + 15) / 16; p.nchunk = (hidden + 31) / 32; p.nks_p = (hidden + 15) / 16;
  const int TH = stride == 1 ? 8 : 4, TW = 8;
  p.tiles_x = (p.Wo + TW - 1) / TW; p.tiles_y = (p.Ho + TH - 1) / TH;
  const size_t lds = ir_lds_bytes(Cin, stride);
  const dim3 grid((unsigned)((long)N * p.tiles_x * p.tiles_y));
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ir_eval<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
    hipLaunchKernelGGL(k_ir_eval<1>, grid, dim3(256), lds, st, p);
  } else {
    static bool attr2 = false;
    if (!attr2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ir_eval<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr2 = true; }
    hipLaunchKernelGGL(k_ir_eval<2>, grid, dim3(256), lds, st, p);
  }
  SC_LAUNCH_OK("sc_ir_block_eval");
  return SC_OK;
}
