/*
 * starcop_hip.h -- C ABI of libstarcop_hip.so (gfx950 / MI355X).
 *
 * The reference (spaceml-org/STARCOP) is pure Python on stock torch ops; it has
 * no FFI of its own.  Each entry point below therefore replaces the *torch op
 * sequence* the reference dispatches on its segmentation hot path, and cites the
 * reference lines whose arithmetic it implements.  A reference maintainer binds
 * these with ctypes; starcop_amd/_lib.py is that binding.
 *
 * Conventions
 *   - every function returns 0 on success or a negative sc_status; it never
 *     throws and never synchronises the device; the text of the last error on
 *     the calling thread is returned by sc_last_error().
 *   - all pointers are DEVICE pointers unless the parameter name ends in _host;
 *     tensors are dense NCHW fp32 unless stated; `stream` is a hipStream_t
 *     (pass torch.cuda.current_stream().cuda_stream).
 *   - no hidden allocation: scratch is caller-provided.
 */
#ifndef STARCOP_HIP_H
#define STARCOP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sc_stream;

enum sc_status {
  SC_OK = 0,
  SC_ERR_ARG = -1,      /* bad shape / unsupported configuration  -> ValueError        */
  SC_ERR_LAUNCH = -2,   /* HIP launch/runtime error               -> RuntimeError      */
  SC_ERR_NOTPD = -3,    /* covariance not positive definite       -> LinAlgError       */
  SC_ERR_NODEV = -4     /* no gfx950 device                       -> RuntimeError      */
};

/* how an activation tensor is read ("normalise on load") */
enum sc_src_mode {
  SC_SRC_RAW = 0,     /* v = x                                                              */
  SC_SRC_AFFINE = 1,  /* v = act(x*c[0] + c[1])            eval/train BatchNorm + ReLU/ReLU6 */
  SC_SRC_BNBWD = 2,   /* v = A*(pass(aux*c[0]+c[1]) ? x : 0) + B*aux + D,  (A,B,D)=c[2..4]  */
                      /*     x = dL/d(act(BN(y))), aux = y : BatchNorm+activation backward  */
  SC_SRC_NORM = 3     /* v = clamp((x-c[0])/c[1], c[2], c[3])   DataNormalizer.normalize_x  */
};
enum sc_act { SC_ACT_NONE = 0, SC_ACT_RELU = 1, SC_ACT_RELU6 = 2 };

#define SC_CST 8          /* floats of per-channel constants per channel            */
/* BatchNorm statistics are written as per-work-group partial rows [rows][C][2] (plain stores, no atomics,
 * deterministic) and summed in fp64 by sc_bn_finalize / sc_bn_bwd_finalize.  rows = sc_stat_rows(kind, ...) */
enum sc_stat_kind { SC_STAT_CONV3 = 0, SC_STAT_CONV1 = 1, SC_STAT_DW = 2, SC_STAT_STEM = 3, SC_STAT_BNBWD = 4,
                    SC_STAT_CONV1K = 5 /* sc_conv1x1_ksplit: one row per 32 pixels */,
                    SC_STAT_PW3 = 6 /* sc_conv1x1_pw3: one row per 32 flat pixels of the whole batch */ };

typedef struct sc_src {
  const float* x;    /* primary tensor  [N, C, H>>up, W>>up]                          */
  const float* aux;  /* secondary tensor (SC_SRC_BNBWD) or NULL                       */
  const float* cst;  /* per-channel constants [C][SC_CST] or NULL (SC_SRC_RAW)        */
  int32_t C;         /* channels of this source                                       */
  int32_t mode;      /* sc_src_mode                                                   */
  int32_t act;       /* sc_act                                                        */
  int32_t up;        /* 1: stored at half resolution, read with nearest x2 upsample   */
} sc_src;

/* ------------------------------------------------------------------------- */
/* library                                                                    */
const char* sc_last_error(void);
int sc_version(void);
int sc_device_check(void);   /* 0 iff the current device is gfx950 */

/* ------------------------------------------------------------------------- */
/* weights: pack OIHW -> kernel layouts (replaces nothing in the reference; it is
 * the layout step torch's conv backends do internally).
 *   fwd  : wpk[co_tile][ci][tap][co_in_tile]                (co_t = 16, 32 or 64; ci zero-padded to the K chunk)
 *   dgrad: same layout of the transposed+flipped filter: W'[ci][co][2-kh][2-kw]
 */
int sc_pack_weights(const float* w_oihw, float* wpk, int Cout, int Cin, int ks,
                    int co_t, int transpose_flip, sc_stream stream);
size_t sc_packed_weight_floats(int Cout, int Cin, int ks, int co_t, int transpose_flip);

/* every filter pack of a network in one launch.  `descs` and `block_starts` (n entries: first 256-thread block of each
 * descriptor, blocks = ceil(sc_pack_work_items / 256)) live in DEVICE memory and are built once by the caller;
 * bx3 = number of bf16 terms (3, 2 or 1) selects the sc_pack_weights_bx3 layout (ks must be 3); 0 = fp32 layout. */
typedef struct sc_pack_desc {
  const float* w; float* wpk;
  int32_t Cout, Cin, ks, co_t, transpose_flip, bx3;
  uint64_t total;        /* = sc_pack_work_items(...) */
} sc_pack_desc;
size_t sc_pack_work_items(int Cout, int Cin, int ks, int co_t, int transpose_flip, int bx3);
int sc_pack_weights_batch(const sc_pack_desc* descs_dev, const uint32_t* block_starts_dev, int n, uint32_t total_blocks,
                          sc_stream stream);

/* ------------------------------------------------------------------------- */
/* dense conv (groups=1, stride 1, pad ks/2, ks in {1,3}) as implicit GEMM on fp32 MFMA.
 * Replaces torch.nn.functional.conv2d (+ cat + interpolate(nearest) + batch_norm + relu/relu6
 * of the producers) as dispatched by smp.Unet at starcop/models/model_module.py:244,
 * and its backward-data when called with transpose_flip-packed weights.
 *   input  = channel concat of nsrc (1|2) sources, each with its own prologue
 *   output = out0 (channels [0,csplit)) and out1 (channels [csplit,Cout)); csplit==Cout -> single
 *            out = acc (+ add0) (+ add1) (+ old out if accum flag)
 *   stats  : if non-NULL, per-channel sum / sum-of-squares of acc over each work-group's pixel tile are written to
 *            stats[row][Cout][2] (float), row = n*tiles + tile, rows = sc_stat_rows(SC_STAT_CONV3|CONV1, N, H, W)
 * ks = 1 on planes of >= 8192 pixels (H*W % 256 == 0, 16-byte aligned tensors) with a short contraction (16-32 source channels, one plain
 * output of <= 160 channels, RAW / AFFINE or BNBWD source) runs on a streaming kernel (16-byte loads and stores, same fp32 products,
 * same statistics rows); everything else on the LDS-staged kernels.
 */
/* BatchNorm-backward sums of the tensor whose gradient a data-gradient launch writes into out0 (sc_conv3x3_bx3, sc_conv3x3_thin16,
 * sc_conv3x3_sp_dgrad): what sc_bn_bwd_reduce(out0, y, cst, act, ...) would compute by streaming both tensors again, left by the
 * launch that produces the gradient -- valid when that launch writes the COMPLETE gradient (single consumer: accum0 = 0, no add
 * tensors).  rows[sc_stat_rows(SC_STAT_CONV3, N, H, W)][C][2] (float) = {sum g', sum g' x_hat} over each work-group's pixel tile,
 * g' = out0 * act'(BN(y)), for sc_bn_bwd_finalize_rows32 (sc_conv3x3_sp_dgrad: not taken);
 * absmax (or NULL): raised (atomic max; zeroed by the caller) to max |scale_c g'|, the range hint of the SC_TERMS_F16X2 kernels. */
typedef struct sc_bnr_args {
  const float* y;        /* raw (pre-BatchNorm) values of the tensor, same shape as out0   */
  const float* cst;      /* its forward constants {scale, shift, mean, invstd, ..} [C][SC_CST] */
  int32_t act;           /* its activation (SC_ACT_*)                                       */
  float* rows;
  float* absmax;
} sc_bnr_args;
#define SC_TERMS_F16X2 4   /* `terms` code: two fp16 terms per operand with exact power-of-two range scaling (22 significand
                            * bits, three products: fp32-level accuracy at half the MFMA work of the three-term bf16 split) */
typedef struct sc_conv_args {
  sc_src src[2];
  int32_t nsrc;
  const float* wpk;      /* packed weights, see sc_pack_weights                 */
  int32_t N, H, W;       /* output (= input) spatial size                        */
  int32_t Cout;
  int32_t ks;            /* 1 or 3                                               */
  int32_t co_t;          /* 16 (thin layers, ks=3, Cout<=16), 32 or 64: cout tile the weights were packed for */
  float* out0; float* out1;
  int32_t csplit;        /* == Cout when there is a single output                */
  int32_t accum0, accum1;/* 1: out += result                                     */
  const float* add0;     /* optional [N,Cout,H,W] tensors added in the epilogue  */
  const float* add1;     /*   (only valid with csplit == Cout)                   */
  float* stats;          /* [rows][Cout][2] partial sums or NULL                 */
  int32_t terms;         /* sc_conv3x3_bx3 only: bf16 terms per operand, 0 or 3 = fp32-accurate split, 2 = two terms (2^-18), 1 = plain bf16 */
  int32_t down0;         /* sc_conv3x3_bx3 only: 1 = channels [0,csplit) are stored 2x2-summed at half resolution into out0
                          * ([N,csplit,H/2,W/2]): the backward of F.interpolate(scale_factor=2, mode="nearest") in
                          * smp's DecoderBlock fused into the data-gradient store (no full-resolution temporary) */
  const float* absmax;   /* terms == SC_TERMS_F16X2 with a BNBWD source: device float >= the tensor's max |A_c g| (written by
                          * sc_bn_bwd_reduce / sc_bn_bwd_small); NULL: the gradient operand is taken to be O(1)              */
  const float* xbound[2];/* terms == SC_TERMS_F16X2, forward sources: per source a device float >= max |activation| of that source
                          * (sc_bn_finalize(act_bound) in training, sc_add_srcs_absmax records) or NULL (ReLU6-bounded / unknown): the
                          * kernels scale the fp16 operand by 2 when 2 M <= 32752 and by the largest power of two with s M <= 32752
                          * otherwise, so a BatchNorm'd activation can never reach the +-65504 clamp                         */
  const sc_bnr_args* bnr;/* data-gradient launches (SC_SRC_BNBWD source): also leave the BatchNorm-backward sums of out0's tensor, or NULL */
} sc_conv_args;
int sc_conv2d_mfma(const sc_conv_args* a, sc_stream stream);
/* 1x1 convolution for few-pixel / long-K layers (the <= 64^2 inverted-residual projections and the data gradients of the
 * expansions): 32-pixel tiles, the four waves of a work-group split K, operands go global -> registers -> MFMA (no LDS
 * staging).  Same arguments as sc_conv2d_mfma with ks = 1, nsrc = 1 and sc_pack_weights(ks=1) filters; co_t in {32, 64};
 * statistics rows = sc_stat_rows(SC_STAT_CONV1K, N, H, W). */
int sc_conv1x1_ksplit(const sc_conv_args* a, sc_stream stream);

/* The same 3x3 convolution (forward, or backward-data with transpose_flip-packed filters) with fp32 accuracy on the
 * bf16 matrix cores: every fp32 operand is split exactly into three bf16 terms while it is staged and the six partial
 * products of weight >= 2^-24 are accumulated in fp32 (6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 block, 2.7x the
 * v_mfma_f32_32x32x2_f32 rate; error of the same order as one fp32 rounding per product).  Same arguments, prologues,
 * epilogue and statistics rows (SC_STAT_CONV3) as sc_conv2d_mfma with ks = 3; co_t in {32, 64}; `wpk` must come from
 * sc_pack_weights_bx3 (16-byte aligned).  A concat needs src[0].C % 16 == 0.
 * `terms` = 1 (args->terms and the pack call) keeps only the leading bf16 term of each operand: one MFMA per block, bf16
 * matrix math with fp32 accumulation -- the network's "bf16" precision mode (BASELINE configs[3]); tensors stay fp32.
 * `terms` = 2 keeps two terms per operand and the three products a0*b1 + a1*b0 + a0*b0 (operand error 2^-18): the opt-in
 * "fp32-bwd2" / "fp32-2" modes.
 * Replaces the same torch.nn.functional.conv2d calls as sc_conv2d_mfma (starcop/models/model_module.py:244). */
int sc_pack_weights_bx3(const float* w_oihw, float* wpk, int Cout, int Cin, int co_t, int transpose_flip, int terms,
                        sc_stream stream);
size_t sc_packed_weight_floats_bx3(int Cout, int Cin, int co_t, int transpose_flip, int terms);
int sc_conv3x3_bx3(const sc_conv_args* a, sc_stream stream);

/* weight gradient of the same conv: dW[co][ci][kh][kw] = sum_{n,y,x} dy * in.
 *   dy  : one source (normally SC_SRC_BNBWD: g + y of the conv output), Cout channels
 *   in  : concat of nsrc sources with the forward prologues, Cin channels
 *   part: scratch [nslices][taps][Cout_pad32][Cin_pad32]; call sc_wgrad_workspace_floats
 *   dw  : [Cout][Cin][ks][ks]  (overwritten)
 */
typedef struct sc_wgrad_args {
  sc_src dy;
  sc_src src[2];
  int32_t nsrc;
  int32_t N, H, W, Cout, Cin, ks;
  float* part; size_t part_floats;
  float* dw;
  int32_t terms;         /* sc_conv3x3_wgrad_bx3 only: 0 or 3 = fp32-accurate split, 2 = two terms, 1 = plain bf16 operands */
  const float* absmax;   /* terms == SC_TERMS_F16X2: scale hint of dy, as in sc_conv_args                                     */
  const float* xbound[2];/* terms == SC_TERMS_F16X2: activation bounds of the input sources, as in sc_conv_args               */
} sc_wgrad_args;
size_t sc_wgrad_workspace_floats(int N, int H, int W, int Cout, int Cin, int ks);
int sc_conv2d_wgrad_mfma(const sc_wgrad_args* a, sc_stream stream);
/* The same kernel without its trailing reduction launches: the K-slice partials stay in a->part (which must then be a buffer
 * of this layer's own, alive until the batch reduction) and `pending` (HOST struct) receives what sc_wgrad_reduce_batch needs.
 * A network's ~35 pointwise layers each end in 2-3 few-microsecond dependent launches otherwise; the batch sums all of them in
 * ONE launch at the end of the backward pass.  descs_dev / block_starts_dev: DEVICE arrays built once by the caller (the
 * descriptors do not change between steps), block_starts[i] = first 256-thread block of descriptor i, blocks_i = ceil(total_i/256). */
typedef struct sc_wgrad_pending {
  const float* part; float* dw;
  int32_t nparts, taps, Cout, Cin, CoP, CiP;
  uint64_t total;        /* taps * Cout * Cin */
} sc_wgrad_pending;
int sc_conv2d_wgrad_mfma_deferred(const sc_wgrad_args* a, sc_wgrad_pending* pending_host, sc_stream stream);
int sc_wgrad_reduce_batch(const sc_wgrad_pending* descs_dev, const uint32_t* block_starts_dev, int n, uint32_t total_blocks,
                          sc_stream stream);
/* Thin 3x3 layers (Cout <= 16, Cin = 16 or 32: smp's decoder.blocks.4 at full resolution) with two fp16 terms on
 * v_mfma_f32_16x16x32_f16: the filter bank stays in registers, all input channels are staged in one pass.  Same arguments as
 * sc_conv3x3_bx3 with terms = SC_TERMS_F16X2, nsrc = 1, a single plain output (csplit = Cout, no add / accumulate / down0);
 * `wpk` from sc_pack_weights_thin16 (or a sc_pack_desc with bx3 = SC_PACK_THIN16); statistics rows SC_STAT_CONV3.
 * Forward, or backward-data with transpose_flip-packed filters (then "Cout" is the layer's input channel count).
 * A 32-channel source with up = 1 (decoder.blocks.4.conv1: conv3x3 of a nearest-2x up-sampled tensor) runs as four 2x2 convolutions on
 * the half-resolution source (sub-pixel form: the pack of a forward filter with Cin = 32 carries the 16 phase filters behind its 3x3
 * entries).  down0 = 1 (the one exception to "no down0"): the data gradient of that layer -- 16-channel SC_SRC_BNBWD source at H x W,
 * Cout = csplit = 32 rows stored 2x2-summed into out0 [N,32,H/2,W/2] (accum0 allowed) -- from sc_pack_weights_thin16(Cout = 16,
 * Cin = 32, transpose_flip = 1), which packs the 4 x 4-position gathered filters of that form (no statistics / bnr / add epilogue). */
#define SC_PACK_THIN16 5
#define SC_PACK_PW3 6      /* sc_pack_desc.bx3 code of the pointwise layout of sc_conv1x1_pw3 (ks = 1; co_t ignored) */
size_t sc_packed_weight_floats_thin16(int Cout, int Cin, int transpose_flip);
int sc_pack_weights_thin16(const float* w_oihw, float* wpk, int Cout, int Cin, int transpose_flip, sc_stream stream);
int sc_conv3x3_thin16(const sc_conv_args* a, sc_stream stream);

/* Decoder conv1 as a SUB-PIXEL convolution: smp's DecoderBlock runs conv3x3(cat([interpolate(prev, x2, "nearest"), skip]))
 * (starcop/models/model_module.py:244-251).  For the up-sampled channels, conv3x3(nearest_up2(x)) is exactly four phase-specific
 * 2x2 convolutions on the LOW-resolution x (phase filter = sum of the taps that land on the same source pixel; zero padding maps
 * 1:1): 2.25x fewer multiply-adds and every low-resolution value staged once instead of once per high-resolution copy.  The
 * skip channels join the same launch as four low-resolution "parity planes" each (csrc/conv_sp_pack.h).
 *   sc_conv3x3_sp: sc_conv_args with src[0] = the half-resolution tensor (up = 1), optional src[1] = the full-resolution skip
 *   tensor (up = 0), both RAW or AFFINE; H x W = OUTPUT size (even); ks = 3; terms = SC_TERMS_F16X2 (the two-fp16-term arithmetic
 *   of sc_conv3x3_bx3) or 1 (one bf16 term per operand: the "bf16" precision mode; xbound unused); csplit = Cout, one plain output (no add / accumulate / down0); co_t ignored (32); `wpk` from
 *   sc_pack_weights_sp (or a sc_pack_desc with bx3 = SC_PACK_SP, Cin = the filter's total input channels and co_t = how many
 *   of them, the leading ones, belong to the up-sampled source; transpose_flip = 4 for the one-bf16-term layout); statistics rows = sc_sp_stat_rows(N, H, W, Cout) (one per work-group tile: 8 x 32 low-resolution pixels; 16 x 16 -- or, for launches of at most 128 such work-groups, 8 x 16 -- on planes narrower than 32). */
#define SC_PACK_SP 7
size_t sc_packed_weight_floats_sp(int Cout, int Cup, int Cskip);
int sc_pack_weights_sp(const float* w_oihw, float* wpk, int Cout, int Cup, int Cskip, int terms, sc_stream stream);
int sc_sp_stat_rows(int N, int H, int W, int Cout);
int sc_conv3x3_sp(const sc_conv_args* a, sc_stream stream);
/* ... and its data gradient w.r.t. the half-resolution source: a stride-2 4x4 convolution of dy (four parity planes x 2x2 taps), written
 * at half resolution directly -- no full-resolution gradient of the up-sampled channels, no 2x2 down-sum (replaces
 * sc_conv3x3_bx3(down0 = 1) on those channels; the skip channels' gradient stays an ordinary 3x3 launch).
 *   sc_conv3x3_sp_dgrad: sc_conv_args with nsrc = 1, src[0] = the SC_SRC_BNBWD operand of the layer's output (g, y, constants; H x W),
 *   Cout = csplit = the up-sampled source's channels, out0 = [N, Cout, H/2, W/2] (accum0 allowed), terms = SC_TERMS_F16X2 or 1, absmax as
 *   in sc_conv3x3_bx3; `wpk` from sc_pack_weights_spd (or a sc_pack_desc with bx3 = SC_PACK_SPD, Cin = the filter's total input
 *   channels, co_t = the up-sampled source's channels, transpose_flip = 2 for vskip, | 4 for the one-bf16-term layout).
 *   vskip (sc_spd_vskip_ok: <= 64 up-sampled and <= 16 skip channels, smp's decoder.blocks.3): the skip channels' full-resolution
 *   gradient rides along in the otherwise idle half of the 128-channel tile -- Cout = all input channels, csplit = the up-sampled ones,
 *   out1 = [N, Cout - csplit, H, W] (accum1 allowed): ONE launch stages dy for both gradients.
 *   skip tiles (any other channel counts, out1 given; pack with vskip = 2 / transpose_flip = 3): the skip channels' gradient as
 *   additional 128-channel tiles of the launch (32 skip channels x 4 output parities each) -- same arguments as vskip; which of the two
 *   forms a launch takes follows from sc_spd_vskip_ok(csplit, Cout - csplit), so the pack must be made with the matching mode. */
#define SC_PACK_SPD 8
size_t sc_packed_weight_floats_spd(int Cout, int Cup, int Cskip_tiles /* skip channels packed as skip tiles (vskip = 2), else 0 */);
int sc_spd_vskip_ok(int Cup, int Cskip);
int sc_pack_weights_spd(const float* w_oihw, float* wpk, int Cout, int CinTotal, int Cup, int vskip, int terms, sc_stream stream);
int sc_conv3x3_sp_dgrad(const sc_conv_args* a, sc_stream stream);
/* ... and the weight gradient of its up-sampled channels as a PLAIN GEMM: up(x) is constant over 2x2 blocks of the output grid, so
 * dW[co][ci][kh][kw] = sum_q x[ci][q] * S_(kh,kw)[co][q] with S the tap-aligned 2x2 box sums of dy -- nine GEMMs over the low-resolution
 * pixels with no spatial shift, a quarter of the 3x3 form's multiply-adds (csrc/conv_spw.hip: box sums + exact two-fp16-term split in
 * one pass over (g, y); the activated, split source; a two-term fp16 GEMM; a fixed-order reduction of its K slices).
 *   sc_conv3x3_sp_wgrad: sc_wgrad_args with nsrc = 1, src[0] = the half-resolution tensor (up = 1; RAW / AFFINE; xbound[0]), dy = the
 *   layer's output gradient (H x W, even; normally SC_SRC_BNBWD; absmax), Cin = the FILTER's total input channels, terms =
 *   SC_TERMS_F16X2; writes dw[co][ci][3][3] for ci < src[0].C inside the [Cout][Cin][3][3] gradient (part / part_floats unused);
 *   `ws`: sc_sp_wgrad_workspace_bytes(N, H, W, Cout, src[0].C) bytes, 256-byte aligned.
 *   sc_wgrad_scatter_cols: a dense [Cout][Ccols][3][3] gradient (the skip channels', from sc_conv3x3_wgrad_bx3 on the skip source
 *   alone) into columns [col_off, col_off + Ccols) of the [Cout][CinTotal][3][3] gradient. */
size_t sc_sp_wgrad_workspace_bytes(int N, int H, int W, int Cout, int Cup);
int sc_conv3x3_sp_wgrad(const sc_wgrad_args* a, void* ws, size_t ws_bytes, sc_stream stream);
int sc_wgrad_scatter_cols(const float* src, float* dw, int Cout, int Ccols, int CinTotal, int col_off, sc_stream stream);

/* weight gradient of the same thin layers (Cout <= 16, Cin = 16 | 32, one source which may be upsampled) with two fp16 terms on
 * v_mfma_f32_16x16x32_f16: same sc_wgrad_args as sc_conv2d_wgrad_mfma with terms = SC_TERMS_F16X2 (absmax = the dy range hint),
 * workspace from sc_wgrad_thin16_workspace_floats.  Replaces the fp32-MFMA thin weight gradient (the step's last MFMA-bound fp32 kernel). */
size_t sc_wgrad_thin16_workspace_floats(int N, int H, int W, int Cout, int Cin);
int sc_conv3x3_wgrad_thin16(const sc_wgrad_args* a, sc_stream stream);

/* Pointwise (1x1) convolutions on the 16-bit matrix cores with fp32 accuracy (every fp32 operand split exactly into three bf16
 * terms, six products, fp32 accumulation: the arithmetic of sc_conv3x3_bx3 with terms = 3), without LDS staging: the MobileNetV2
 * encoder's expansion / projection convolutions (torchvision InvertedResidual.conv, smp.Unet's encoder at
 * starcop/models/model_module.py:244-251), forward or backward-data (transpose_flip-packed filters, SC_SRC_BNBWD source).
 * Same sc_conv_args as sc_conv2d_mfma with ks = 1, nsrc = 1, a single output (csplit = Cout; add0 / accum0 allowed; no add1,
 * out1, down0, bnb_*); `wpk` from a sc_pack_desc with bx3 = SC_PACK_PW3 (sc_packed_weight_floats_pw3 floats); co_t / terms are
 * ignored; statistics rows = sc_stat_rows(SC_STAT_PW3, N, H, W) = one row per 32 flat pixels of the batch. */
size_t sc_packed_weight_floats_pw3(int Cout, int Cin, int transpose_flip);
int sc_conv1x1_pw3(const sc_conv_args* a, sc_stream stream);
/* its weight gradient (K = pixels; H*W must be a multiple of 8): same sc_wgrad_args as sc_conv2d_wgrad_mfma with ks = 1, nsrc = 1;
 * workspace from sc_wgrad_pw3_workspace_floats; pending != NULL defers the sum over the K-slice partials to
 * sc_wgrad_reduce_batch (as sc_conv2d_wgrad_mfma_deferred), NULL finishes it here. */
size_t sc_wgrad_pw3_workspace_floats(int N, int H, int W, int Cout, int Cin);
int sc_conv1x1_wgrad_pw3(const sc_wgrad_args* a, sc_wgrad_pending* pending_host, sc_stream stream);

/* One launch per MobileNetV2 inverted-residual block in INFERENCE (conv_irb.hip; torchvision InvertedResidual.conv inside
 * smp.Unet('mobilenet_v2').encoder, eval-mode BatchNorm: starcop/models/model_module.py:90-98 forward, :244-251 the network; the
 * notebook / padded_predict path starcop/models/utils/padding.py:13-50):
 *     x -> conv1x1 (Cin -> hidden) -> BN_e + ReLU6 -> depthwise 3x3 (stride 1 | 2, pad 1) -> BN_d + ReLU6 -> conv1x1 (hidden -> Cout)
 *       -> raw p (residual = 0: BN_p is the consumers' business, as for every other convolution here)
 *       -> z = x + BN_p(p)  (residual = 1: Cin == Cout; z_absmax, if not NULL, is raised to max |z| like sc_add_srcs_absmax does)
 * The expanded tensors never leave the CU.  Both 1x1 filters come in the sc_conv1x1_pw3 layout (sc_pack_weights_batch with
 * SC_PACK_PW3, transpose_flip 0); fp32 accuracy (three exact bf16 terms per operand, six MFMA products, fp32 stencil).
 * Stride 1: 8 <= Cin <= 160, hidden % 32 == 0, Cout <= 384 subject to the accumulator budget; stride 2 (no residual; H, W are the INPUT
 * plane, the output is ((H - 1) / 2 + 1) x ((W - 1) / 2 + 1)): Cin <= 96, hidden % 64 == 0 (sc_irb_supported). */
typedef struct sc_irb_args {
  sc_src x;                  /* block input [N,Cin,H,W]: SC_SRC_RAW or SC_SRC_AFFINE                                  */
  const float* wpk_expand;   /* PW3 pack of the expansion filter [hidden][Cin]                                         */
  const float* cst_expand;   /* [hidden][SC_CST]: {scale, shift, ..} of BN_e (sc_bn_finalize, eval)                    */
  const float* w_dw;         /* [hidden][3][3]                                                                         */
  const float* cst_dw;       /* [hidden][SC_CST] of BN_d                                                               */
  const float* wpk_project;  /* PW3 pack of the projection filter [Cout][hidden]                                       */
  const float* cst_project;  /* [Cout][SC_CST] of BN_p (residual = 1) or NULL                                          */
  float* out;                /* [N,Cout,Ho,Wo]: raw p, or z                                                            */
  float* z_absmax;           /* device float raised to max |z| (residual = 1) or NULL                                  */
  int32_t N, Cin, hidden, Cout, H, W, stride, residual;
} sc_irb_args;
int sc_irb_supported(int Cin, int hidden, int Cout, int H, int W, int stride);
int sc_irb_eval(const sc_irb_args* a, sc_stream stream);

/* Fused TRAINING execution of the expansion + depthwise pair of a STRIDE-2 MobileNetV2 inverted-residual block (torchvision
 * InvertedResidual.conv[0..1] inside smp.Unet('mobilenet_v2'): starcop/models/model_module.py:244-251; train-mode BatchNorm):
 *     e = conv1x1(x, w_expand) -> BN_e + ReLU6 -> depthwise 3x3 (stride 2, pad 1) -> d (raw; BN_d is the consumers' business)
 * The 6x-expanded tensor e (four times the size of d) and its gradient are NEVER stored: every sweep recomputes e from the block
 * input x on the matrix cores (fp32 accuracy: three exact bf16 terms per operand, six products).  Replaces, per block and step,
 *   sc_conv1x1_* (expand, +stats) | sc_dwconv3x3_fwd | sc_dwconv3x3_bwd_fused | expand data gradient | expand weight gradient.
 * Cin in {8, 16, 24, 32}, hidden <= 192, H % 4 == 0, W % 8 == 0, stride 2 (sc_irt_supported).
 *   forward :  sc_irt_expand_stats -> sc_bn_finalize(BN_e, rows = sc_irt_rows(0,..)) -> sc_irt_fwd -> sc_bn_finalize(BN_d, rows =
 *              sc_irt_rows(1,..)) -> the projection convolution reads d as an SC_SRC_AFFINE source, as before
 *   backward:  (dy_d = the SC_SRC_BNBWD source of d, from sc_bn_bwd_* on d;  `work`: sc_irt_bwd_workspace_floats floats, 16-byte aligned)
 *              sc_irt_bwd        ONE sweep over (dy_d, x): e_sums rows for sc_bn_bwd_finalize(BN_e) (rows = sc_irt_bwd_rows), the
 *                                depthwise filter gradient (dw_acc[hidden][9] +=, fp64 atomics, as sc_dwconv3x3_bwd_fused), partial
 *                                rows of G = sum_px g'_e x^T and the part of dx that does not depend on the batch sums,
 *                                W_e^T (scale (.) g'_e), into `work`
 *              sc_bn_bwd_finalize(BN_e) -> cst_bwd_expand (.., .., A, B, D)
 *              sc_irt_bwd_fix    dx = that part + Q x + r (+ add0) (+ dx),  Q = W_e^T diag(B) W_e, r = W_e^T D: exact, because
 *                                dy_e = A g'_e + B e + D is affine per channel and e = W_e x
 *              sc_irt_xmoments   M = sum_px x x^T, s = sum_px x into `work` (any time after x exists; off the critical path)
 *              sc_irt_wgrad_finalize  dw_expand[hidden][Cin] = A (.) G + B (.) (W_e M) + D (x) s                              */
typedef struct sc_irt_args {
  sc_src x;                 /* block input [N,Cin,H,W]: SC_SRC_RAW or SC_SRC_AFFINE                                   */
  const float* w_expand;    /* [hidden][Cin]   (the 1x1 filter, OIHW)                                                 */
  const float* w_dw;        /* [hidden][3][3]                                                                         */
  const float* cst_expand;  /* [hidden][SC_CST] forward constants of BN_e (sc_bn_finalize); unused by sc_irt_expand_stats */
  int32_t N, Cin, hidden, H, W, stride;
} sc_irt_args;
int sc_irt_supported(int Cin, int hidden, int H, int W, int stride);
int sc_irt_rows(int stage, int N, int H, int W, int stride);        /* stage 0: sc_irt_expand_stats, 1: sc_irt_fwd */
int sc_irt_bwd_rows(int N, int hidden, int H, int W);
size_t sc_irt_bwd_workspace_floats(int N, int Cin, int hidden, int H, int W);
int sc_irt_expand_stats(const sc_irt_args* a, float* stats /*[rows][hidden][2]*/, sc_stream stream);
int sc_irt_fwd(const sc_irt_args* a, float* d_out /*[N,hidden,Ho,Wo] raw*/, float* stats_d /*[rows][hidden][2] or NULL (inference)*/, sc_stream stream);
int sc_irt_bwd(const sc_irt_args* a, const sc_src* dy_d, double* e_sums /*[rows][hidden][2]*/, double* dw_acc /*[hidden][9]*/,
               float* work, sc_stream stream);
int sc_irt_xmoments(const sc_irt_args* a, float* work, sc_stream stream);
int sc_irt_bwd_fix(const sc_irt_args* a, const float* cst_bwd_expand, float* work, float* dx, const float* add0, int accum,
                   sc_stream stream);
int sc_irt_wgrad_finalize(const sc_irt_args* a, const float* cst_bwd_expand, float* work, float* dw_expand, sc_stream stream);

/* the 3x3 weight gradient with split-bf16 operands on the bf16 matrix cores (see sc_conv3x3_bx3); same arguments,
 * ks must be 3; workspace from sc_wgrad_bx3_workspace_floats */
size_t sc_wgrad_bx3_workspace_floats(int N, int H, int W, int Cout, int Cin);
int sc_conv3x3_wgrad_bx3(const sc_wgrad_args* a, sc_stream stream);

/* ------------------------------------------------------------------------- */
/* depthwise 3x3 (groups=C, pad 1, stride 1|2): forward, backward-data, backward-weight */
int sc_dwconv3x3_fwd(const sc_src* in, const float* w /*[C][3][3]*/, float* out,
                     int N, int C, int Hin, int Win, int stride, float* stats /* rows: SC_STAT_DW on (Hout,Wout) */,
                     sc_stream stream);
/* Producer-tail BatchNorm finalize: the launch that writes a tensor's statistics rows also finalizes that tensor's BatchNorm (what
 * sc_bn_finalize(stats, rows, count, ..) does in a launch of its own, ~5 us of idle chip behind every producer of the training
 * forward).  The work-group that arrives LAST for a channel -- a ticket per channel: `tickets[C]`, zero-initialised ONCE, counts
 * monotonically over launches -- sums that channel's rows in a fixed order (fp64) and writes constants, running statistics and the
 * activation bound: bit-reproducible whichever work-group does it. */
typedef struct sc_bn_tail {
  const float* gamma; const float* beta;      /* BatchNorm weight / bias [C]                              */
  float* running_mean; float* running_var;    /* updated with `momentum` (training semantics)              */
  float momentum, eps;
  float* cst;                                 /* [C][SC_CST] forward constants {scale, shift, mean, invstd} */
  float* act_bound;                           /* as in sc_bn_finalize, or NULL                             */
  uint32_t* tickets;                          /* [C]                                                       */
} sc_bn_tail;
/* sc_dwconv3x3_fwd + the BatchNorm of its output finalized by the launch itself (the depthwise layers of a batch leave 16-32 rows
 * per channel: the last (image, tile) of a channel reads 128-256 bytes) */
int sc_dwconv3x3_fwd_bn(const sc_src* in, const float* w, float* out, int N, int C, int Hin, int Win, int stride, float* stats,
                        const sc_bn_tail* bn, sc_stream stream);
int sc_dwconv3x3_dgrad(const sc_src* dy, const float* w, float* dx, int accum,
                       int N, int C, int Hin, int Win, int stride, sc_stream stream);
int sc_dwconv3x3_wgrad(const sc_src* dy, const sc_src* in, double* dw_acc /*[C][9] zeroed*/,
                       int N, int C, int Hin, int Win, int stride, sc_stream stream);
/* the three of them in one pass: dx (overwritten), dw_acc (+=, fp64 atomics) and, if in_sums != NULL, the BatchNorm-backward
 * partial sums of the INPUT tensor, in_sums[row][C][2] = {sum g_bn, sum g_bn * xhat}, g_bn = dx * act'(BN(y_in)), rows =
 * sc_stat_rows(SC_STAT_DW, N, Hin, Win) -- what sc_bn_bwd_reduce(dx, y_in) would compute by streaming both tensors again
 * (`in` must then be the SC_SRC_AFFINE source of that BatchNorm'd tensor with its cst_fwd constants).  In a MobileNetV2
 * inverted-residual block both sides of the depthwise conv are the 6x-expanded tensors: this kernel reads (g, y) of the
 * output once instead of twice and saves the (dx, y_in) pass of the expansion's BatchNorm backward. */
int sc_dwconv3x3_bwd_fused(const sc_src* dy, const sc_src* in, const float* w, float* dx, double* dw_acc, double* in_sums,
                           int N, int C, int Hin, int Win, int stride, sc_stream stream);
int sc_cast_f64_f32(const double* in, float* out, size_t n, sc_stream stream);
/* the same for n_descs (in, out, n) triples in one launch; descs_dev: DEVICE array (static for a plan: the caller builds it once).
 * The depthwise filter gradients of a backward walk (fp64 accumulators of sc_dwconv3x3_bwd_fused -> the flat fp32 gradient). */
typedef struct sc_cast_desc { const double* in; float* out; uint64_t n; } sc_cast_desc;
int sc_cast_f64_f32_batch(const sc_cast_desc* descs_dev, int n_descs, sc_stream stream);

/* stem: conv 3x3 stride 2 pad 1, Cin<=8 -> 32, input read through its prologue
 * (SC_SRC_NORM fuses DataNormalizer.normalize_x, starcop/data/normalizer_module.py:134-135) */
int sc_stem_conv_fwd(const sc_src* in, const float* w /*[32][Cin][3][3]*/, float* out,
                     int N, int Cin, int Hin, int Win, float* stats /* rows: SC_STAT_STEM on (Hout,Wout) */,
                     sc_stream stream);
size_t sc_stem_wgrad_workspace_floats(int N, int Cin, int Hin, int Win);
int sc_stem_conv_wgrad(const sc_src* dy, const sc_src* in, float* part, size_t part_floats,
                       float* dw, int N, int Cin, int Hin, int Win, sc_stream stream);

/* segmentation head: conv 3x3 pad 1, Cin -> 1, bias (fwd/dgrad: Cin <= 32; wgrad: Cin in {8, 16}) */
int sc_head_conv_fwd(const sc_src* in, const float* w /*[1][Cin][3][3]*/, const float* bias,
                     float* out, int N, int Cin, int H, int W, sc_stream stream);
int sc_head_conv_dgrad(const float* dlogits, const float* w, float* gin,
                       int N, int Cin, int H, int W, sc_stream stream);
size_t sc_head_wgrad_workspace_floats(int N, int Cin, int H, int W);
int sc_head_conv_wgrad(const float* dlogits, const sc_src* in, float* part, size_t part_floats,
                       float* dw, float* dbias, int N, int Cin, int H, int W, sc_stream stream);
/* dgrad + wgrad + dbias of the head in ONE sweep over (dlogits, in) -- Cin = 16 only (model_module.py:244-251: the decoder's last
 * block has 16 channels); same results as the two calls above, gin = gradient w.r.t. the ACTIVATED input.  Workspace as for
 * sc_head_conv_wgrad. */
/* bn_sums != NULL (then `in` must be the SC_SRC_AFFINE source of a BatchNorm'd tensor): the launch also leaves what
 * sc_bn_bwd_reduce(gin, in->x, in->cst, in->act, ...) would compute in a pass of its own over both tensors -- rows
 * bn_sums[sc_head_bwd_bn_rows(N, H, W)][Cin][2] = {sum g', sum g' x_hat} for sc_bn_bwd_finalize, and (bn_absmax != NULL, zeroed by
 * the caller) the range hint max |scale_c g'| -- because gin is that tensor's complete gradient and its values stream through
 * this kernel anyway. */
int sc_head_bwd_bn_rows(int N, int H, int W);
int sc_head_conv_bwd(const float* dlogits, const sc_src* in, const float* w, float* gin, float* part, size_t part_floats,
                     float* dw, float* dbias, int N, int Cin, int H, int W, double* bn_sums, float* bn_absmax, sc_stream stream);

/* ------------------------------------------------------------------------- */
/* BatchNorm2d bookkeeping (torch.nn.BatchNorm2d inside smp/torchvision blocks)
 * training=1: batch statistics from the `nrows` partial rows of `stats` (count = N*H*W), running stats updated with
 *             `momentum` (unbiased variance), cst_fwd = {scale, shift, mean, invstd,...}
 * training=0: cst_fwd from running stats */
int sc_stat_rows(int kind, int N, int H, int W);
/* scratch (optional, SC_BN_FINALIZE_SCRATCH_DOUBLES(C) doubles): with >= 4096 rows the rows of all channels are first summed as
 * one coalesced stream into 64 fp64 partial rows there (the per-channel walk reads a 64-byte sector per 8 useful bytes) */
#define SC_BN_FINALIZE_SCRATCH_DOUBLES(C) (64 * 2 * (size_t)(C))
/* act_bound (optional, training=1): device float raised (order-independent atomic max; never lowered) to
 * max_c |gamma_c| sqrt(count - 1) + |beta_c| -- no normalised sample of `count` can exceed sqrt(count - 1), so this bounds every
 * |BatchNorm output| of the tensor BY CONSTRUCTION; the SC_TERMS_F16X2 kernels derive their activation scale from it (xbound). */
int sc_bn_finalize(const float* stats, int nrows, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   int training, float* cst_fwd, int C, double* scratch, float* act_bound, sc_stream stream);
/* sums[row][C][2] = { sum g_bn, sum g_bn * xhat } over the row's pixels, g_bn = g * act'(BN(y));
 * rows = sc_stat_rows(SC_STAT_BNBWD, N, H, W).
 * absmax (optional, device float, zeroed by the caller before the first launch of a step): raised to
 * max_c |gamma_c * invstd_c| * max |g_bn| by an order-independent atomic max -- the range hint of the SC_TERMS_F16X2 kernels.
 * act_absmax (optional, device float, NEVER lowered -- a sticky record): raised to max |act(BN(y))|, the largest activation
 * value a consumer of this tensor stages; the two-fp16-term kernels need it below 32752 (HyperStarcopUNet.split_range_report) */
int sc_bn_bwd_reduce(const float* g, const float* y, const float* cst_fwd, int act,
                     double* sums, int N, int C, int HW, float* absmax, float* act_absmax, sc_stream stream);
/* dgamma, dbeta and the SC_SRC_BNBWD constants {scale, shift, A, B, D}  (sc_bn_bwd_finalize_rows32: float rows, sc_bnr_args) */
int sc_bn_bwd_finalize(const double* sums, int nrows, double count, const float* cst_fwd,
                       float* dgamma, float* dbeta, float* cst_bwd, int C, sc_stream stream);
int sc_bn_bwd_finalize_rows32(const float* rows, int nrows, double count, const float* cst_fwd,
                       float* dgamma, float* dbeta, float* cst_bwd, int C, double* scratch /* 64 * 2 * C doubles or NULL: coalesced pre-reduction from 4096 rows up */, sc_stream stream);
/* both steps in one launch for few-pixel layers (one block per channel over all N*HW elements); same outputs */
int sc_bn_bwd_small(const float* g, const float* y, const float* cst_fwd, int act, int N, int C, int HW,
                    float* dgamma, float* dbeta, float* cst_bwd, float* absmax, float* act_absmax, sc_stream stream);
/* out = v(a) + v(b)   (residual add of an inverted-residual block; b may be NULL) */
int sc_add_srcs(const sc_src* a, const sc_src* b, float* out, int N, int C, int HW, sc_stream stream);
/* `waiter` does not run anything queued after this call before everything queued on `signaller` up to this call has finished
 * (both streams of the SAME device; the calling thread owns both).  What torch.cuda.Stream.wait_stream does, with an event created
 * with hipEventDisableTiming | hipEventDisableSystemFence: no system-scope cache write-back at the fork / join points of the
 * weight-gradient stream (HyperStarcopUNet's backward, the counterpart of autograd's multi-stream backward under
 * starcop/models/model_module.py:69-88).  Not for ordering against the host or another device. */
int sc_stream_wait_stream(sc_stream waiter, sc_stream signaller);

/* same, and *absmax (device float, never lowered) is raised to max |out|: the range evidence for residual sums that feed a
 * two-fp16-term convolution without a BatchNorm in between (smp skip connections taken after an inverted-residual add).
 * out may be NULL (b too): then the call only records max |v(a)| -- the inference-time range check of a BatchNorm'd tensor */
int sc_add_srcs_absmax(const sc_src* a, const sc_src* b, float* out, int N, int C, int HW, float* absmax, sc_stream stream);
/* out[n,c,y,x] (+)= sum of the 2x2 block of in (backward of nearest x2 upsample) */
int sc_downsum2x2(const float* in, float* out, int accum, int N, int C, int Hout, int Wout,
                  sc_stream stream);
/* the two resampling ops of the reference's in-repo UNet (starcop/models/architectures/unet.py:15,35-43), each reading its input
 * through the source's prologue (bias + ReLU of the producing convolution):
 *   sc_maxpool2x2          : nn.MaxPool2d(2)            in [N,C,2*Hout,2*Wout] -> out [N,C,Hout,Wout]
 *   sc_upsample_bilinear2x : F.interpolate(scale_factor=2, mode='bilinear', align_corners=True)   in [N,C,Hin,Win] -> out [N,C,2Hin,2Win] */
int sc_maxpool2x2(const sc_src* in, float* out, int N, int C, int Hout, int Wout, sc_stream stream);
int sc_upsample_bilinear2x(const sc_src* in, float* out, int N, int C, int Hin, int Win, sc_stream stream);
/* their backward passes (autograd of the same two torch ops):
 *   sc_maxpool2x2_bwd          : gin[window] (+)= gpool at the FIRST maximum of each 2x2 window of v(in) (row-major order, as
 *                                max_pool2d_with_indices), 0 elsewhere;  in [N,C,2Hout,2Wout], gpool [N,C,Hout,Wout]
 *   sc_upsample_bilinear2x_bwd : gin [N,C,Hin,Win] = transpose of the interpolation applied to gout [N,C,2Hin,2Win]; a gather
 *                                with the forward weights (deterministic, no atomics)                                       */
int sc_maxpool2x2_bwd(const sc_src* in, const float* gpool, float* gin, int accum, int N, int C, int Hout, int Wout,
                      sc_stream stream);
int sc_upsample_bilinear2x_bwd(const float* gout, float* gin, int N, int C, int Hin, int Win, sc_stream stream);
int sc_fill_f64(double* p, double v, size_t n, sc_stream stream);
int sc_apply_src(const sc_src* a, float* out, int N, int C, int HW, sc_stream stream);

/* ------------------------------------------------------------------------- */
/* loss (starcop/models/model_module.py:53-58,76-79): BCEWithLogits(pos_weight) * weight -> mean
 *   loss_sum : double[1] (zeroed by caller) accumulates sum of per-pixel weighted loss
 *   dlogits  : grad of mean(loss*weight) w.r.t. logits  (may be NULL)
 *   loss_px  : per-pixel unweighted loss (may be NULL)  */
int sc_bce_logits_weighted(const float* logits, const float* target, const float* weight,
                           float pos_weight, size_t n, double* loss_sum, float* dlogits,
                           float* loss_px, sc_stream stream);

/* Adam (torch.optim.Adam defaults, model_module.py:174): one fused pass over flat buffers.
 * bias corrections are computed on the host from the step count. */
int sc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                 float lr, float beta1, float beta2, float eps, float weight_decay,
                 float bias_correction1, float bias_correction2_sqrt, float grad_scale,
                 const float* hp_dev /* optional device {lr, bc1, bc2_sqrt}: overrides the host values */,
                 sc_stream stream);
/* step[0] += 1 (device int64); hp_dev = {lr_dev[0], 1-beta1^step, sqrt(1-beta2^step)}: lets a captured
 * hipGraph replay the optimiser step with the right bias corrections and a scheduler-driven lr */
int sc_adam_prepare(int64_t* step, const float* lr_dev, float beta1, float beta2, float* hp_dev,
                    sc_stream stream);

/* masks (model_module.py:124,193-212,268-269):
 *   prediction = sigmoid(logits); pred_binary = ge0 ? logits>=0 : sigmoid>0.5   (int64)
 *   differences = 2*pred_binary + (target==1) (int64, if target)
 *   tile_count[n] = sum pred_binary (int64, zeroed by caller);
 *   then sc_pred_classification: cls[n] = count > 10*H*W/64^2 */
int sc_threshold_masks(const float* logits, const float* target, int ge0, float* prediction,
                       int64_t* pred_binary, int64_t* differences, int64_t* tile_count,
                       int N, int HW, sc_stream stream);
int sc_pred_classification(const int64_t* tile_count, int64_t* cls, int N, int H, int W,
                           sc_stream stream);

/* ------------------------------------------------------------------------- */
/* mag1c matched filters (starcop/models/mag1c.py: rmf :284-348, acrwl1mf :177-280), one work-group per
 * group of pixels (a detector column / column block: func_by_groups :116-174, mag1c_emit.py:58-84).
 * Pixels of a group are packed band-major: element (s, p) of group g is x[xoff[g] + s*Ppad[g] + p].
 * All statistics, the Cholesky factorisation and the per-pixel filter run in fp64 whatever the element type.
 * The covariance of the target-removed data is formed from the fixed scatter matrix of the group by the exact
 * rank-2 identity  N*C_k = C_0 - v t^T - t v^T + q t t^T  (see DESIGN.md), so X is streamed, never re-multiplied. */
typedef struct sc_mag1c_args {
  const void* x;            /* packed groups (see above)                                                  */
  int32_t x_is_f64;         /* element type of x / mf_out / albedo_out: 0 f32, 1 f64                      */
  const int64_t* xoff;      /* [G] element offsets of each group in x                                     */
  const int32_t* P;         /* [G] pixels per group                                                       */
  const int32_t* Ppad;      /* [G] row pitch (elements)                                                   */
  const int64_t* poff;      /* [G] offsets of each group in the per-pixel arrays                          */
  const uint8_t* statmask;  /* optional per-pixel (packed order) mask of pixels that enter mean/covariance */
  int32_t G, S;             /* groups, bands (S <= 128)                                                   */
  int64_t npix;             /* total packed pixels (sum of P)                                             */
  const double* templ;      /* [S] unit absorption spectrum                                               */
  int32_t num_iter;         /* reweighted-L1 iterations; -1: plain rmf()                                  */
  double alpha;             /* covariance shrinkage towards its diagonal (lerp)                           */
  double cov_update_scaling;
  int32_t albedo_override, zero_override, sparse_override, apply_scaling;
  double* work;             /* scratch: sc_mag1c_workspace_doubles(G, S, npix)                            */
  void* mf_out;             /* [npix] per-pixel outputs, element type of x                                */
  void* albedo_out;         /* [npix]                                                                     */
  int32_t* status;          /* [G] 0 ok, 1 covariance not positive definite (-> torch.linalg.LinAlgError), 2 see DIRECT mode */
  /* compute_energy of the reference (starcop/models/mag1c.py:270-275, 337-343), both null or both set:
   * energy [G][max(num_iter,0)+1] = sum of all entries of (x-mu) C_k^{-1} (x-mu)^T per group and iteration (entry 0: the rmf stage),
   * logdet [G] = P/2 * log(1 / prod diag chol C) of the rmf stage */
  double* energy;
  double* logdet;
  /* DIRECT mode (cube != NULL; fp32, S > 64, no energy, every group of at most 512 pixels -- a 512-row tile filtered per detector
   * column, func_by_groups at starcop/models/mag1c.py:116-174): the groups' pixels are read straight from the pixel-major cube through
   * pix_index (element (s, p) of group g = cube[pix_index[poff[g] + p] * S_total + band0 + s]) into the kernel's register tile, which
   * holds them for all iterations, and the results go straight to image order (scatter_mf / scatter_alb[pix_index[..]], element type
   * scatter_is_f64) -- no sc_mag1c_pack pass (one write + one read of the whole tile), no sc_scatter launches.  x / xoff / Ppad /
   * mf_out / albedo_out are unused (may be NULL); a group with P > 512 gets status 2. */
  const float* cube;
  int32_t S_total, band0;
  const int64_t* pix_index;
  void* scatter_mf; void* scatter_alb;
  int32_t scatter_is_f64;
} sc_mag1c_args;
size_t sc_mag1c_workspace_doubles(int G, int S, int64_t npix);
int sc_mag1c_groups(const sc_mag1c_args* a, sc_stream stream);
/* gather the pixels of a (H,W,S_total) band-interleaved cube into the packed band-major layout:
 * x[xoff[g] + s*Ppad[g] + p] = cube[pix_index[poff[g]+p]*S_total + band0 + s] */
int sc_mag1c_pack(const void* cube, int cube_is_f64, int S_total, int band0, int S,
                  const int64_t* pix_index, const int64_t* xoff, const int32_t* Ppad,
                  const int64_t* poff, const int32_t* P, int G, void* xpacked, int out_is_f64,
                  sc_stream stream);
/* out[pix_index[i]] = val[i]  (results back to image order; fill the rest before calling) */
/* valid[p] = all(cube[p][band0 .. band0+S) > nodata): the default pixel mask of func_by_groups
 * (starcop/models/mag1c.py:140-142, torch.all(x > NODATA, dim=-1)) in one pass over the pixel-major cube */
int sc_valid_mask(const void* cube, int cube_is_f64, int S_total, int band0, int S, double nodata, int64_t npix,
                  unsigned char* valid, sc_stream stream);
int sc_scatter(const void* val, int val_is_f64, const int64_t* pix_index, size_t n, void* out,
               int out_is_f64, sc_stream stream);
/* same with the element count in device memory (n_dev[0] <= n_max): no host synchronisation between layout and scatter */
int sc_scatter_n(const void* val, int val_is_f64, const int64_t* pix_index, const int64_t* n_dev, size_t n_max, void* out,
                 int out_is_f64, sc_stream stream);
/* valid[p] = all(cube[p][band0 .. band0+S) != fill): the EMIT driver's pixel mask (starcop/models/mag1c_emit.py:60-66,
 * torch.any(raw == fill_value, dim=-1) inverted) in one pass over the pixel-major cube */
int sc_valid_mask_ne(const void* cube, int cube_is_f64, int S_total, int band0, int S, double fill, int64_t npix,
                     unsigned char* valid, sc_stream stream);
/* Packed layout of COLUMN-STRUCTURED groups, built on the device with no sort and no host round trip.  Both drivers of the
 * reference group by detector columns: one column per group in starcop/process_aviris.py:209-219 (un-orthorectified cubes),
 * blocks of column_step columns in starcop/models/mag1c_emit.py:58-84.  Group g = columns [gcol[g], gcol[g+1]) of a
 * (rows, cols) image; its pixels are the valid ones (valid[r*cols + c] != 0) in row-major order -- the order of the
 * reference's boolean indexing; a group with <= min_keep valid pixels gets P[g] = 0 and is skipped by sc_mag1c_groups
 * (starcop/models/mag1c.py:166: groups of <= 10 pixels are not filtered).
 * Outputs (device): P[G], Ppad[G] (P rounded up to 64), poff[G] / xoff[G] (exclusive scans of P and of Ppad*S),
 * pix_index[rows*cols] (the first sum(P) entries are written), totals[2] = {sum P, sum Ppad*S}. */
int sc_mag1c_layout_columns(const unsigned char* valid, int rows, int cols, const int32_t* gcol /*[G+1]*/, int G, int S,
                            int min_keep, int32_t* P, int32_t* Ppad, int64_t* poff, int64_t* xoff, int64_t* pix_index,
                            int64_t* totals, sc_stream stream);

/* The same packed layout for ARBITRARY integer groups -- the orthorectified AVIRIS-NG cube, where the group of a pixel is
 * |GLT sample index| (starcop/process_aviris.py:211-217: samples_glt_file = abs(glt[..., 0]), 0 = no data), ids bounded by the
 * detector width: a stable counting sort on the device (per-1024-pixel-block histograms, a scan over the blocks per id, ranked
 * scatter) instead of torch.nonzero / argsort / unique_consecutive / repeat_interleave and their host round trip.
 * Group g = id value g in [0, nids): its pixels are the valid ones (valid[p] != 0, ids[p] == g) in ascending pixel index (the
 * order of the reference's boolean indexing x[groups == g]); ids outside [0, nids) are ignored; a group with <= min_keep valid
 * pixels gets P[g] = 0.  Outputs as sc_mag1c_layout_columns with G = nids; work: sc_mag1c_layout_ids_workspace_ints ints. */
size_t sc_mag1c_layout_ids_workspace_ints(int64_t npix, int nids);
int sc_mag1c_layout_ids(const unsigned char* valid, const int32_t* ids, int64_t npix, int nids, int S, int min_keep,
                        int32_t* P, int32_t* Ppad, int64_t* poff, int64_t* xoff, int64_t* pix_index, int64_t* totals,
                        int32_t* work, sc_stream stream);

/* band-ratio feature (starcop/data/feature_extration.py:37-56).  For B tiles of n pixels:
 *   sc_trimmed_sums : sums[b] = sum of x[b][i] with lower <= x <= upper, lower/upper = numpy.percentile(x[b], p / 100-p)
 *                     (exact order statistics by radix select + numpy's linear interpolation)   == np.sum(no_outliers(x, p))
 *   sc_band_ratio   : R = (c*signal - background)/(background + 1e-6), c = sum_bg/sum_sig per tile (device sums) or c_host;
 *                     pixels with signal < 1e-6 and background < 1e-6 get zero_value_out                                    */
size_t sc_trimmed_sum_workspace_bytes(int B);
int sc_trimmed_sums(const float* x, int B, size_t n, double p, double* sums, void* work, size_t work_bytes,
                    sc_stream stream);
int sc_band_ratio(const float* background, const float* signal, float* out, int B, size_t n,
                  const double* sum_bg, const double* sum_sig, float c_host, float zero_value_out,
                  sc_stream stream);
/* out = clip(x/div, lo, hi) * mult (+ nan_to_num): weight_mag1c (feature_extration.py:32-35: div 400, clip [0.1,1], mult 1)
 * and the EMIT->AVIRIS rescale (emit_tools/emit_dataset.py:62-106: mf/240 -> [0,2] * 1750, rgb/20 -> [0,2] * 60) */
int sc_clip_scale(const float* x, float* out, size_t n, float div, float lo, float hi, float mult,
                  int nan_to_num, sc_stream stream);

/* ------------------------------------------------------------------------- */
/* evaluation masks of the baselines and of run_validation.
 * Thresholded prediction with an optional binary opening by a 3x3 structuring element:
 *   starcop/baselines.py:25-27 binary_opening = dilation(erosion(x)) ; :53-57 Mag1cBaseline.apply_threshold (pred > thr, cross SE)
 * kornia.morphology.{erosion,dilation} (third-party, absent from /root/reference: restated from its published algorithm,
 * border_type='geodesic', engine='unfold'): erosion = min over the SE's set cells, pixels outside the image never lower the
 * minimum; dilation = max over the cells of the SE flipped in both axes, pixels outside never raise the maximum.
 * se_bits: bit (3*r + c) set <=> SE[r][c] != 0; 0 = no morphology (plain pred > thr); the cross of the reference is 0xBA.
 *   sc_binary_opening      : out[n][h][w] (int64) = opening(pred > thr); tile_count[n] += sum(out[n]) if not NULL
 *   sc_threshold_confusion : for T <= 32 thresholds at once (validation.py:38,121-127: the PR curve; T = 1: the per-tile
 *                            matrix :106) cm[n][t][2*target + p] += 1, p = the mask above at thresholds[t] (host array),
 *                            target = (int64)target_f32 in {0, 1} (other values are counted in *invalid and skipped);
 *                            pixels with ignore[i] != 0 are skipped (validation.py:84-100 mask_from_magic).            */
#define SC_SE_CROSS 0xBA
int sc_binary_opening(const float* pred, float threshold, int se_bits, int64_t* out, int64_t* tile_count,
                      int N, int H, int W, sc_stream stream);
int sc_threshold_confusion(const float* pred, const float* target, const unsigned char* ignore,
                           const float* thresholds, int T, int se_bits, int64_t* cm, int64_t* invalid,
                           int N, int H, int W, sc_stream stream);

/* ------------------------------------------------------------------------- */
/* training-batch assembly from tiles resident in HBM: crop a window of a stored tile and apply the
 * spatial augmentations of starcop/data/datamodule.py:128-134 (kornia 0.6.7 RandomRotation(degrees=90, p=.5) ->
 * RandomHorizontalFlip(p=.5) -> RandomVerticalFlip(p=.5), applied in that order by starcop/data/dataset.py:99-102) in one
 * gather: out[b][c][y][x] = crop_b(ys, xs) with (x, y) un-flipped, then rotated about the crop centre ((w-1)/2, (h-1)/2):
 *   xs = cx + cos*(x - cx) - sin*(y - cy),  ys = cy + sin*(x - cx) + cos*(y - cy)
 * sampled bilinearly (mode 0) or at the nearest pixel (mode 1) with zeros outside the crop -- what kornia's rotate() ->
 * warp_affine -> F.grid_sample(align_corners=True, padding_mode="zeros") computes.  Items without the rotate flag are
 * exact copies.  Per-item arrays are device pointers: tile index, window offsets, cos/sin, flags (1 rotate, 2 hflip,
 * 4 vflip).  tiles: [M][C][Hs][Ws] f32, out: [B][C][h][w] f32.                                                       */
int sc_gather_augment(const float* tiles, int M, int C, int Hs, int Ws, const int32_t* tile, const int32_t* row_off,
                      const int32_t* col_off, const float* cos_t, const float* sin_t, const int32_t* flags, int B,
                      int h, int w, int mode, float* out, sc_stream stream);

/* ------------------------------------------------------------------------- */
/* HOST functions (no device involved): decoders of the on-disk sample format -- one tiled GeoTIFF per product per sample,
 * read by rasterio in the reference (starcop/data/dataset.py:66-76, written by save_cog at sampling_dataset.py:332-355).
 *   sc_tiff_lzw_decode : TIFF 6.0 LZW (GDAL's default COG compression); *written = bytes produced (<= n_out)
 *   sc_tiff_unpredict  : undo TIFF predictor 2 (integer samples) or 3 (floating point) of one decoded block of
 *                        rows x cols x spp samples of bps bytes; `out` receives little-endian samples               */
int sc_tiff_lzw_decode(const uint8_t* in_host, size_t n_in, uint8_t* out_host, size_t n_out, size_t* written_host);
int sc_tiff_unpredict(const uint8_t* in_host, int predictor, int rows, int cols, int spp, int bps, int big_endian,
                      uint8_t* out_host);

#ifdef __cplusplus
}
#endif
#endif /* STARCOP_HIP_H */
