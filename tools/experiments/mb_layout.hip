// Microbenchmark for DESIGN section 14's "channel-blocked layout" item: what the staging and store patterns of the full-resolution thin
// 3x3 layers (16 channels, 16 x 512 x 512, tile 8 rows x 64 columns + halo) cost by themselves, in the two layouts
//   NCHW                : eight 4-byte loads per staged pixel (one per channel plane), four 4-byte stores per output pixel and lane
//   [N][C/8][H][W][8]   : two 16-byte loads per staged pixel (its eight channels are 32 contiguous bytes), one 16-byte store
// No arithmetic beyond a sum that keeps the loads alive; the patch goes through LDS as in k_conv3_thin_h (one barrier).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mb_layout.hip -o /tmp/mb_layout && /tmp/mb_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef MB_TH
#define MB_TH 8
#endif
#ifndef MB_TW
#define MB_TW 64
#endif
constexpr int C = 16, TH = MB_TH, TW = MB_TW, PR = TH + 2, PC = TW + 2, NPX = PR * PC;      // -DMB_TH=16 / -DMB_TW=128: other tile shapes

template <bool BLOCKED>
__global__ __launch_bounds__(256) void k_stage(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W) {
  __shared__ float4 s_p[2][2][NPX];                      // [half of the 8-channel group][group][pixel]: 42 KB like the thin kernel
  const int tid = threadIdx.x, wave = tid >> 6;
  const int tiles_x = W / TW, per_img = tiles_x * (H / TH);
  const int n = blockIdx.x / per_img, tile = blockIdx.x % per_img;
  const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;
  const int grp = wave >> 1, e0 = tid - grp * 128;       // a pair of waves stages one 8-channel group, lanes = consecutive patch pixels
  const size_t plane = (size_t)H * W;
  for (int r = 0; r < (NPX + 127) / 128; ++r) {
    const int e = e0 + 128 * r;
    const int pr = e / PC, pc = e - pr * PC;
    const int y = y0 - 1 + pr, x = x0 - 1 + pc;
    const bool ok = e < NPX && y >= 0 && y < H && x >= 0 && x < W;
    const size_t off = ok ? (size_t)y * W + x : 0;
    float4 a, b;
    if (BLOCKED) {
      const float4* p = reinterpret_cast<const float4*>(in + (((size_t)n * (C / 8) + grp) * plane + off) * 8);
      a = p[0]; b = p[1];
    } else {
      const float* p = in + ((size_t)n * C + grp * 8) * plane + off;
      a = make_float4(p[0], p[plane], p[2 * plane], p[3 * plane]);
      b = make_float4(p[4 * plane], p[5 * plane], p[6 * plane], p[7 * plane]);
    }
    if (e < NPX) { s_p[0][grp][e] = ok ? a : make_float4(0, 0, 0, 0); s_p[1][grp][e] = ok ? b : make_float4(0, 0, 0, 0); }
  }
  __syncthreads();
  // outputs: the D layout of v_mfma_f32_16x16x32: lane -> pixel l15 of a 16-pixel block, channels 4*(lane>>4) .. +3; wave = 2 rows
  const int lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  constexpr int RPW = TH / 4, BPR = TW / 16;            // rows per wave, 16-pixel blocks per row
  for (int pb = 0; pb < RPW * BPR; ++pb) {
    const int row = RPW * wave + pb / BPR, col = 16 * (pb % BPR) + l15;
    const float4 c = s_p[lg & 1][lg >> 1][(row + 1) * PC + col + 1];       // stand-in for the accumulator: depends on the staged patch
    const int oy = y0 + row, ox = x0 + col;
    if (BLOCKED) {
      *reinterpret_cast<float4*>(out + (((size_t)n * (C / 8) + (lg >> 1)) * plane + (size_t)oy * W + ox) * 8 + 4 * (lg & 1)) = c;
    } else {
      float* o = out + ((size_t)n * C + 4 * lg) * plane + (size_t)oy * W + ox;
      o[0] = c.x; o[plane] = c.y; o[2 * plane] = c.z; o[3 * plane] = c.w;
    }
  }
}

int main() {
  const int N = 16, H = 512, W = 512;
  const size_t elems = (size_t)N * C * H * W;
  float *in, *out;
  hipMalloc(&in, elems * 4); hipMalloc(&out, elems * 4);
  hipMemset(in, 0, elems * 4);
  const int grid = N * (H / TH) * (W / TW);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocked = 0; blocked < 2; ++blocked) {
    for (int rep = 0; rep < 3; ++rep) {
      if (blocked) hipLaunchKernelGGL(k_stage<true>, dim3(grid), dim3(256), 0, 0, in, out, N, H, W);
      else hipLaunchKernelGGL(k_stage<false>, dim3(grid), dim3(256), 0, 0, in, out, N, H, W);
    }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int R = 20;
    for (int rep = 0; rep < R; ++rep) {
      if (blocked) hipLaunchKernelGGL(k_stage<true>, dim3(grid), dim3(256), 0, 0, in, out, N, H, W);
      else hipLaunchKernelGGL(k_stage<false>, dim3(grid), dim3(256), 0, 0, in, out, N, H, W);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= R;
    printf("tile %d x %d  %s: %.1f us per pass over 16 x 16 x 512 x 512 (read + write 2 x %.0f MB): %.2f TB/s algorithmic\n",
           TH, TW, blocked ? "[N][C/8][H][W][8]" : "NCHW             ", ms * 1e3, elems * 4 / 1e6, 2.0 * elems * 4 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
