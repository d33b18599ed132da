// What do the full-resolution thin 3x3 layers (k_conv3_thin_h: 16 planes in, 16 planes out, 16 x 512^2) pay for moving their bytes
// 4 per lane?  Two tile COPIES with that kernel's exact global access pattern (8 x 64 tiles, 10 x 66 patch of 16 planes staged through
// LDS, one output value per (cout, pixel)) and nothing else:
//   A: the kernel's pattern -- patch loads 4 B per lane in 264-byte runs, stores 4 B per lane in 64-byte runs (16 pixels of 4 couts per
//      instruction)
//   B: 16 B per lane -- interior columns as float4 (256-byte runs, four patch rows per instruction), the two halo columns as scalars;
//      stores as float4 (lane = 4 consecutive pixels of one cout: 256-byte runs of 4 couts per instruction)
//   C: B's loads with A's stores;  D: A's loads with B's stores
// build: hipcc --offload-arch=gfx950 -O3 -o thin_pattern tools/membench/thin_pattern.hip ; run: ./thin_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int CH = 16, TH = 8, TW = 64, PR = TH + 2, PC = TW + 2;

template <bool VLOAD, bool VSTORE>
__global__ __launch_bounds__(256, 2) void k_copy(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
  __shared__ __attribute__((aligned(16))) float s[CH][PR][72];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = W / TW;
  const int n = blockIdx.z, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const size_t HW = (size_t)H * W;
  const float* inn = in + (size_t)n * CH * HW;
  if (!VLOAD) {
    // a wave stages one 8-channel group's half of the patch entries; lanes = consecutive patch pixels (k_conv3_thin_h)
    const int grp = wave >> 1, e0 = tid - grp * 128;
    float xv[6][8];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int e = e0 + 128 * r, pr = e / PC, pc = e - pr * PC;
      const int y = y0 - 1 + pr, x = x0 - 1 + pc;
      const bool ok = e < PR * PC && y >= 0 && y < H && x >= 0 && x < W;
      const size_t off = ok ? (size_t)y * W + x : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[r][j] = inn[(size_t)(grp * 8 + j) * HW + off];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int e = e0 + 128 * r, pr = e / PC, pc = e - pr * PC;
      if (e < PR * PC) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s[grp * 8 + j][pr][pc + 3] = xv[r][j];
      }
    }
  } else {
    // 160 (channel, patch row) pairs; a wave request = four of them (16 lanes x 16 B each); 10 requests per wave, all in flight
    const int lr = lane >> 4, xg = x0 + 4 * (lane & 15);
    float4 v[10];
    float hv[2];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int rr = 4 * (wave + 4 * u) + lr, c = rr / PR, pr = rr - c * PR, y = y0 - 1 + pr;
      const bool ok = y >= 0 && y < H;
      v[u] = *reinterpret_cast<const float4*>(inn + (size_t)c * HW + (size_t)(ok ? y : 0) * W + xg);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = tid + 256 * h, ec = e < 320 ? e : 0, rr = ec >> 1, c = rr / PR, pr = rr - c * PR, y = y0 - 1 + pr;
      const int x = (ec & 1) ? x0 + TW : x0 - 1;
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      hv[h] = inn[(size_t)c * HW + (ok ? (size_t)y * W + x : 0)];
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int rr = 4 * (wave + 4 * u) + lr, c = rr / PR, pr = rr - c * PR;
      *reinterpret_cast<float4*>(&s[c][pr][4 + 4 * (lane & 15)]) = v[u];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = tid + 256 * h;
      if (e < 320) { const int rr = e >> 1, c = rr / PR, pr = rr - c * PR; s[c][pr][(e & 1) ? 68 : 3] = hv[h]; }
    }
  }
  __syncthreads();
  float* outn = out + (size_t)n * CH * HW;
  const int l15 = lane & 15, lg = lane >> 4;
  if (!VSTORE) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int pb = 0; pb < 8; ++pb) {
        const int co = 4 * lg + r, oy = 2 * wave + pb / 4, ox = 16 * (pb % 4) + l15;
        outn[(size_t)co * HW + (size_t)(y0 + oy) * W + x0 + ox] = s[co][oy + 1][ox + 4] + s[co][oy][ox + 3];
      }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int row = 0; row < 2; ++row) {
        const int co = 4 * lg + r, oy = 2 * wave + row, ox = 4 * l15;
        const float4 a = *reinterpret_cast<const float4*>(&s[co][oy + 1][ox + 4]);
        const float b0 = s[co][oy][ox + 3], b1 = s[co][oy][ox + 4], b2 = s[co][oy][ox + 5], b3 = s[co][oy][ox + 6];
        *reinterpret_cast<float4*>(outn + (size_t)co * HW + (size_t)(y0 + oy) * W + x0 + ox) = make_float4(a.x + b0, a.y + b1, a.z + b2, a.w + b3);
      }
  }
}

template <bool VL, bool VS>
double run(const char* name, const float* in, float* out, int N, int H, int W, std::vector<float>* keep) {
  dim3 grid((H / TH) * (W / TW), 1, N);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_copy<VL, VS>), grid, dim3(256), 0, 0, in, out, H, W);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  const int reps = 30;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_copy<VL, VS>), grid, dim3(256), 0, 0, in, out, H, W);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / reps * 1e3, bytes = 2.0 * N * CH * H * W * 4;
  std::vector<float> h(1 << 16);
  hipMemcpy(h.data(), out + (size_t)3 * H * W + 7 * W, h.size() * 4, hipMemcpyDeviceToHost);
  double cs = 0; for (float v : h) cs += v;
  printf("%-44s %8.1f us   %5.2f TB/s (read + write)   checksum %.6e\n", name, us, bytes / us / 1e6, cs);
  return us;
}

int main() {
  const int N = 16, H = 512, W = 512;
  const size_t n = (size_t)N * CH * H * W;
  float *in, *out;
  hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) * 0.001f;
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<false, false>("A  4 B loads, 4 B stores (the kernel's)", in, out, N, H, W, nullptr);
    run<true, true>("B  16 B loads, 16 B stores", in, out, N, H, W, nullptr);
    run<true, false>("C  16 B loads, 4 B stores", in, out, N, H, W, nullptr);
    run<false, true>("D  4 B loads, 16 B stores", in, out, N, H, W, nullptr);
  }
  return 0;
}
