// Phase-filter pack of the sub-pixel decoder convolution (conv_sp.hip), shared with the one-launch batch pack (k_pack_batch in
// conv_bx3.hip).  Layout: [cout tile of 32][chunk] stages of SP_WST 16-byte entries, 8 fp16 (consecutive input channels) each:
//   entry = (((((py*2 + term)*2 + px)*2 + a)*2 + b)*2 + channel half)*32 + cout
// chunks 0 .. nk_up-1: 16 channels of the up-sampled (low-resolution) source,
//   Wph[py][px][a][b] = sum_{kh in S(py,a)} sum_{kw in S(px,b)} w[co][ci][kh][kw],  S(0,0)={0} S(0,1)={1,2} S(1,0)={0,1} S(1,1)={2}
// chunks nk_up + 4*sc + (2*qy + qx): 16 channels of the skip source seen as the low-resolution "parity plane" (qy,qx) (its pixels
//   (2i+qy, 2j+qx)): slot (py,px,a,b) reads that plane at the SAME offset (a-1+py, b-1+px) as an up-sampled channel does, with
//   the single tap kh = 2a+py+qy-1, kw = 2b+px+qx-1 (zero when outside 0..2: 9 of the 16 slots are used per parity)
// two fp16 terms of the value * 2^6 (a phase filter is a sum of up to four taps: |w| < 255 keeps it inside the fp16 range).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

constexpr float SP_SW = 64.f;
constexpr int SP_WST = 2 * 2 * 2 * 2 * 2 * 2 * 32;      // 16-byte entries per chunk (32 KB)

static inline int sp_chunks(int Cup, int Csk) { return (Cup + 15) / 16 + 4 * ((Csk + 15) / 16); }
static inline size_t sp_pack_items(int Cout, int Cup, int Csk) {
  return (size_t)((Cout + 31) / 32) * sp_chunks(Cup, Csk) * (SP_WST / 2) * 8;     // one item = both terms of one value
}

// item i -> (cout tile, chunk, py, px, a, b, half, col, j); w is the OIHW filter [Cout][Cup + Csk][3][3]
// bf = true: ONE bf16 term of the plain value (the "bf16" precision mode), stages of SP_WST / 2 entries, entry index without the term
__device__ __forceinline__ void sp_pack_item(const float* __restrict__ w, unsigned short* __restrict__ out, size_t i, int Cout, int Cup, int Csk,
                                             bool bf = false) {
  const int nku = (Cup + 15) / 16, nkt = nku + 4 * ((Csk + 15) / 16), CinTot = Cup + Csk;
  size_t r = i;
  const int j = (int)(r % 8); r /= 8;
  const int col = (int)(r % 32); r /= 32;
  const int half = (int)(r % 2); r /= 2;
  const int b = (int)(r % 2); r /= 2;
  const int a = (int)(r % 2); r /= 2;
  const int px = (int)(r % 2); r /= 2;
  const int py = (int)(r % 2); r /= 2;
  const int chunk = (int)(r % nkt);
  const int mt = (int)(r / nkt);
  const int m = mt * 32 + col;
  float v = 0.f;
  if (m < Cout) {
    if (chunk < nku) {
      const int k = chunk * 16 + half * 8 + j;
      if (k < Cup) {
        const float* wp = w + ((size_t)m * CinTot + k) * 9;
        const int kh0 = (py == 0) ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), kh1 = (py == 0) ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
        const int kw0 = (px == 0) ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kw1 = (px == 0) ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
        for (int kh = kh0; kh <= kh1; ++kh)
          for (int kw = kw0; kw <= kw1; ++kw) v += wp[kh * 3 + kw];
      }
    } else {
      const int c4 = chunk - nku, q = c4 & 3, qy = q >> 1, qx = q & 1;
      const int k = (c4 >> 2) * 16 + half * 8 + j;
      const int kh = 2 * a + py + qy - 1, kw = 2 * b + px + qx - 1;
      if (k < Csk && kh >= 0 && kh <= 2 && kw >= 0 && kw <= 2) v = w[((size_t)m * CinTot + Cup + k) * 9 + kh * 3 + kw];
    }
  }
  if (bf) {
    const size_t e = ((size_t)mt * nkt + chunk) * (SP_WST / 2) + (((((size_t)py * 2 + px) * 2 + a) * 2 + b) * 2 + half) * 32 + col;
    out[e * 8 + j] = __builtin_bit_cast(unsigned short, (__bf16)v);
    return;
  }
  const float vs = __builtin_amdgcn_fmed3f(v * SP_SW, -65504.f, 65504.f);
  const _Float16 h0 = (_Float16)vs;
  const _Float16 h1 = (_Float16)(vs - (float)h0);
  const size_t base = ((size_t)mt * nkt + chunk) * SP_WST;
  const size_t e0 = base + (((((size_t)(py * 2 + 0) * 2 + px) * 2 + a) * 2 + b) * 2 + half) * 32 + col;
  const size_t e1 = base + (((((size_t)(py * 2 + 1) * 2 + px) * 2 + a) * 2 + b) * 2 + half) * 32 + col;
  out[e0 * 8 + j] = __builtin_bit_cast(unsigned short, h0);
  out[e1 * 8 + j] = __builtin_bit_cast(unsigned short, h1);
}

// ---- data gradient w.r.t. the LOW-resolution (up-sampled) source: dprev[ci][i][j] = sum over the four parity planes (qy,qx) of dy
// (dy[co][2i'+qy][2j'+qx]) and taps (a,b) of  Wph[qy][qx][a][b][co][ci] * dyP[qy][qx][co][i - (a-1+qy)][j - (b-1+qx)]  (a stride-2 4x4
// convolution of dy; no full-resolution gradient, no 2x2 down-sum).  Layout: [cin tile of 128][chunk = 4*(16-cout chunk) + 2*qy + qx]
// stages of SP_WST entries, entry = (((((h*2 + term)*2 + mx)*2 + a)*2 + b)*2 + cout half)*32 + ci  (cin block = 2h + mx of the tile)
// VIRTUAL SKIP CHANNELS (vskip; Cup <= 64 and at most 16 skip channels: decoder.blocks.3): the two cin blocks of wave half h = 1 -- zero
// filters otherwise -- carry the skip channels' FULL-resolution gradient as 4 output parities x 16 channels: virtual channel
// v = 16 * (2 oy + ox) + c is dskip[c][2i + oy][2j + ox], whose tap for dy parity chunk (qy, qx) and slot (a, b) is the single filter
// entry kh = oy + 2a + qy - 1, kw = ox + 2b + qx - 1 (zero outside 0..2): one launch stages dy once for both gradients.
// SKIP TILES (skt > 0 skip channels; any Cup): the skip channels' full-resolution gradient as ADDITIONAL 128-channel tiles of the same
// launch, 32 skip channels x 4 output parities each: block (h, mx) of such a tile is output parity (oy, ox) = (h, mx) of its 32 channels,
// with the same single-tap filters as above -- the launch stages dy once for the up-sampled channels' and the skip channels' gradient.
static inline size_t spd_pack_items(int Cout, int Cup, int skt = 0) {
  return (size_t)((Cup + 127) / 128 + (skt + 31) / 32) * 4 * ((Cout + 15) / 16) * (SP_WST / 2) * 8;
}
__device__ __forceinline__ void spd_pack_item(const float* __restrict__ w, unsigned short* __restrict__ out, size_t i, int Cout, int CinTot, int Cup,
                                              bool vskip = false, bool bf = false, int skt = 0) {
  const int nkt = 4 * ((Cout + 15) / 16);
  size_t r = i;
  const int j = (int)(r % 8); r /= 8;
  const int col = (int)(r % 32); r /= 32;
  const int half = (int)(r % 2); r /= 2;
  const int b = (int)(r % 2); r /= 2;
  const int a = (int)(r % 2); r /= 2;
  const int mx = (int)(r % 2); r /= 2;
  const int h = (int)(r % 2); r /= 2;
  const int chunk = (int)(r % nkt);
  const int mt = (int)(r / nkt);
  const int q = chunk & 3, py = q >> 1, px = q & 1;
  const int co = (chunk >> 2) * 16 + half * 8 + j, ci = mt * 128 + (2 * h + mx) * 32 + col;
  float v = 0.f;
  const int ntu = (Cup + 127) / 128;
  if (skt > 0 && mt >= ntu) {
    const int oy = h, ox = mx, c = Cup + 32 * (mt - ntu) + col;
    const int kh = oy + 2 * a + py - 1, kw = ox + 2 * b + px - 1;
    if (co < Cout && c < Cup + skt && c < CinTot && kh >= 0 && kh <= 2 && kw >= 0 && kw <= 2) v = w[((size_t)co * CinTot + c) * 9 + kh * 3 + kw];
  } else if (vskip && h == 1) {
    const int vch = mx * 32 + col, par = vch >> 4, c = Cup + (vch & 15);
    const int kh = (par >> 1) + 2 * a + py - 1, kw = (par & 1) + 2 * b + px - 1;
    if (co < Cout && c < CinTot && kh >= 0 && kh <= 2 && kw >= 0 && kw <= 2) v = w[((size_t)co * CinTot + c) * 9 + kh * 3 + kw];
  } else if (co < Cout && ci < Cup) {
    const float* wp = w + ((size_t)co * CinTot + ci) * 9;
    const int kh0 = (py == 0) ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), kh1 = (py == 0) ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int kw0 = (px == 0) ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kw1 = (px == 0) ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    for (int kh = kh0; kh <= kh1; ++kh)
      for (int kw = kw0; kw <= kw1; ++kw) v += wp[kh * 3 + kw];
  }
  if (bf) {
    const size_t e = ((size_t)mt * nkt + chunk) * (SP_WST / 2) + (((((size_t)h * 2 + mx) * 2 + a) * 2 + b) * 2 + half) * 32 + col;
    out[e * 8 + j] = __builtin_bit_cast(unsigned short, (__bf16)v);
    return;
  }
  const float vs = __builtin_amdgcn_fmed3f(v * SP_SW, -65504.f, 65504.f);
  const _Float16 h0 = (_Float16)vs;
  const _Float16 h1 = (_Float16)(vs - (float)h0);
  const size_t base = ((size_t)mt * nkt + chunk) * SP_WST;
  const size_t e0 = base + (((((size_t)(h * 2 + 0) * 2 + mx) * 2 + a) * 2 + b) * 2 + half) * 32 + col;
  const size_t e1 = base + (((((size_t)(h * 2 + 1) * 2 + mx) * 2 + a) * 2 + b) * 2 + half) * 32 + col;
  out[e0 * 8 + j] = __builtin_bit_cast(unsigned short, h0);
  out[e1 * 8 + j] = __builtin_bit_cast(unsigned short, h1);
}
