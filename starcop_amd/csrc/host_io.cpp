// Host-side decoders for the on-disk sample format (starcop_amd/io_formats.py): TIFF LZW and the TIFF predictors.
// The reference reads its samples through rasterio/GDAL (starcop/data/dataset.py:66-76); GDAL's COG driver compresses with LZW
// by default, and a Python loop over LZW codes costs seconds per 512 x 512 tile, so these two loops are native.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

// TIFF 6.0 LZW (MSB-first codes, 9..12 bits, ClearCode 256, EndOfInformation 257, "early change").
// Returns 0 and the number of bytes written in *written; -1 on a corrupt stream.  Decoding stops at n_out bytes.
int sc_tiff_lzw_decode(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out, size_t* written) {
  if (!in || !out || !written) return -1;
  struct Entry { uint16_t prefix; uint8_t last; uint8_t first; uint16_t len; };
  static thread_local std::vector<Entry> tab(4096);
  for (int i = 0; i < 256; ++i) tab[i] = Entry{0xFFFF, (uint8_t)i, (uint8_t)i, 1};
  int next = 258, bits = 9;
  uint32_t acc = 0; int nacc = 0;
  size_t ip = 0, op = 0;
  int prev = -1;
  while (op < n_out) {
    while (nacc < bits && ip < n_in) { acc = (acc << 8) | in[ip++]; nacc += 8; }
    if (nacc < bits) break;
    const int code = (int)((acc >> (nacc - bits)) & ((1u << bits) - 1));
    nacc -= bits;
    if (code == 257) break;
    if (code == 256) { next = 258; bits = 9; prev = -1; continue; }
    int emit;
    if (prev < 0) {
      if (code >= 256) return -1;
      emit = code;
    } else if (code < next) {
      emit = code;
      if (next < 4096) { tab[next] = Entry{(uint16_t)prev, tab[code].first, tab[prev].first, (uint16_t)(tab[prev].len + 1)}; ++next; }
    } else if (code == next && next < 4096) {
      tab[next] = Entry{(uint16_t)prev, tab[prev].first, tab[prev].first, (uint16_t)(tab[prev].len + 1)};
      emit = next; ++next;
    } else {
      return -1;
    }
    const size_t len = tab[emit].len;
    size_t end = op + len;
    int c = emit;
    for (size_t k = 0; k < len; ++k) {          // the string is stored back to front through the prefix chain
      const size_t pos = end - 1 - k;
      if (pos < n_out) out[pos] = tab[c].last;
      c = tab[c].prefix;
    }
    op = end < n_out ? end : n_out;
    prev = emit;
    if (next + 1 >= (1 << bits) && bits < 12) ++bits;      // early change: widen one code before the table fills the width
  }
  *written = op;
  return 0;
}

// Undo TIFF predictor 2 (horizontal differencing of little-endian integer samples, in place semantics on `out`) or
// predictor 3 (floating point: byte planes most-significant first, byte-wise differencing with stride spp).
// `in`: rows x cols x spp samples of `bps` bytes as stored; `out`: the same samples in LITTLE-endian byte order.
int sc_tiff_unpredict(const uint8_t* in, int predictor, int rows, int cols, int spp, int bps, int big_endian, uint8_t* out) {
  if (!in || !out || rows <= 0 || cols <= 0 || spp <= 0 || bps <= 0) return -1;
  const size_t wc = (size_t)cols * spp, row_bytes = wc * bps;
  if (predictor == 3) {
    std::vector<uint8_t> tmp(row_bytes);
    for (int r = 0; r < rows; ++r) {
      const uint8_t* src = in + (size_t)r * row_bytes;
      std::memcpy(tmp.data(), src, row_bytes);
      for (size_t i = (size_t)spp; i < row_bytes; ++i) tmp[i] = (uint8_t)(tmp[i] + tmp[i - spp]);
      uint8_t* dst = out + (size_t)r * row_bytes;
      for (size_t j = 0; j < wc; ++j)
        for (int b = 0; b < bps; ++b) dst[j * bps + (bps - 1 - b)] = tmp[(size_t)b * wc + j];
    }
    return 0;
  }
  if (predictor == 2) {
    if (big_endian) return -1;
    std::memcpy(out, in, (size_t)rows * row_bytes);
    for (int r = 0; r < rows; ++r) {
      uint8_t* row = out + (size_t)r * row_bytes;
      if (bps == 1) { for (size_t i = (size_t)spp; i < wc; ++i) row[i] = (uint8_t)(row[i] + row[i - spp]); }
      else if (bps == 2) { uint16_t* p = (uint16_t*)row; for (size_t i = (size_t)spp; i < wc; ++i) p[i] = (uint16_t)(p[i] + p[i - spp]); }
      else if (bps == 4) { uint32_t* p = (uint32_t*)row; for (size_t i = (size_t)spp; i < wc; ++i) p[i] = p[i] + p[i - spp]; }
      else if (bps == 8) { uint64_t* p = (uint64_t*)row; for (size_t i = (size_t)spp; i < wc; ++i) p[i] = p[i] + p[i - spp]; }
      else return -1;
    }
    return 0;
  }
  return -1;
}

}  // extern "C"
