// rans_host.cpp — bit-compatible serial range-ANS coder (host side).
//
// Replaces compressai.ans.RansEncoder.encode_with_indexes /
// RansDecoder.decode_with_indexes (CompressAI 1.2.4, compressai/cpp_exts/rans/
// rans_interface.cpp over ryg_rans rans64.h), reached in the reference through
// entropy_bottleneck.compress/decompress (codec_pipeline.py:305-306,
// codec_parallel.py:307) and gaussian_conditional.compress/decompress
// (codec_pipeline.py:426-430, codec_parallel.py:400).  CompressAI is not in the
// reference tree; the stream format is restated from its published algorithm
// (SURVEY.md §8a [RECALL]): one 64-bit-state rANS stream, 16-bit probability
// precision, 32-bit renormalisation words, symbols pushed in reverse so the
// decoder pops them forward, out-of-range symbols escaped through the last CDF
// bin followed by 4-bit bypass nibbles (count in unary-of-15, value LSB first).
// The stream is inherently serial, which is why it stays on the host while the
// GPU forms symbols and indexes (entropy.hip); the Q quality streams of one GOP
// are independent and are coded on Q threads.
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <thread>
#include <vector>
#include "../../include/pcc.h"

void pcc_set_error(const char* fmt, ...);

namespace {

constexpr uint32_t kPrecision = 16;
constexpr uint32_t kBypassBits = 4;
constexpr uint32_t kMaxBypass = (1u << kBypassBits) - 1;  // 15
constexpr uint64_t kRansL = 1ull << 31;

struct Enc {
  uint64_t x;
  uint32_t* ptr;    // next word is written at --ptr
  uint32_t* floor;  // lowest legal address
  bool overflow;

  inline void emit(uint32_t w) {
    if (ptr == floor) { overflow = true; return; }
    *--ptr = w;
  }
  inline void put(uint32_t start, uint32_t freq) {
    const uint64_t x_max = ((kRansL >> kPrecision) << 32) * freq;
    if (x >= x_max) { emit((uint32_t)x); x >>= 32; }
    x = ((x / freq) << kPrecision) + (x % freq) + start;
  }
  inline void put_bits(uint32_t val) {
    const uint32_t freq = 1u << (16 - kBypassBits);
    const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
    if (x >= x_max) { emit((uint32_t)x); x >>= 32; }
    x = (x << kBypassBits) | val;
  }
};

inline int n_nibbles(uint32_t raw) {
  int nb = 0;
  while ((raw >> (nb * kBypassBits)) != 0) ++nb;
  return nb;
}

// number of coder steps a symbol expands to (1 + escape nibbles)
inline int64_t steps_for(int32_t sym, int32_t offset, int32_t max_value) {
  int32_t value = sym - offset;
  uint32_t raw;
  if (value < 0) raw = (uint32_t)(-2 * (int64_t)value - 1);
  else if (value >= max_value) raw = (uint32_t)(2 * ((int64_t)value - max_value));
  else return 1;
  const int nb = n_nibbles(raw);
  return 1 + (nb / (int)kMaxBypass + 1) + nb;
}

int encode_stream(const int32_t* sym, const int32_t* idx, int64_t n, const int32_t* cdfs, int pitch,
                  const int32_t* sizes, const int32_t* offsets, int n_cdf, uint8_t* out, int64_t cap,
                  int64_t* len, char* err, size_t errlen) {
  int64_t steps = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t ci = idx[i];
    if (ci < 0 || ci >= n_cdf) {
      snprintf(err, errlen, "rans encode: index %d out of range at %lld", ci, (long long)i);
      return PCC_E_ARG;
    }
    steps += steps_for(sym[i], offsets[ci], sizes[ci] - 2);
  }
  std::vector<uint32_t> buf((size_t)steps + 4);
  Enc e;
  e.x = kRansL;
  e.floor = buf.data();
  e.ptr = buf.data() + buf.size();
  e.overflow = false;
  for (int64_t i = n - 1; i >= 0; --i) {
    const int32_t ci = idx[i];
    const int32_t* cdf = cdfs + (int64_t)ci * pitch;
    const int32_t max_value = sizes[ci] - 2;
    int32_t value = sym[i] - offsets[ci];
    uint32_t raw = 0;
    bool esc = false;
    if (value < 0) {
      raw = (uint32_t)(-2 * (int64_t)value - 1);
      value = max_value;
      esc = true;
    } else if (value >= max_value) {
      raw = (uint32_t)(2 * ((int64_t)value - max_value));
      value = max_value;
      esc = true;
    }
    if (esc) {
      // forward order: main, unary(n_bypass) nibbles, raw nibbles LSB first;
      // the encoder consumes that list back to front
      const int nb = n_nibbles(raw);
      for (int j = nb - 1; j >= 0; --j) e.put_bits((raw >> (j * kBypassBits)) & kMaxBypass);
      const int full = nb / (int)kMaxBypass;          // number of 15-valued nibbles
      e.put_bits((uint32_t)(nb - full * (int)kMaxBypass));
      for (int j = 0; j < full; ++j) e.put_bits(kMaxBypass);
    }
    const uint32_t start = (uint32_t)cdf[value];
    const uint32_t freq = (uint32_t)(cdf[value + 1] - cdf[value]);
    if (freq == 0) {
      snprintf(err, errlen, "rans encode: zero frequency (cdf %d, value %d)", ci, value);
      return PCC_E_ARG;
    }
    e.put(start, freq);
  }
  // flush: two words, low then high
  e.emit((uint32_t)(e.x >> 32));
  e.emit((uint32_t)(e.x >> 0));
  if (e.overflow) {
    snprintf(err, errlen, "rans encode: internal buffer overflow");
    return PCC_E_NOMEM;
  }
  const int64_t nbytes = (int64_t)((buf.data() + buf.size()) - e.ptr) * 4;
  if (nbytes > cap) {
    snprintf(err, errlen, "rans encode: output needs %lld bytes, capacity %lld", (long long)nbytes,
             (long long)cap);
    return PCC_E_NOMEM;
  }
  memcpy(out, e.ptr, (size_t)nbytes);  // little-endian u32 words, as CompressAI returns them
  *len = nbytes;
  return PCC_OK;
}

struct Dec {
  uint64_t x;
  const uint8_t* p;
  const uint8_t* end;
  bool bad;
  inline uint32_t word() {
    if (end - p < 4) { bad = true; return 0; }
    uint32_t w;
    memcpy(&w, p, 4);
    p += 4;
    return w;
  }
  inline void init() {
    const uint64_t lo = word();
    const uint64_t hi = word();
    x = lo | (hi << 32);
  }
  inline uint32_t get() const { return (uint32_t)(x & ((1u << kPrecision) - 1)); }
  inline void advance(uint32_t start, uint32_t freq) {
    const uint64_t mask = (1ull << kPrecision) - 1;
    x = (uint64_t)freq * (x >> kPrecision) + (x & mask) - start;
    if (x < kRansL) x = (x << 32) | word();
  }
  inline uint32_t get_bits() {
    const uint32_t val = (uint32_t)(x & kMaxBypass);
    x >>= kBypassBits;
    if (x < kRansL) x = (x << 32) | word();
    return val;
  }
};

}  // namespace

extern "C" int pcc_rans_encode(const int32_t* h_sym, const int32_t* h_idx, int64_t n,
                               const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                               const int32_t* h_offsets, int n_cdf, uint8_t* h_out, int64_t cap,
                               int64_t* h_len) {
  if (!h_len || n < 0 || (n > 0 && (!h_sym || !h_idx)) || !h_cdfs || !h_sizes || !h_offsets || !h_out ||
      cdf_pitch < 2 || n_cdf < 1) {
    pcc_set_error("pcc_rans_encode: bad argument");
    return PCC_E_ARG;
  }
  char err[256] = "";
  const int r = encode_stream(h_sym, h_idx, n, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_out, cap,
                              h_len, err, sizeof(err));
  if (r != PCC_OK) pcc_set_error("%s", err);
  return r;
}

extern "C" int pcc_rans_encode_multi(const int32_t* h_sym, const int32_t* h_idx, int64_t n,
                                     int n_streams, const int32_t* h_cdfs, int cdf_pitch,
                                     const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                                     uint8_t* h_out, int64_t cap_each, int64_t* h_lens) {
  if (n_streams < 1 || n_streams > 64 || !h_lens || n < 0 || (n > 0 && (!h_sym || !h_idx)) || !h_cdfs ||
      !h_sizes || !h_offsets || !h_out || cdf_pitch < 2 || n_cdf < 1) {
    pcc_set_error("pcc_rans_encode_multi: bad argument");
    return PCC_E_ARG;
  }
  std::vector<int> rc((size_t)n_streams, PCC_OK);
  std::vector<std::vector<char>> errs((size_t)n_streams, std::vector<char>(256, 0));
  auto work = [&](int s) {
    rc[s] = encode_stream(h_sym + (int64_t)s * n, h_idx + (int64_t)s * n, n, h_cdfs, cdf_pitch, h_sizes,
                          h_offsets, n_cdf, h_out + (int64_t)s * cap_each, cap_each, &h_lens[s],
                          errs[s].data(), errs[s].size());
  };
  std::vector<std::thread> th;
  for (int s = 1; s < n_streams; ++s) th.emplace_back(work, s);
  work(0);
  for (auto& t : th) t.join();
  for (int s = 0; s < n_streams; ++s)
    if (rc[s] != PCC_OK) {
      pcc_set_error("stream %d: %s", s, errs[s].data());
      return rc[s];
    }
  return PCC_OK;
}

extern "C" int pcc_rans_decode(const uint8_t* h_in, int64_t len, const int32_t* h_idx, int64_t n,
                               const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                               const int32_t* h_offsets, int n_cdf, int32_t* h_sym) {
  if (!h_in || len < 8 || n < 0 || (n > 0 && (!h_idx || !h_sym)) || !h_cdfs || !h_sizes || !h_offsets ||
      cdf_pitch < 2 || n_cdf < 1) {
    pcc_set_error("pcc_rans_decode: bad argument (len=%lld)", (long long)len);
    return len < 8 ? PCC_E_STREAM : PCC_E_ARG;
  }
  Dec d;
  d.p = h_in;
  d.end = h_in + len;
  d.bad = false;
  d.init();
  for (int64_t i = 0; i < n; ++i) {
    const int32_t ci = h_idx[i];
    if (ci < 0 || ci >= n_cdf) {
      pcc_set_error("pcc_rans_decode: index %d out of range at %lld", ci, (long long)i);
      return PCC_E_ARG;
    }
    const int32_t* cdf = h_cdfs + (int64_t)ci * cdf_pitch;
    const int32_t size = h_sizes[ci];
    const int32_t max_value = size - 2;
    const uint32_t cum = d.get();
    // first entry > cum, minus one (CompressAI: std::find_if over the CDF)
    const int32_t* it = std::upper_bound(cdf, cdf + size, (int32_t)cum);
    int32_t s = (int32_t)(it - cdf) - 1;
    if (s < 0 || s > max_value) {
      pcc_set_error("pcc_rans_decode: corrupt stream at symbol %lld", (long long)i);
      return PCC_E_STREAM;
    }
    d.advance((uint32_t)cdf[s], (uint32_t)(cdf[s + 1] - cdf[s]));
    int32_t value = s;
    if (value == max_value) {
      int32_t val = (int32_t)d.get_bits();
      int32_t n_bypass = val;
      while (val == (int32_t)kMaxBypass && !d.bad) {
        val = (int32_t)d.get_bits();
        n_bypass += val;
      }
      if (n_bypass > 8) {  // a 32-bit raw value has at most 8 nibbles
        pcc_set_error("pcc_rans_decode: corrupt escape at symbol %lld", (long long)i);
        return PCC_E_STREAM;
      }
      uint32_t raw = 0;
      for (int j = 0; j < n_bypass; ++j) raw |= d.get_bits() << (j * kBypassBits);
      value = (int32_t)(raw >> 1);
      if (raw & 1u) value = -value - 1;
      else value += max_value;
    }
    if (d.bad) {
      pcc_set_error("pcc_rans_decode: truncated stream at symbol %lld", (long long)i);
      return PCC_E_STREAM;
    }
    h_sym[i] = value + h_offsets[ci];
  }
  return PCC_OK;
}
