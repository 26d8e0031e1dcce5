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
//
// Two element widths cross the boundary: int32 symbols / int32 indexes (generic)
// and int16 symbols / uint8 indexes (what the device kernels emit for the y
// latents: 3 instead of 8 bytes per symbol over PCIe).  The decoder resolves the
// symbol with a 1024-entry per-table lookup on the top 10 bits of the cumulative
// frequency followed by a short forward scan (same result as CompressAI's linear
// std::find_if over the CDF); the encoder works from per-symbol entries holding
// start, frequency, its exact reciprocal and the renormalisation threshold.
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <memory>
#include <thread>
#include <vector>
#include "../../include/pcc.h"
#include "rans_gate.h"

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
  // one coding step from a precomputed per-symbol entry (one 32-byte load)
  inline void put(const struct EncSym& e);
  inline void put_bits(uint32_t val) {
    const uint32_t freq = 1u << (16 - kBypassBits);
    const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
    if (x >= x_max) { emit((uint32_t)x); x >>= 32; }
    x = (x << kBypassBits) | val;
  }
};

// Per-symbol encoder entry.  The division x / freq is a multiply by ceil(2^(shift+63) / freq) with
// shift = ceil(log2 freq) (Alverson, "Integer division using reciprocals"; the form ryg_rans's
// Rans64EncSymbol uses): exact for every state the coder can hold (x < 2^63; checked against x / freq
// for all 65535 frequencies on 1.3e8 states incl. both ends of the renormalisation interval), and the
// update (x / freq) << 16 + x % freq + start becomes x + bias + q * (2^16 - freq).
struct EncSym {
  uint64_t rcp;     // reciprocal of freq
  uint64_t x_max;   // renormalisation threshold ((L >> 16) << 32) * freq
  uint32_t bias;    // start (freq == 1: start + 2^16 - 1, with rcp = ~0, shift 0)
  uint32_t cmpl;    // 2^16 - freq
  uint32_t freq;    // 0 marks an unusable bin
  uint32_t rs;      // shift - 1
};
inline void Enc::put(const EncSym& e) {
  if (x >= e.x_max) { emit((uint32_t)x); x >>= 32; }
  const uint64_t q = (uint64_t)(((unsigned __int128)e.rcp * x) >> 64) >> e.rs;   // == x / freq
  x = x + e.bias + q * e.cmpl;
}

inline int n_nibbles(uint32_t raw) {
  int nb = 0;
  while ((raw >> (nb * kBypassBits)) != 0) ++nb;
  return nb;
}

struct EncTables {
  std::vector<int32_t> base;  // [n_cdf + 1]: first entry of every table
  std::vector<EncSym> ent;
};

int build_enc_tables(const int32_t* cdfs, int pitch, const int32_t* sizes, int n_cdf, EncTables* t, char* err,
                     size_t errlen) {
  t->base.assign((size_t)n_cdf + 1, 0);
  for (int c = 0; c < n_cdf; ++c) {
    if (sizes[c] < 2 || sizes[c] > pitch) {
      snprintf(err, errlen, "rans encode: cdf %d has length %d", c, sizes[c]);
      return PCC_E_ARG;
    }
    t->base[c + 1] = t->base[c] + sizes[c] - 1;
  }
  t->ent.resize((size_t)t->base[n_cdf]);
  for (int c = 0; c < n_cdf; ++c) {
    const int32_t* cdf = cdfs + (int64_t)c * pitch;
    for (int v = 0; v < sizes[c] - 1; ++v) {
      EncSym& e = t->ent[(size_t)t->base[c] + v];
      const int64_t f = (int64_t)cdf[v + 1] - cdf[v];
      e.freq = (f >= 1 && f <= 65536) ? (uint32_t)f : 0u;   // 0 marks an unusable bin (checked at use)
      e.bias = (uint32_t)cdf[v];
      e.cmpl = (1u << kPrecision) - e.freq;
      if (e.freq < 2) {
        e.rcp = ~0ull;
        e.rs = 0;
        e.bias += (1u << kPrecision) - 1;
      } else {
        uint32_t shift = 0;
        while (e.freq > (1u << shift)) ++shift;
        e.rcp = (uint64_t)((((unsigned __int128)1 << (shift + 63)) + e.freq - 1) / e.freq);
        e.rs = shift - 1;
      }
      e.x_max = ((kRansL >> kPrecision) << 32) * (uint64_t)e.freq;
    }
  }
  return PCC_OK;
}

template <typename SymT, typename IdxT>
int encode_stream(const SymT* sym, const IdxT* idx, int64_t n, const int32_t* cdfs, int pitch,
                  const int32_t* sizes, const int32_t* offsets, int n_cdf, uint8_t* out, int64_t cap,
                  int64_t* len, char* err, size_t errlen, const PccRansGate* gate = nullptr,
                  const EncTables* pre = nullptr, PccRansSeek* seek = nullptr) {
  // worst case per symbol: 1 main step + (1..2 unary) + 8 raw nibbles; each step emits at most
  // one 32-bit word, and in-range symbols (the common case) emit 16 bits on average.  Size the
  // staging buffer for the common case and grow on demand.
  size_t buf_words = (size_t)(n / 2 + n / 8) + 1024;
  std::unique_ptr<uint32_t[]> buf(new uint32_t[buf_words]);  // not cleared: 2 MB per 844k symbols
  // per-table symbol entries: prebuilt (a codec holds them for its lifetime) or built for this call — 27k entries
  // with a 128-bit division each for the Gaussian tables, ~0.2 ms that a per-frame call should not pay
  EncTables local;
  if (!pre) {
    const int rc = build_enc_tables(cdfs, pitch, sizes, n_cdf, &local, err, errlen);
    if (rc != PCC_OK) return rc;
    pre = &local;
  }
  const std::vector<int32_t>& base = pre->base;
  const std::vector<EncSym>& ent = pre->ent;
  for (int attempt = 0; attempt < 3; ++attempt) {
    Enc e;
    e.x = kRansL;
    e.floor = buf.get();
    e.ptr = buf.get() + buf_words;
    e.overflow = false;
    const int n_chunks = gate ? gate->n_chunks : 1;
    // seek points (PccRansSeek): the coder's state and the words it has emitted, noted when it has coded every symbol
    // from index[k] on — where a decoder stands when it has decoded the symbols in front of index[k].  The coder walks
    // the array backwards: points are met from the last to the first, between two symbols (never inside an escape).
    int sk = seek ? seek->n - 1 : -1;
    while (sk >= 0 && seek->index[sk] >= n) --sk;   // (a point at or behind the end is never met: left at state 0)
    for (int ch = 0; ch < n_chunks; ++ch) {
    const int64_t c_hi = (ch == 0 ? n : gate->bound[ch - 1]) - 1, c_lo = gate ? gate->bound[ch] : 0;
    if (gate) gate->fn(gate->user, ch);  // chunk ch has arrived (returns at once on a second attempt)
    for (int64_t i_hi = c_hi; i_hi >= c_lo;) {
    // the piece of this chunk down to the next seek point (or to the chunk's start)
    const int64_t i_lo = (sk >= 0 && seek->index[sk] > c_lo) ? std::min<int64_t>(seek->index[sk], i_hi + 1) : c_lo;
    for (int64_t i = i_hi; i >= i_lo; --i) {
      const int32_t ci = (int32_t)idx[i];
      if ((uint32_t)ci >= (uint32_t)n_cdf) {
        snprintf(err, errlen, "rans encode: index %d out of range at %lld", ci, (long long)i);
        return PCC_E_ARG;
      }
      const int32_t max_value = sizes[ci] - 2;
      int32_t value = (int32_t)sym[i] - offsets[ci];
      if ((uint32_t)value >= (uint32_t)max_value) {
        // escape: forward order is main, unary(n_bypass) nibbles, raw nibbles LSB first;
        // the encoder consumes that list back to front
        uint32_t raw;
        if (value < 0) raw = (uint32_t)(-2 * (int64_t)value - 1);
        else raw = (uint32_t)(2 * ((int64_t)value - max_value));
        value = max_value;
        const int nb = n_nibbles(raw);
        for (int j = nb - 1; j >= 0; --j) e.put_bits((raw >> (j * kBypassBits)) & kMaxBypass);
        const int full = nb / (int)kMaxBypass;  // number of 15-valued nibbles
        e.put_bits((uint32_t)(nb - full * (int)kMaxBypass));
        for (int j = 0; j < full; ++j) e.put_bits(kMaxBypass);
      }
      const EncSym& es = ent[(size_t)base[ci] + value];
      if (es.freq == 0) {
        snprintf(err, errlen, "rans encode: zero frequency (cdf %d, value %d)", ci, value);
        return PCC_E_ARG;
      }
      e.put(es);
    }
    while (sk >= 0 && seek->index[sk] >= i_lo && seek->index[sk] > 0) {   // the symbols from index[sk] on are coded
      seek->state[sk] = e.x;
      seek->word[sk] = (int64_t)((buf.get() + buf_words) - e.ptr);   // words emitted so far (turned into an offset below)
      --sk;
    }
    i_hi = i_lo - 1;
    }
    }
    // flush: two words, low then high
    e.emit((uint32_t)(e.x >> 32));
    e.emit((uint32_t)(e.x >> 0));
    if (e.overflow) {
      buf_words = (size_t)n * 12 + 1024;  // absolute worst case
      buf.reset(new uint32_t[buf_words]);
      continue;
    }
    const int64_t nbytes = (int64_t)((buf.get() + buf_words) - e.ptr) * 4;
    if (nbytes > cap) {
      snprintf(err, errlen, "rans encode: output needs %lld bytes, capacity %lld", (long long)nbytes,
               (long long)cap);
      return PCC_E_NOMEM;
    }
    memcpy(out, e.ptr, (size_t)nbytes);  // little-endian u32 words, as CompressAI returns them
    *len = nbytes;
    if (seek) {
      // the decoder reads the stream forwards: the two flush words, then the coder's words last to first.  With m words
      // emitted when point k was met and M in all, the decoder has consumed 2 + (M - m) words when it gets there
      const int64_t total = nbytes / 4;   // M + 2
      for (int k = 0; k < seek->n; ++k) {
        if (seek->index[k] <= 0 || seek->index[k] >= n) { seek->state[k] = 0; seek->word[k] = 0; continue; }
        seek->word[k] = total - seek->word[k];
      }
    }
    return PCC_OK;
  }
  snprintf(err, errlen, "rans encode: internal buffer overflow");
  return PCC_E_NOMEM;
}

template <typename SymT, typename IdxT>
int encode_multi(const SymT* h_sym, const IdxT* h_idx, int64_t n, int n_streams, const int32_t* h_cdfs,
                 int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf, uint8_t* h_out,
                 int64_t cap_each, int64_t* h_lens, const char* who) {
  if (n_streams < 1 || n_streams > 64 || !h_lens || n < 0 || (n > 0 && (!h_sym || !h_idx)) || !h_cdfs ||
      !h_sizes || !h_offsets || !h_out || cdf_pitch < 2 || n_cdf < 1) {
    pcc_set_error("%s: bad argument", who);
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
      pcc_set_error("%s stream %d: %s", who, s, errs[s].data());
      return rc[s];
    }
  return PCC_OK;
}

// Decoder tables, built once per call (a few thousand entries): sym[s] = freq << 16 | start, and a
// 1024-entry lookup on the top 10 bits of the cumulative frequency giving the first candidate symbol;
// a short forward scan finishes it (same result as CompressAI's linear std::find_if over the CDF).
struct DecTab {
  const uint32_t* sym;   // [size - 1]
  const uint64_t* lut;   // [1024]: start | freq << 16 | symbol << 32 | ambiguous << 63
  int32_t max_value, offset;
};
constexpr int kLutBits = 10;

struct DecTables {
  std::vector<DecTab> tabs;    // [n_cdf]; entries of unused CDFs stay empty
  std::vector<uint64_t> luts;  // [n_cdf << kLutBits]
  std::vector<uint32_t> syms;
};

int build_dec_tables(const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                     const uint8_t* used /*nullable: all*/, DecTables* t, const char* who) {
  t->tabs.assign((size_t)n_cdf, DecTab{nullptr, nullptr, 0, 0});
  t->luts.assign((size_t)n_cdf << kLutBits, 0ull);
  size_t total = 0;
  for (int c = 0; c < n_cdf; ++c) {
    if (used && !used[c]) continue;
    if (h_sizes[c] < 2 || h_sizes[c] > cdf_pitch) {
      pcc_set_error("%s: cdf %d has length %d", who, c, h_sizes[c]);
      return PCC_E_ARG;
    }
    total += (size_t)h_sizes[c] - 1;
  }
  t->syms.assign(total + 1, 0u);
  total = 0;
  for (int c = 0; c < n_cdf; ++c) {
    if (used && !used[c]) continue;
    const int32_t* cdf = h_cdfs + (int64_t)c * cdf_pitch;
    const int size = h_sizes[c];
    uint32_t* sy = t->syms.data() + total;
    for (int v = 0; v < size - 1; ++v) sy[v] = ((uint32_t)(cdf[v + 1] - cdf[v]) << 16) | ((uint32_t)cdf[v] & 0xFFFFu);
    // lut[b]: s = largest symbol <= max_value with cdf[s] <= first slot of bucket b, together with its
    // (start, freq) so that a bucket lying inside ONE symbol's interval resolves with this single load;
    // a bucket that a CDF boundary cuts is flagged and finishes with the forward scan
    uint64_t* lut = t->luts.data() + ((size_t)c << kLutBits);
    int s = 0;
    for (int b = 0; b < (1 << kLutBits); ++b) {
      const int lo = b << (16 - kLutBits), hi = lo + (1 << (16 - kLutBits));
      while (s + 1 < size - 1 && cdf[s + 1] <= lo) ++s;
      const bool cut = s + 1 < size - 1 && cdf[s + 1] < hi;
      lut[b] = (uint64_t)sy[s] | ((uint64_t)s << 32) | ((uint64_t)cut << 63);
    }
    t->tabs[c].sym = sy;
    t->tabs[c].lut = lut;
    t->tabs[c].max_value = size - 2;
    t->tabs[c].offset = h_offsets[c];
    total += (size_t)size - 1;
  }
  return PCC_OK;
}

// where a decoder stands between two symbols: its state and the 32-bit words of the stream it has consumed
struct DecPos {
  uint64_t x;
  int64_t word;
};
template <typename IdxT>
int decode_stream(const uint8_t* h_in, int64_t len, const IdxT* h_idx, int64_t n, const int32_t* h_cdfs,
                  int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf, int32_t* h_sym,
                  const char* who, const PccRansGate* gate = nullptr, const DecTables* pre = nullptr,
                  int64_t r_lo = 0, int64_t r_hi = -1, const DecPos* from = nullptr, DecPos* to = nullptr) {
  if (!h_in || len < 8 || n < 0 || (n > 0 && (!h_idx || !h_sym)) || !h_cdfs || !h_sizes || !h_offsets ||
      cdf_pitch < 2 || n_cdf < 1) {
    pcc_set_error("%s: bad argument (len=%lld)", who, (long long)len);
    return len < 8 ? PCC_E_STREAM : PCC_E_ARG;
  }
  // a range [r_lo, r_hi) of the symbols from a given position (seek points): no chunk gate, every CDF table prebuilt
  const bool ranged = from != nullptr;
  if (r_hi < 0) r_hi = n;
  if (ranged && (gate || r_lo < 0 || r_lo > r_hi || r_hi > n || from->word < 2 || from->word * 4 > len)) {
    pcc_set_error("%s: bad range", who);
    return PCC_E_ARG;
  }
  // ---- indexes are validated up front (keeps the check out of the serial loop)
  {
    uint32_t worst = 0;
    for (int64_t i = r_lo; i < r_hi; ++i) worst = std::max(worst, (uint32_t)(int32_t)h_idx[i]);
    if (r_hi > r_lo && worst >= (uint32_t)n_cdf) {
      for (int64_t i = r_lo; i < r_hi; ++i)
        if ((uint32_t)(int32_t)h_idx[i] >= (uint32_t)n_cdf) {
          pcc_set_error("%s: index %d out of range at %lld", who, (int32_t)h_idx[i], (long long)i);
          return PCC_E_ARG;
        }
    }
  }
  // ---- tables: prebuilt for every CDF (a codec holds them for its lifetime), or built here for the CDFs in use
  DecTables local;
  if (!pre) {
    std::vector<uint8_t> used((size_t)n_cdf, 0);
    for (int64_t i = r_lo; i < r_hi; ++i) used[(size_t)(int32_t)h_idx[i]] = 1;
    const int rc = build_dec_tables(h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, used.data(), &local, who);
    if (rc != PCC_OK) return rc;
    pre = &local;
  }
  const std::vector<DecTab>& tabs = pre->tabs;
  const uint8_t* p = h_in;
  const uint8_t* const end = h_in + len;
  auto word = [&](bool& bad) -> uint32_t {
    if (end - p < 4) { bad = true; return 0; }
    uint32_t w;
    memcpy(&w, p, 4);
    p += 4;
    return w;
  };
  bool bad = false;
  uint64_t x;
  if (ranged) {
    x = from->x;
    p = h_in + from->word * 4;
  } else {
    x = word(bad);
    x |= (uint64_t)word(bad) << 32;
  }
  const int n_chunks = gate ? gate->n_chunks : 1;
  for (int ch = 0; ch < n_chunks; ++ch) {
  const int64_t i_lo = gate ? (ch == 0 ? 0 : gate->bound[ch - 1]) : r_lo, i_hi = gate ? gate->bound[ch] : r_hi;
  for (int64_t i = i_lo; i < i_hi; ++i) {
    const DecTab& t = tabs[(size_t)(int32_t)h_idx[i]];
    const uint32_t cum = (uint32_t)(x & 0xFFFFu);
    const uint64_t l = t.lut[cum >> (16 - kLutBits)];
    int32_t s = (int32_t)((l >> 32) & 0xFFFFu);
    uint32_t e = (uint32_t)l;
    if (__builtin_expect((int64_t)l < 0, 0)) {
      // forward scan: entry s+1 starts at (sym[s+1] & 0xFFFF); == find_if(v > cum) - 1
      while (s < t.max_value && (t.sym[s + 1] & 0xFFFFu) <= cum) ++s;
      e = t.sym[s];
    }
    x = (uint64_t)(e >> 16) * (x >> kPrecision) + cum - (e & 0xFFFFu);
    if (x < kRansL) x = (x << 32) | word(bad);
    int32_t value = s;
    if (__builtin_expect(value == t.max_value, 0)) {
      auto get_bits = [&]() -> uint32_t {
        const uint32_t v = (uint32_t)(x & kMaxBypass);
        x >>= kBypassBits;
        if (x < kRansL) x = (x << 32) | word(bad);
        return v;
      };
      int32_t val = (int32_t)get_bits();
      int32_t n_bypass = val;
      while (val == (int32_t)kMaxBypass && !bad) {
        val = (int32_t)get_bits();
        n_bypass += val;
      }
      if (n_bypass > 8) {  // a 32-bit raw value has at most 8 nibbles
        pcc_set_error("%s: corrupt escape at symbol %lld", who, (long long)i);
        return PCC_E_STREAM;
      }
      uint32_t raw = 0;
      for (int j = 0; j < n_bypass; ++j) raw |= get_bits() << (j * kBypassBits);
      value = (int32_t)(raw >> 1);
      if (raw & 1u) value = -value - 1;
      else value += t.max_value;
    }
    h_sym[i] = value + t.offset;
  }
  if (gate && !bad) gate->fn(gate->user, ch);  // chunk ch is complete
  }
  if (bad) {
    pcc_set_error("%s: truncated stream", who);
    return PCC_E_STREAM;
  }
  if (to) {
    to->x = x;
    to->word = (int64_t)(p - h_in) / 4;
  }
  return PCC_OK;
}

}  // namespace

static bool gate_ok(const PccRansGate* g, int64_t n, bool descending) {
  if (!g) return true;
  if (g->n_chunks < 1 || !g->bound || !g->fn) return false;
  for (int c = 0; c < g->n_chunks; ++c) {
    const int64_t b = g->bound[c], prev = c == 0 ? (descending ? n : 0) : g->bound[c - 1];
    if (b < 0 || b > n || (descending ? b > prev : b < prev)) return false;
  }
  return g->bound[g->n_chunks - 1] == (descending ? 0 : n);
}

struct PccRansTables {
  EncTables enc;
  DecTables dec;
};

PccRansTables* pcc_rans_tables_build(const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                                     const int32_t* h_offsets, int n_cdf) {
  if (!h_cdfs || !h_sizes || !h_offsets || cdf_pitch < 2 || n_cdf < 1) {
    pcc_set_error("pcc_rans_tables_build: bad argument");
    return nullptr;
  }
  std::unique_ptr<PccRansTables> t(new PccRansTables);
  char err[256] = {0};
  if (build_enc_tables(h_cdfs, cdf_pitch, h_sizes, n_cdf, &t->enc, err, sizeof(err)) != PCC_OK) {
    pcc_set_error("pcc_rans_tables_build: %s", err);
    return nullptr;
  }
  if (build_dec_tables(h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, nullptr, &t->dec, "pcc_rans_tables_build") != PCC_OK)
    return nullptr;
  return t.release();
}

void pcc_rans_tables_free(PccRansTables* t) { delete t; }

int pcc_rans_encode16_gated(const int16_t* h_sym, const uint8_t* h_idx, int64_t n, const int32_t* h_cdfs,
                            int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                            uint8_t* h_out, int64_t cap, int64_t* h_len, const PccRansGate* gate,
                            const PccRansTables* tables) {
  if (!h_len || n < 0 || (n > 0 && (!h_sym || !h_idx)) || !h_cdfs || !h_sizes || !h_offsets || !h_out ||
      cdf_pitch < 2 || n_cdf < 1 || !gate_ok(gate, n, true)) {
    pcc_set_error("pcc_rans_encode16_gated: bad argument");
    return PCC_E_ARG;
  }
  char err[256] = {0};
  const int rc = encode_stream(h_sym, h_idx, n, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_out, cap, h_len, err,
                               sizeof(err), gate, tables ? &tables->enc : nullptr);
  if (rc != PCC_OK) pcc_set_error("pcc_rans_encode16_gated: %s", err);
  return rc;
}

int pcc_rans_encode16_seek(const int16_t* h_sym, const uint8_t* h_idx, int64_t n, const int32_t* h_cdfs,
                           int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                           uint8_t* h_out, int64_t cap, int64_t* h_len, const PccRansGate* gate,
                           const PccRansTables* tables, PccRansSeek* seek) {
  if (!h_len || n < 0 || (n > 0 && (!h_sym || !h_idx)) || !h_cdfs || !h_sizes || !h_offsets || !h_out ||
      cdf_pitch < 2 || n_cdf < 1 || !gate_ok(gate, n, true)) {
    pcc_set_error("pcc_rans_encode16_seek: bad argument");
    return PCC_E_ARG;
  }
  if (seek)
    for (int k = 0; k < seek->n; ++k)
      if (!seek->index || !seek->state || !seek->word || (k > 0 && seek->index[k] < seek->index[k - 1])) {
        pcc_set_error("pcc_rans_encode16_seek: seek indexes must ascend");
        return PCC_E_ARG;
      }
  char err[256] = {0};
  const int rc = encode_stream(h_sym, h_idx, n, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_out, cap, h_len, err,
                               sizeof(err), gate, tables ? &tables->enc : nullptr, seek);
  if (rc != PCC_OK) pcc_set_error("pcc_rans_encode16_seek: %s", err);
  return rc;
}

// symbols [i_lo, i_hi) of a stream from a seek point (state_in, word_in: 32-bit words consumed so far; i_lo == 0 with
// word_in == 0 starts at the head of the stream); *state_out / *word_out = where the decoder stands behind symbol i_hi - 1
int pcc_rans_decode8_range(const uint8_t* h_in, int64_t len, const uint8_t* h_idx, int64_t n, const int32_t* h_cdfs,
                           int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf, int32_t* h_sym,
                           const PccRansTables* tables, int64_t i_lo, int64_t i_hi, uint64_t state_in, int64_t word_in,
                           uint64_t* state_out, int64_t* word_out) {
  if (!tables || !state_out || !word_out || !h_in || len < 8) {
    pcc_set_error("pcc_rans_decode8_range: bad argument");
    return PCC_E_ARG;
  }
  DecPos from{state_in, word_in}, to{0, 0};
  if (i_lo == 0 && word_in == 0) {   // the head of the stream: its first two words are the state
    uint32_t w[2];
    memcpy(w, h_in, 8);
    from.x = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    from.word = 2;
  }
  const int rc = decode_stream(h_in, len, h_idx, n, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_sym,
                               "pcc_rans_decode8_range", nullptr, &tables->dec, i_lo, i_hi, &from, &to);
  *state_out = to.x;
  *word_out = to.word;
  return rc;
}

int pcc_rans_decode8_gated(const uint8_t* h_in, int64_t len, const uint8_t* h_idx, int64_t n, const int32_t* h_cdfs,
                           int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                           int32_t* h_sym, const PccRansGate* gate, const PccRansTables* tables) {
  if (!gate_ok(gate, n, false)) {
    pcc_set_error("pcc_rans_decode8_gated: bad chunk table");
    return PCC_E_ARG;
  }
  return decode_stream(h_in, len, h_idx, n, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_sym,
                       "pcc_rans_decode8", gate, tables ? &tables->dec : nullptr);
}

extern "C" int pcc_rans_encode(const int32_t* h_sym, const int32_t* h_idx, int64_t n,
                               const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                               const int32_t* h_offsets, int n_cdf, uint8_t* h_out, int64_t cap,
                               int64_t* h_len) {
  return encode_multi(h_sym, h_idx, n, 1, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_out, cap, h_len,
                      "pcc_rans_encode");
}

extern "C" int pcc_rans_encode_multi(const int32_t* h_sym, const int32_t* h_idx, int64_t n,
                                     int n_streams, const int32_t* h_cdfs, int cdf_pitch,
                                     const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                                     uint8_t* h_out, int64_t cap_each, int64_t* h_lens) {
  return encode_multi(h_sym, h_idx, n, n_streams, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_out,
                      cap_each, h_lens, "pcc_rans_encode_multi");
}

extern "C" int pcc_rans_encode_multi16(const int16_t* h_sym, const uint8_t* h_idx, int64_t n,
                                       int n_streams, const int32_t* h_cdfs, int cdf_pitch,
                                       const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                                       uint8_t* h_out, int64_t cap_each, int64_t* h_lens) {
  return encode_multi(h_sym, h_idx, n, n_streams, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_out,
                      cap_each, h_lens, "pcc_rans_encode_multi16");
}

extern "C" int pcc_rans_decode(const uint8_t* h_in, int64_t len, const int32_t* h_idx, int64_t n,
                               const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                               const int32_t* h_offsets, int n_cdf, int32_t* h_sym) {
  return decode_stream(h_in, len, h_idx, n, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_sym,
                       "pcc_rans_decode");
}

// op-level forms of the seek points (include/pcc.h): int32 symbols / indexes, tables built per call
extern "C" int pcc_rans_encode_seek(const int32_t* h_sym, const int32_t* h_idx, int64_t n, const int32_t* h_cdfs,
                                    int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                                    uint8_t* h_out, int64_t cap, int64_t* h_len, const int64_t* h_seek_index, int n_seek,
                                    uint64_t* h_seek_state, int64_t* h_seek_word) {
  if (!h_len || n < 0 || (n > 0 && (!h_sym || !h_idx)) || !h_cdfs || !h_sizes || !h_offsets || !h_out || cdf_pitch < 2 ||
      n_cdf < 1 || n_seek < 0 || (n_seek > 0 && (!h_seek_index || !h_seek_state || !h_seek_word))) {
    pcc_set_error("pcc_rans_encode_seek: bad argument");
    return PCC_E_ARG;
  }
  for (int k = 1; k < n_seek; ++k)
    if (h_seek_index[k] < h_seek_index[k - 1]) {
      pcc_set_error("pcc_rans_encode_seek: seek indexes must ascend");
      return PCC_E_ARG;
    }
  PccRansSeek seek{n_seek, h_seek_index, h_seek_state, h_seek_word};
  char err[256] = {0};
  const int rc = encode_stream(h_sym, h_idx, n, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_out, cap, h_len, err,
                               sizeof(err), nullptr, nullptr, n_seek ? &seek : nullptr);
  if (rc != PCC_OK) pcc_set_error("pcc_rans_encode_seek: %s", err);
  return rc;
}

extern "C" int pcc_rans_decode_range(const uint8_t* h_in, int64_t len, const int32_t* h_idx, int64_t n,
                                     const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets,
                                     int n_cdf, int32_t* h_sym, int64_t i_lo, int64_t i_hi, uint64_t state_in,
                                     int64_t word_in, uint64_t* h_state_out, int64_t* h_word_out) {
  if (!h_state_out || !h_word_out || !h_in || len < 8) {
    pcc_set_error("pcc_rans_decode_range: bad argument");
    return len < 8 && h_in ? PCC_E_STREAM : PCC_E_ARG;
  }
  DecPos from{state_in, word_in}, to{0, 0};
  if (i_lo == 0 && word_in == 0) {
    uint32_t w[2];
    memcpy(w, h_in, 8);
    from.x = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    from.word = 2;
  }
  const int rc = decode_stream(h_in, len, h_idx, n, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_sym,
                               "pcc_rans_decode_range", nullptr, nullptr, i_lo, i_hi, &from, &to);
  *h_state_out = to.x;
  *h_word_out = to.word;
  return rc;
}

extern "C" int pcc_rans_decode8(const uint8_t* h_in, int64_t len, const uint8_t* h_idx, int64_t n,
                                const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                                const int32_t* h_offsets, int n_cdf, int32_t* h_sym) {
  return decode_stream(h_in, len, h_idx, n, h_cdfs, cdf_pitch, h_sizes, h_offsets, n_cdf, h_sym,
                       "pcc_rans_decode8");
}
