// rans_gpu.hip — range-ANS coding of the latent symbols ON the GPU: the wave-interleaved stream of container
// version 1 (a flagged extension; version 0 is the reference's single stream, coded on the host: rans_host.cpp).
//
// Replaces, for containers that carry the flag, compressai.ans.RansEncoder.encode_with_indexes / RansDecoder.
// decode_with_indexes as called from entropy_bottleneck / gaussian_conditional .compress / .decompress
// (sender/encoder/codec_pipeline.py:305-306,426-430; receiver/decoder/codec_parallel.py:307,398-400).  The
// reference's stream is ONE rANS state over the channel-major [C, N] symbols: serial by construction, 2.0 ms to encode
// and 2.3 ms to decode per quality of a 1M-point frame on a host core, with the GPU idle meanwhile and a PCIe round trip
// of the symbols around it.  Here the coding step is the same rANS step on the same 16-bit CDFs with the same escape bin
// followed by 4-bit bypass nibbles, on a state half as wide — 32 bits, L = 2^16, 16-bit renormalisation words (round 3;
// rounds 1-2 kept CompressAI's 64-bit state: 8 bytes of final state per lane and chunk, which priced chunks of 64 x 512
// symbols at +4.5 % rate; a state of 32 bits flushes 4 bytes, so chunks of 64 x 256 cost the same rate and a launch —
// which lasts as long as one chunk — half the steps, each of them 32-bit arithmetic) — and the symbols are dealt to 64
// independent states per wave:
//
//   stream  = u32 'PCI2' | u32 n | u32 T | u32 n_chunks | u32 words[n_chunks] | chunk payloads (16-bit words)
//   chunk c = symbols [c 64 T, (c+1) 64 T): step t of lane l codes symbol c 64 T + 64 t + l (coalesced)
//   payload = 64 x (state lo, state hi) | block(step 0, round 0) | block(0, 1) .. | block(1, 0) ..
//   round 0 of a step is the symbol's bin, rounds 1.. are the bypass nibbles of the lanes whose symbol escaped;
//   a block holds the renormalisation words the decoder needs after that round, in ascending lane order.
//
// The encoder walks steps and rounds backwards and grows its chunk downwards from the end of a private buffer
// (ballot + mbcnt give a lane its place in a block), so the decoder reads every chunk strictly forwards.  One wave
// per chunk, four chunks per workgroup; the CDF rows (27k entries for the 64 Gaussian tables) live in LDS as uint16
// and a symbol is found by binary search there.  All integer: bit-exact against oracle/pcc_oracle.c
// (orc_rans_interleaved_*), which restates the same order sequentially.
//
// Time: a launch lasts as long as ONE chunk — T sequential steps of one wave alone on its SIMD (rounds 1-2: ~0.66 us per
// step of 64-bit arithmetic in 32-bit pieces; measured figures of this form in DESIGN.md §6b) at the ~8 cycles a lone wave
// gets per dependent instruction.  What is NOT on that chain any more: the symbol / index loads (eight
// steps in flight), the encoder's table lookups and 1 / freq (done one step ahead), the decoder's stream words (128 of
// them in two registers per lane, refilled 64 words before they are needed).  Measured and dropped: 256 buckets with a
// four-entry resolve instead of 64 buckets + binary search in the decoder (two dependent LDS reads instead of four to
// five: 0.36 -> 0.42 ms, more instructions).  Shorter chunks buy time with rate (256 B of final states per chunk).
#include "common.h"

#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

namespace {

constexpr uint32_t kMagic = 0x32494350u;  // "PCI2" little-endian
constexpr uint32_t kL = 1u << 16;         // 32-bit states, 16-bit renormalisation words
constexpr int kLanes = 64;
constexpr int kChunksPerWg = 4;
constexpr int kHeaderWords = 4;

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

}  // namespace

struct pcc_rans_dev {
  uint16_t* d_cdf = nullptr;   // all rows back to back; an entry of 65536 (end of a row) is stored as 0
  int32_t* d_row = nullptr;    // [3 n_cdf]: entry offset, length, symbol offset of every row
  uint16_t* d_lut = nullptr;   // [n_cdf][65]: lut[r][b] = largest s with cdf[s] <= b << 10 (decoder: search window)
  int n_cdf = 0;
  int64_t entries = 0;
  int device = 0;
  size_t lds_bytes() const { return ((size_t)entries * 2 + 15) / 16 * 16 + (size_t)n_cdf * 12 + (size_t)n_cdf * 65 * 2; }
};

extern "C" pcc_rans_dev* pcc_rans_dev_create(const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                                             const int32_t* h_offsets, int n_cdf) {
  if (!h_cdfs || !h_sizes || !h_offsets || n_cdf < 1 || n_cdf > 256 || cdf_pitch < 2) {
    pcc_set_error("pcc_rans_dev_create: bad argument");
    return nullptr;
  }
  std::vector<uint16_t> flat, lut;
  std::vector<int32_t> row((size_t)3 * n_cdf);
  for (int r = 0; r < n_cdf; ++r) {
    const int len = h_sizes[r];
    const int32_t* c = h_cdfs + (int64_t)r * cdf_pitch;
    bool ok = len >= 2 && len <= cdf_pitch && c[0] == 0 && c[len - 1] == 65536;
    for (int j = 0; ok && j + 1 < len; ++j) ok = c[j + 1] > c[j];
    if (!ok) {
      pcc_set_error("pcc_rans_dev_create: row %d is not a 16-bit CDF (length %d)", r, len);
      return nullptr;
    }
    row[3 * r] = (int32_t)flat.size();
    row[3 * r + 1] = len;
    row[3 * r + 2] = h_offsets[r];
    for (int j = 0; j < len; ++j) flat.push_back((uint16_t)(c[j] & 0xFFFF));
    // 64 buckets of 1024 cumulative counts: the symbol of bucket b's first count; entry 64 closes the last window
    int sidx = 0;
    for (int b = 0; b < 64; ++b) {
      while (sidx + 1 < len - 1 && c[sidx + 1] <= (b << 10)) ++sidx;
      lut.push_back((uint16_t)sidx);
    }
    lut.push_back((uint16_t)(len - 2));
  }
  pcc_rans_dev* t = new (std::nothrow) pcc_rans_dev();
  if (!t) return nullptr;
  t->n_cdf = n_cdf;
  t->entries = (int64_t)flat.size();
  if (t->lds_bytes() > 64 * 1024) {   // dynamic LDS of a launch without further attributes
    pcc_set_error("pcc_rans_dev_create: %lld CDF entries do not fit the LDS of a workgroup", (long long)t->entries);
    delete t;
    return nullptr;
  }
  (void)hipGetDevice(&t->device);
  if (hipMalloc((void**)&t->d_cdf, flat.size() * 2) != hipSuccess || hipMalloc((void**)&t->d_row, row.size() * 4) != hipSuccess ||
      hipMalloc((void**)&t->d_lut, lut.size() * 2) != hipSuccess ||
      hipMemcpy(t->d_cdf, flat.data(), flat.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(t->d_row, row.data(), row.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(t->d_lut, lut.data(), lut.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
    pcc_set_error("pcc_rans_dev_create: upload failed");
    if (t->d_cdf) (void)hipFree(t->d_cdf);
    if (t->d_row) (void)hipFree(t->d_row);
    if (t->d_lut) (void)hipFree(t->d_lut);
    delete t;
    return nullptr;
  }
  return t;
}

extern "C" void pcc_rans_dev_destroy(pcc_rans_dev* t) {
  if (!t) return;
  (void)hipFree(t->d_cdf);
  (void)hipFree(t->d_row);
  (void)hipFree(t->d_lut);
  delete t;
}

// steps per chunk for an array of n symbols.  A launch lasts as long as ONE chunk (its steps are sequential), and every
// chunk costs 256 B of final states: whole arrays of up to 32768 symbols are one chunk; up to 262144 symbols (the z
// string of a large GOP, the y string of a small one) chunks of 64 x 80; above, chunks of 64 x 320: +4.4 % on the
// 1M-point frame's y strings (2.4 bits per symbol), of which 0.5 % is the narrow state's coding loss (L equals the
// 16-bit probability scale); measured on that frame: T = 256: +5.3 %, 0.21 ms per direction; T = 384: +3.7 %, 0.30 ms
static inline int64_t steps_for(int64_t n) {
  return n > 262144 ? 320 : n > 32768 ? 80 : std::max<int64_t>((n + kLanes - 1) / kLanes, 1);
}
static inline int64_t chunks_for(int64_t n, int64_t T) { return std::max<int64_t>((n + kLanes * T - 1) / (kLanes * T), 1); }

extern "C" int64_t pcc_rans_dev_bound(int64_t n) {
  if (n < 0) return 0;
  const int64_t T = steps_for(n), nc = chunks_for(n, T);
  // header + per chunk: states + up to one 16-bit word per coding round (a symbol that escapes has up to 10 rounds)
  return 4 * (kHeaderWords + nc) + 2 * (nc * (2 * kLanes + kLanes * T * 11)) + 4;
}

// ---- shared device pieces -------------------------------------------------------------------------------------
struct RansView {
  const uint16_t* cdf;
  const int32_t* row;
  const uint16_t* lut;
  int n_cdf;
  int64_t entries;
};

// LDS image: uint16 cdf[entries] (padded to 16 B) | int32 row[3 n_cdf] | uint16 lut[65 n_cdf] (decoder only)
__device__ __forceinline__ void stage_tables(const RansView& v, uint16_t* s_cdf, int32_t* s_row, uint16_t* s_lut) {
  // 16-B pieces of the CDF image, then the small tables
  const int64_t n16 = (v.entries * 2 + 15) / 16;
  const uint4* src = reinterpret_cast<const uint4*>(v.cdf);
  uint4* dst = reinterpret_cast<uint4*>(s_cdf);
  for (int64_t i = threadIdx.x; i < n16 - 1; i += blockDim.x) dst[i] = src[i];
  for (int64_t i = (n16 - 1) * 8 + threadIdx.x; i < v.entries; i += blockDim.x) s_cdf[i] = v.cdf[i];
  for (int i = threadIdx.x; i < 3 * v.n_cdf; i += blockDim.x) s_row[i] = v.row[i];
  if (s_lut)
    for (int i = threadIdx.x; i < 65 * v.n_cdf; i += blockDim.x) s_lut[i] = v.lut[i];
  __syncthreads();
}

__device__ __forceinline__ int lane_rank(unsigned long long bal) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
}

static RansView view_of(const pcc_rans_dev* t) { return RansView{t->d_cdf, t->d_row, t->d_lut, t->n_cdf, t->entries}; }

// ---- encoder -----------------------------------------------------------------------------------------------------
// One wave per chunk.  work: per stream and chunk a private buffer of cap_words; the chunk ends at the buffer's end.
// words_out[s * n_chunks + c] = words of the chunk (0xFFFFFFFF: the buffer was too small).
// Every global load and store of the coding loop is UNCONDITIONAL (clamped index / an offset beyond the buffer for a
// lane that has nothing to write): vmcnt retires in order, and with one conditional access in the loop the compiler
// can only wait with vmcnt(0) — for the symbol it had requested a moment ago and for every 2-byte store before it, so a
// step lasted one memory latency whatever the depth of the queue.
template <bool HAS_IDX>
__global__ __launch_bounds__(256) void k_rans_enc(RansView tv, const int32_t* __restrict__ sym,
                                                  const uint8_t* __restrict__ idx, int64_t idx_run, int64_t n,
                                                  int64_t T, int64_t n_chunks, uint16_t* __restrict__ work,
                                                  int64_t cap_words, uint32_t* __restrict__ words_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  uint16_t* s_cdf = reinterpret_cast<uint16_t*>(s_raw);
  int32_t* s_row = reinterpret_cast<int32_t*>(s_raw + ((size_t)tv.entries * 2 + 15) / 16 * 16);
  stage_tables(tv, s_cdf, s_row, nullptr);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * kChunksPerWg + wave;
  const int64_t s = blockIdx.y;
  if (c >= n_chunks) return;
  const int32_t* ssym = sym + s * n;
  const uint8_t* sidx = idx ? idx + s * n : nullptr;
  uint16_t* buf = work + (s * n_chunks + c) * cap_words;
  int64_t ptr = cap_words;   // 16-bit words [ptr, cap_words) are written
  bool overflow = false;
  uint32_t x = kL;
  const int64_t base = c * kLanes * T;

  // symbol and table index of the step after this one are requested before this step's arithmetic
  // the stream's table indexes as dwords from the aligned address below their first byte (pointer arithmetic only: a
  // pointer made from an integer is a FLAT pointer, and flat loads count on both wait counters)
  const int64_t sidx_mis = HAS_IDX ? (int64_t)((uintptr_t)sidx & 3) : 0;
  const uint32_t* sidx32 = HAS_IDX ? reinterpret_cast<const uint32_t*>(sidx - sidx_mis) : nullptr;
  auto fetch = [&](int64_t t, int32_t& sv, int& rv) {
    const int64_t i = base + t * kLanes + lane;
    const bool live = t >= 0 && i < n;
    const int64_t ic = live ? i : 0;   // n >= 1 (the entry point does not launch for an empty stream)
    // raw: nothing here may consume a loaded value (the consumer, prep, runs kEncAhead - 1 steps later) — the table
    // index comes as the aligned dword that holds its byte (a byte load is followed at once by its zero-extension)
    sv = ssym[ic];
    if constexpr (HAS_IDX) rv = (int)sidx32[(ic + sidx_mis) >> 2];
    else rv = (int)(ic / idx_run);
  };
  // the chunk's word buffer as a buffer resource: a lane without a word stores beyond it (dropped, no traffic)
  // (the chunk is the wave's: its address is the same in every lane, said to the compiler with readfirstlane — a
  // descriptor it cannot prove uniform is applied lane by lane in a loop)
  const uint64_t buf_u = (uint64_t)buf;
  uint16_t* buf_w = reinterpret_cast<uint16_t*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(buf_u >> 32)) << 32) |
                                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)buf_u));
  const __amdgpu_buffer_rsrc_t buf_rs = __builtin_amdgcn_make_buffer_rsrc(buf_w, 0, (int)(uint32_t)(cap_words * 2), 0x00027000);
  auto emit = [&](bool need) {   // the low halves of the states that renormalise, packed downwards from ptr
    const unsigned long long bal = __ballot(need);
    const int cnt = __popcll(bal);
    const bool room = ptr - cnt >= 2 * kLanes;
    overflow |= !room;
    ptr -= room ? cnt : 0;
    const uint32_t off = (need && room) ? (uint32_t)(ptr + lane_rank(bal)) * 2u : 0xFFFFFFF0u;
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)x, buf_rs, off, 0, 0);
  };
  // kEncAhead steps of symbols and table indexes are in flight: a step's own arithmetic is a few hundred cycles, a
  // load that misses L2 takes longer than that, and with one step of distance the step time WAS the memory latency.
  // The table work of a step — row, bin, escape nibbles, 1 / freq — does not depend on the coder state either: it is
  // done one step ahead (prep), so that the chain from state to state is the renormalisation ballot, one multiply by the
  // reciprocal and the remainder correction, with no LDS round trip and no division on it.
  constexpr int kEncAhead = 8;
  int32_t sv_q[kEncAhead];
  int rv_q[kEncAhead];
#pragma unroll
  for (int d = 0; d < kEncAhead; ++d) fetch(T - 1 - d, sv_q[d], rv_q[d]);
  struct Prep {
    int32_t start, freq, nb;
    uint32_t raw;
    bool esc, act;
    float rinv;
  };
  // Without a branch (round 4): a lone wave pays for the NUMBER of instructions it issues, and a divergent branch costs it
  // more than the handful it would skip (octree2.hip has the measurements).  The lanes of a step that codes nothing run
  // the same instructions on the first symbol's values and are masked where the state is updated.
  auto prep = [&](int64_t t, int32_t sv, int r) -> Prep {
    Prep p;
    const int64_t i = base + t * kLanes + lane;
    p.act = t >= 0 && i < n;
    if constexpr (HAS_IDX) r = (int)(((uint32_t)r >> (8 * (int)((i + sidx_mis) & 3))) & 0xFFu);
    r = p.act ? r : 0;
    r = r < tv.n_cdf ? r : tv.n_cdf - 1;   // a table index out of range never reaches the LDS tables (the entry points check what they can)
    const int off = s_row[3 * r], len = s_row[3 * r + 1];
    const int32_t max_value = len - 2;
    const int32_t v0 = sv - s_row[3 * r + 2];
    const bool neg = v0 < 0, big = v0 >= max_value;
    p.raw = neg ? (uint32_t)(-2 * (int64_t)v0 - 1) : (big ? (uint32_t)(2 * ((int64_t)v0 - max_value)) : 0u);
    p.esc = p.act && (neg || big);            // (v0 == max_value is `big` with raw 0: the escape bin followed by a count of 0)
    const int32_t v = (neg || big) ? max_value : v0;
    p.nb = p.raw ? (35 - __clz((int)p.raw)) >> 2 : 0;   // 4-bit nibbles of raw: at most 8
    const uint32_t c0 = s_cdf[off + v], c1 = s_cdf[off + v + 1];
    p.start = (int32_t)c0;
    const int32_t f = (int32_t)((c1 - c0) & 0xFFFFu);
    p.freq = f == 0 ? 65536 : f;
    // 1 / freq in single precision is enough for a quotient below 2^16 that the remainder corrects by one (the relative
    // errors of (float)x, of v_rcp_f32 and of the product add up to ~2^-22: 2^-6 of a unit of the quotient) — the
    // double-precision division this replaces was 16 of a step's instructions
    p.rinv = __frcp_rn((float)p.freq);
    return p;
  };
  Prep cur = prep(T - 1, sv_q[0], rv_q[0]);
  for (int64_t t0 = T - 1; t0 >= 0; t0 -= kEncAhead) {
#pragma unroll
   for (int d = 0; d < kEncAhead; ++d) {
    const int64_t t = t0 - d;
    if (t < 0) break;   // wave-uniform
    // the queue slot of step t is free (prep(t) consumed it a step ago): request step t - kEncAhead into it, then
    // prepare step t - 1 from the slot behind it
    fetch(t - kEncAhead, sv_q[d], rv_q[d]);
    const Prep nxt = prep(t - 1, sv_q[(d + 1) % kEncAhead], rv_q[(d + 1) % kEncAhead]);
    const bool act = cur.act, esc = cur.esc;
    const int32_t start = cur.start, freq = cur.freq, nb = cur.nb;
    const uint32_t raw = cur.raw;
    // bypass rounds of the escaped lanes, last round first: round 1 = nibble count, round 2 + j = nibble j
    if (__ballot(esc) != 0ull) {
      for (int r = 9; r >= 1; --r) {
        const bool in = esc && r <= 1 + nb;
        if (__ballot(in) == 0ull) continue;
        const bool need = in && x >= (1u << 28);   // ((L >> 16) << 16) * 2^12
        emit(need);
        if (need) x >>= 16;
        if (in) {
          const uint32_t val = r == 1 ? (uint32_t)nb : (raw >> (4 * (r - 2))) & 15u;
          x = (x << 4) | val;
        }
      }
    }
    {  // round 0: the symbol's bin
      const bool need = act && (uint64_t)x >= ((uint64_t)(uint32_t)freq << 16);   // ((L >> 16) << 16) * freq (freq may be 2^16)
      emit(need);
      x = need ? x >> 16 : x;
      // x / freq, x % freq with x < 2^16 freq: the quotient from one multiplication by 1 / freq (quotient below 2^16: off by
      // at most one either way), corrected by the remainder — no division on the chain from state to state, no branch
      const uint32_t f = (uint32_t)freq;
      const uint32_t q0 = (uint32_t)((float)x * cur.rinv);
      const int32_t r0 = (int32_t)(x - q0 * f);                  // in (-f, 2f): fits 32 bits (f <= 2^16)
      const bool under = r0 < 0, over = r0 >= (int32_t)f;
      const uint32_t qd = q0 - (under ? 1u : 0u) + (over ? 1u : 0u);
      const uint32_t rem = (uint32_t)(r0 + (under ? (int32_t)f : 0) - (over ? (int32_t)f : 0));
      x = act ? (qd << 16) + rem + (uint32_t)start : x;
    }
    cur = nxt;
   }
  }
  if (!overflow) {
    ptr -= 2 * kLanes;
    buf[ptr + 2 * lane] = (uint16_t)x;
    buf[ptr + 2 * lane + 1] = (uint16_t)(x >> 16);
  }
  if (lane == 0) words_out[s * n_chunks + c] = overflow ? 0xFFFFFFFFu : (uint32_t)(cap_words - ptr);
}

// header + chunk table + payloads of every stream, packed: workgroup (c, s) moves chunk c of stream s (c == n_chunks:
// writes the header); len_out[s] = bytes of stream s, or -1 when a chunk overflowed its buffer
__global__ __launch_bounds__(256) void k_rans_pack(const uint16_t* __restrict__ work, int64_t cap_words,
                                                   const uint32_t* __restrict__ words, int64_t n, int64_t T,
                                                   int64_t n_chunks, uint8_t* __restrict__ out, int64_t cap_each,
                                                   long long* __restrict__ len_out) {
  __shared__ unsigned long long s_sum[256];
  __shared__ int s_bad;
  const int64_t c = blockIdx.x, s = blockIdx.y;
  const uint32_t* w = words + s * n_chunks;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  // words in front of chunk c (all chunks for the header block), and whether any chunk overflowed
  unsigned long long part = 0;
  const int64_t upto = c < n_chunks ? c : n_chunks;
  for (int64_t j = threadIdx.x; j < n_chunks; j += blockDim.x) {
    const uint32_t v = w[j];
    if (v == 0xFFFFFFFFu) s_bad = 1;
    if (j < upto) part += v;
  }
  s_sum[threadIdx.x] = part;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) s_sum[threadIdx.x] += s_sum[threadIdx.x + d];
    __syncthreads();
  }
  const unsigned long long before = s_sum[0];
  const bool bad = s_bad != 0;
  uint32_t* o = reinterpret_cast<uint32_t*>(out + s * cap_each);
  const unsigned long long head = kHeaderWords + (unsigned long long)n_chunks;
  if (c == n_chunks) {
    const unsigned long long total = head * 4 + before * 2;   // bytes: u32 header and chunk table, 16-bit payload words
    const bool fits = !bad && (long long)total <= cap_each;
    if (threadIdx.x == 0) len_out[s] = fits ? (long long)total : -1;
    if (!fits) return;
    if (threadIdx.x == 0) {
      o[0] = kMagic;
      o[1] = (uint32_t)n;
      o[2] = (uint32_t)T;
      o[3] = (uint32_t)n_chunks;
    }
    for (int64_t j = threadIdx.x; j < n_chunks; j += blockDim.x) o[kHeaderWords + j] = w[j];
    return;
  }
  if (bad) return;
  const uint32_t cw = w[c];
  if ((long long)(head * 4 + (before + cw) * 2) > cap_each) return;   // the header block reports it
  const uint16_t* src = work + (s * n_chunks + c) * cap_words + (cap_words - cw);
  uint16_t* dst = reinterpret_cast<uint16_t*>(o + head) + before;
  for (uint32_t j = threadIdx.x; j < cw; j += blockDim.x) dst[j] = src[j];
}

// ---- decoder -----------------------------------------------------------------------------------------------------
// status (int32): OR of 1 = a chunk ran out of words, 2 = malformed escape, 4 = a table index out of range.
// As in the encoder every global access of the coding loop is unconditional (a clamped index, a buffer whose bounds
// drop what must not be written, a window load whose result is only adopted where it is needed), so that the waits
// keep the depth of the queues: with one conditional access in the loop every wait was vmcnt(0).
// The stream buffer must be readable up to the next multiple of 4 bytes (its words are fetched as aligned dwords).
template <bool HAS_IDX>
__global__ __launch_bounds__(256) void k_rans_dec(RansView tv, const uint32_t* __restrict__ in /* the stream */,
                                                  const uint8_t* __restrict__ idx, int64_t idx_run, int64_t n,
                                                  int64_t T, int64_t n_chunks, int32_t* __restrict__ sym,
                                                  int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  uint16_t* s_cdf = reinterpret_cast<uint16_t*>(s_raw);
  int32_t* s_row = reinterpret_cast<int32_t*>(s_raw + ((size_t)tv.entries * 2 + 15) / 16 * 16);
  uint16_t* s_lut = reinterpret_cast<uint16_t*>(s_row + 3 * tv.n_cdf);
  stage_tables(tv, s_cdf, s_row, s_lut);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * kChunksPerWg + wave;
  if (c >= n_chunks) return;
  // words in front of this chunk
  unsigned long long before = 0;
  for (int64_t j = lane; j < c; j += kLanes) before += in[kHeaderWords + j];
  for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
  const uint32_t cw = in[kHeaderWords + c];
  const uint16_t* p = reinterpret_cast<const uint16_t*>(in + kHeaderWords + n_chunks) + before;
  int bad = 0;
  if (cw < 2 * kLanes) {
    if (lane == 0) atomicOr(status, 1);
    return;
  }
  uint32_t x = (uint32_t)p[2 * lane] | ((uint32_t)p[2 * lane + 1] << 16);
  int64_t ptr = 2 * kLanes;
  const int64_t base = c * kLanes * T;

  // The chunk's words behind a buffer descriptor over its aligned dwords: word j is half (j + mis) & 1 of dword
  // (j + mis) >> 1, a dword past the chunk's end reads as 0.  (Wave-uniform values are said to be so with readfirstlane.)
  auto uniform_ptr = [](const void* q) -> uint64_t {
    const uint64_t u = (uint64_t)q;
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
  };
  const uint64_t p_u = uniform_ptr(p);
  const int mis = (int)((p_u >> 1) & 1);
  const uint32_t cw_r = (uint32_t)__builtin_amdgcn_readfirstlane((int)cw);
  const uint32_t cw_u = cw_r < 0x3FFFFFF0u ? cw_r : 0x3FFFFFF0u;   // the descriptor's byte count is 32 bits (a count this large is refused below: out of words)
  const __amdgpu_buffer_rsrc_t win_rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(p_u & ~(uint64_t)3), 0, (int)((((cw_u + (uint32_t)mis) * 2u) + 3u) & ~3u), 0x00027000);
  // The words of the chunk are held 128 at a time in two registers per lane — A = words [wb, wb + 64), B = the 64 behind
  // them, wb a multiple of 64 with wb <= ptr < wb + 64 — so a lane that renormalises takes its word from a register of
  // another lane (ds_bpermute).  Lane l of a window at `at` holds the DWORD with word at + l (the half is picked after
  // the shuffle).
  auto window = [&](int64_t at) -> uint32_t {
    return __builtin_amdgcn_raw_buffer_load_b32(win_rs, (uint32_t)(((at + lane + mis) >> 1) << 2), 0, 0);
  };
  int64_t wb = ptr & ~(int64_t)63;
  uint32_t win_a = window(wb), win_b = window(wb + 64);
  auto refill = [&](bool need) {
    const unsigned long long bal = __ballot(need);
    const int cnt = __popcll(bal);
    const int at = (int)(ptr - wb) + lane_rank(bal);   // 0 .. 126
    const uint32_t wa = (uint32_t)__shfl((int)win_a, at & 63, 64), wbv = (uint32_t)__shfl((int)win_b, at & 63, 64);
    const uint32_t dw = at < 64 ? wa : wbv;
    const uint32_t w = (dw >> (16 * ((at + mis) & 1))) & 0xFFFFu;   // wb is even: the parity of word wb + at is that of at
    const bool fits = ptr + cnt <= (int64_t)cw;
    bad |= (!fits && cnt) ? 1 : 0;
    x = need ? (fits ? (x << 16) | w : kL) : x;   // out of words: keep the arithmetic defined; the status word reports the stream
    ptr += fits ? cnt : 0;
    const bool cross = ptr - wb >= 64;   // wave-uniform
    wb += cross ? 64 : 0;
    win_a = cross ? win_b : win_a;   // B is stale after a crossing until request_b()
  };
  // B for the next refill: requested once per step at a place every path of the step runs through, crossing or not (the
  // same window again when not: the same words) and always adopted — no branch and no select on a value in flight, and
  // the state of the wait counter is the same on every path into the next step.  It is first read one step from here.
  auto request_b = [&]() { win_b = window(wb + 64); };
  // table indexes: kDecAhead steps in flight, as the aligned dword that holds the byte (extracted where it is used)
  constexpr int kDecAhead = 8;
  const int64_t idx_mis = HAS_IDX ? (int64_t)((uintptr_t)idx & 3) : 0;
  const uint32_t* idx32 = HAS_IDX ? reinterpret_cast<const uint32_t*>(idx - idx_mis) : nullptr;
  auto fetch_idx = [&](int64_t t) -> int {
    const int64_t i = base + t * kLanes + lane;
    const int64_t ic = (t < T && i < n) ? i : 0;
    if constexpr (HAS_IDX) return (int)idx32[(ic + idx_mis) >> 2];
    return (int)(ic / idx_run);
  };
  int r_q[kDecAhead];
#pragma unroll
  for (int d = 0; d < kDecAhead; ++d) r_q[d] = fetch_idx(d);
  // the chunk's symbols behind a descriptor: a lane past the end of the stream stores beyond it (dropped)
  const int64_t chunk_syms = n - base < kLanes * T ? n - base : kLanes * T;
  const __amdgpu_buffer_rsrc_t sym_rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(uniform_ptr(sym + base)), 0,
      __builtin_amdgcn_readfirstlane((int)(uint32_t)((chunk_syms > 0 ? chunk_syms : 0) * 4)), 0x00027000);

  for (int64_t t0 = 0; t0 < T; t0 += kDecAhead) {
#pragma unroll
   for (int d = 0; d < kDecAhead; ++d) {
    const int64_t t = t0 + d;
    if (t >= T) break;   // wave-uniform
    const int64_t i = base + t * kLanes + lane;
    const bool act = i < n;
    int r = r_q[d];
    r_q[d] = fetch_idx(t + kDecAhead);
    int32_t value = 0, max_value = 0, off_sym = 0;
    bool esc = false;
    // (round 4: this block as selects under a wave-uniform search loop, like the encoder's step, measured 174 us against
    // 169 — the decoder's step is the chain of dependent LDS reads of the search, not its instruction count)
    if (act) {
      if constexpr (HAS_IDX) r = (int)(((uint32_t)r >> (8 * (int)((i + idx_mis) & 3))) & 0xFFu);
      if (r >= tv.n_cdf) {   // reported through the status word; the tables are read at the last row instead
        bad |= 4;
        r = tv.n_cdf - 1;
      }
      const int off = s_row[3 * r], len = s_row[3 * r + 1];
      off_sym = s_row[3 * r + 2];
      max_value = len - 2;
      const uint32_t cum = (uint32_t)(x & 0xFFFFu);
      // largest s with cdf[s] <= cum, searched inside the window its 1024-count bucket gives
      int lo = s_lut[65 * r + (cum >> 10)], hi = s_lut[65 * r + (cum >> 10) + 1];
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((uint32_t)s_cdf[off + mid] <= cum) lo = mid;
        else hi = mid - 1;
      }
      const uint32_t c0 = s_cdf[off + lo], c1 = s_cdf[off + lo + 1];
      uint32_t freq = (c1 - c0) & 0xFFFFu;
      if (freq == 0) freq = 65536;
      x = freq * (x >> 16) + cum - c0;
      value = lo;
      esc = lo == max_value;
    }
    refill(act && x < kL);
    if (__ballot(esc) != 0ull) {
      // bypass rounds: round 1 reads the nibble count, the following rounds the nibbles, least significant first
      request_b();   // (the rare path requests and waits on its own)
      bool in = esc;
      int remaining = -1, j = 0;
      uint32_t raw = 0;
      while (__ballot(in) != 0ull) {
        uint32_t val = 0;
        if (in) {
          val = (uint32_t)(x & 15u);
          x >>= 4;
        }
        refill(in && x < kL);
        request_b();
        if (in) {
          if (remaining < 0) {
            if (val > 8) bad |= 2;   // a 32-bit value has at most 8 nibbles
            remaining = val > 8 ? 0 : (int)val;
          } else {
            raw |= val << (4 * j);
            ++j;
            --remaining;
          }
          if (remaining == 0) in = false;
        }
      }
      if (esc) {
        value = (int32_t)(raw >> 1);
        if (raw & 1u) value = -value - 1;
        else value += max_value;
      }
    }
    request_b();
    __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(value + off_sym), sym_rs, (uint32_t)((t * kLanes + lane) * 4), 0, 0);
   }
  }
  const unsigned long long b1 = __ballot((bad & 1) != 0), b2 = __ballot((bad & 2) != 0), b4 = __ballot((bad & 4) != 0);
  if (lane == 0 && (b1 | b2 | b4) != 0ull) atomicOr(status, (b1 ? 1 : 0) | (b2 ? 2 : 0) | (b4 ? 4 : 0));
}

// ======================================================================== C-ABI

// internal (rans_gate.h): the two launches without the read-back.  d_lens (device, int64 [n_streams]) receives the
// stream lengths, -1 for a stream that did not fit (attempt 0: room for 1.5 words per symbol — escapes are rare;
// attempt 1: the worst case).  The scratch of a call lives in the ctx arena: valid until the next call on this ctx,
// which the stream orders behind this one.
int pcc_rans_encode_dev_async(pcc_ctx* ctx, const pcc_rans_dev* tables, const int32_t* d_sym, const uint8_t* d_idx,
                              int64_t idx_run, int64_t n, int n_streams, uint8_t* d_out, int64_t cap_each,
                              long long* d_lens, int attempt) {
  PCC_REQUIRE(ctx && tables && d_out && d_lens && n >= 0 && n < ((int64_t)1 << 32) && n_streams >= 1 && n_streams <= 64 &&
                  cap_each >= 4 * (kHeaderWords + 1 + 2 * kLanes) && cap_each % 4 == 0 && (n == 0 || d_sym) &&
                  (d_idx || (idx_run >= 1 && (n + idx_run - 1) / idx_run <= tables->n_cdf)),
              PCC_E_ARG, "pcc_rans_encode_dev: bad argument (without an index array, symbol i uses table i / idx_run)");
  PCC_REQUIRE((uintptr_t)d_out % 4 == 0, PCC_E_ARG, "pcc_rans_encode_dev: output must be 4-byte aligned");
  hipStream_t st = ctx->stream;
  const int64_t T = steps_for(n), nc = chunks_for(n, T);
  PccProfScope prof(ctx, "rans_encode_dev", n, n_streams, T, nc);
  const int64_t cap_words = 2 * kLanes + (attempt == 0 ? kLanes * T * 3 / 2 + 64 : kLanes * T * 11);
  const size_t work_bytes = (size_t)n_streams * nc * cap_words * 2;   // 16-bit words
  PCC_TRY(pcc_arena_reserve(ctx, work_bytes + pcc_align((size_t)n_streams * nc * 4) + 1024));
  uint16_t* work = (uint16_t*)pcc_arena_alloc(ctx, work_bytes);
  uint32_t* words = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n_streams * nc * 4);
  if (!work || !words) return PCC_E_NOMEM;
  // an empty stream still runs one chunk (its final states): its (masked) loads read the scratch instead of a null array
  const int32_t* sym_arg = n ? d_sym : (const int32_t*)words;
  const int64_t run_arg = idx_run >= 1 ? idx_run : 1;
  if (d_idx && n)
    hipLaunchKernelGGL(k_rans_enc<true>, dim3(nblk(nc, kChunksPerWg), n_streams), dim3(256), tables->lds_bytes(), st,
                       view_of(tables), sym_arg, d_idx, run_arg, n, T, nc, work, cap_words, words);
  else
    hipLaunchKernelGGL(k_rans_enc<false>, dim3(nblk(nc, kChunksPerWg), n_streams), dim3(256), tables->lds_bytes(), st,
                       view_of(tables), sym_arg, (const uint8_t*)nullptr, run_arg, n, T, nc, work, cap_words, words);
  PCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_rans_pack, dim3((unsigned)(nc + 1), n_streams), dim3(256), 0, st, (const uint16_t*)work, cap_words,
                     (const uint32_t*)words, n, T, nc, d_out, cap_each, d_lens);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_rans_encode_dev(pcc_ctx* ctx, const pcc_rans_dev* tables, const int32_t* d_sym, const uint8_t* d_idx,
                                   int64_t idx_run, int64_t n, int n_streams, uint8_t* d_out, int64_t cap_each,
                                   int64_t* h_lens) {
  PCC_REQUIRE(ctx && h_lens && n_streams >= 1 && n_streams <= 64, PCC_E_ARG, "pcc_rans_encode_dev: bad argument");
  // the packing kernel writes the stream lengths straight into the context's pinned host block (device-visible): no
  // device allocation per call, no copy
  long long* d_lens = (long long*)ctx->pinned + 64;   // bytes 512 .. 1023 of the 4-KB block
  int rc = PCC_OK;
  for (int attempt = 0; attempt < 2; ++attempt) {
    rc = pcc_rans_encode_dev_async(ctx, tables, d_sym, d_idx, idx_run, n, n_streams, d_out, cap_each, d_lens, attempt);
    if (rc != PCC_OK) break;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
      rc = PCC_E_HIP;
      pcc_set_error("pcc_rans_encode_dev: the coder launches failed");
      break;
    }
    rc = PCC_OK;
    for (int s = 0; s < n_streams; ++s) {
      h_lens[s] = ((volatile long long*)d_lens)[s];
      if (h_lens[s] < 0) rc = PCC_E_NOMEM;
    }
    if (rc == PCC_OK) break;
    pcc_set_error("pcc_rans_encode_dev: a stream does not fit %lld bytes", (long long)cap_each);
  }
  return rc;
}

extern "C" int pcc_rans_stream_info(const uint8_t* h_in, int64_t len, int64_t* h_n, int64_t* h_steps, int64_t* h_chunks) {
  PCC_REQUIRE(h_in && len >= 4 * kHeaderWords, PCC_E_STREAM, "interleaved rANS stream: shorter than its header");
  uint32_t w[4];
  memcpy(w, h_in, 16);
  const int64_t n = w[1], T = w[2], nc = w[3];
  PCC_REQUIRE(w[0] == kMagic, PCC_E_STREAM, "interleaved rANS stream: bad magic");
  PCC_REQUIRE(T >= 1 && T <= 65536 && nc >= 1 && nc <= ((int64_t)1 << 24) && (int64_t)kLanes * T * nc >= n &&
                  (int64_t)kLanes * T * (nc - 1) < std::max<int64_t>(n, 1),
              PCC_E_STREAM, "interleaved rANS stream: %lld symbols in %lld chunks of 64 x %lld", (long long)n, (long long)nc,
              (long long)T);
  PCC_REQUIRE(len >= 4 * (kHeaderWords + nc), PCC_E_STREAM, "interleaved rANS stream: truncated chunk table");
  int64_t total = 4 * (kHeaderWords + nc);   // bytes: u32 header and chunk table, then 16-bit payload words
  for (int64_t c = 0; c < nc; ++c) {
    uint32_t cw;
    memcpy(&cw, h_in + 4 * (kHeaderWords + c), 4);
    PCC_REQUIRE(cw >= 2 * kLanes, PCC_E_STREAM, "interleaved rANS stream: chunk %lld has no states", (long long)c);
    total += 2 * (int64_t)cw;
  }
  PCC_REQUIRE(total == len, PCC_E_STREAM, "interleaved rANS stream: chunks take %lld bytes, stream has %lld",
              (long long)total, (long long)len);
  if (h_n) *h_n = n;
  if (h_steps) *h_steps = T;
  if (h_chunks) *h_chunks = nc;
  return PCC_OK;
}

extern "C" int pcc_rans_decode_dev(pcc_ctx* ctx, const pcc_rans_dev* tables, const uint8_t* d_in, int64_t len,
                                   int64_t n, int64_t steps, int64_t n_chunks, const uint8_t* d_idx, int64_t idx_run,
                                   int32_t* d_sym, int32_t* d_status) {
  PCC_REQUIRE(ctx && tables && d_in && d_status && n >= 0 && (n == 0 || d_sym) && (d_idx || idx_run >= 1) && steps >= 1 &&
                  n_chunks >= 1 && len >= 4 * (kHeaderWords + n_chunks) && (int64_t)kLanes * steps * n_chunks >= n,
              PCC_E_ARG, "pcc_rans_decode_dev: bad argument");
  PCC_REQUIRE((uintptr_t)d_in % 4 == 0, PCC_E_ARG, "pcc_rans_decode_dev: stream must be 4-byte aligned");
  PccProfScope prof(ctx, "rans_decode_dev", n, 1, steps, n_chunks);
  const int64_t run_arg = idx_run >= 1 ? idx_run : 1;
  if (d_idx && n)
    hipLaunchKernelGGL(k_rans_dec<true>, dim3(nblk(n_chunks, kChunksPerWg)), dim3(256), tables->lds_bytes(), ctx->stream,
                       view_of(tables), (const uint32_t*)d_in, d_idx, run_arg, n, steps, n_chunks, d_sym, d_status);
  else
    hipLaunchKernelGGL(k_rans_dec<false>, dim3(nblk(n_chunks, kChunksPerWg)), dim3(256), tables->lds_bytes(), ctx->stream,
                       view_of(tables), (const uint32_t*)d_in, (const uint8_t*)nullptr, run_arg, n, steps, n_chunks, d_sym,
                       d_status);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}
