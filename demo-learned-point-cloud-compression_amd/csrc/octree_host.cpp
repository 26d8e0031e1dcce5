// octree_host.cpp — entropy coding of octree occupancy bytes (host side).
//
// Host half of the replacement for utils.gpcc_encode / gpcc_decode
// (shared/utils.py:169-240), which in the reference shell out to MPEG's tmc3
// once per frame through ASCII PLY temp files.  tmc3 is not in the reference
// tree and its bitstream cannot be reproduced here, so the blob format below is
// this build's own (DESIGN.md "points slot"); the container only sees an opaque
// length-prefixed byte string (codec_pipeline.py:503,510), so the container
// layout is unchanged.
//
// Blob version 3 (round 4): the frame's leaves in K parts, each a complete version-1 blob under the FRAME's root cube,
// coded and decoded side by side (the serial decoder of a 26k-leaf latent was 0.21 ms at the head of every decode with
// the GPU waiting for it):
//   'O' 3 depth K | u32 n_points | i32 origin[3] | u32 payload_len | u32 len[K] | part 0 | part 1 | ..
// The leaves are cut in Morton order between grandparent cells (leaf cell >> 6), so the nodes of the two lowest inner
// levels of the parts add up to the frame's (the decoder takes the sizes of its stride-16 / stride-32 coordinate sets
// from them); the levels above are coded once per part that reaches them.  oracle/pcc_oracle.c states the rule.
//
// Blob: 'O' ver depth 0 | u32 n_points | i32 origin[3] | u32 payload_len |
// payload.  Payload = one rANS64 stream (12-bit probabilities, 32-bit words)
// of the occupancy bytes in breadth-first order (root first, Morton order
// inside a level), each byte as 8 binary decisions (octant 0..7) under an
// adaptive context = (level class, bit position, ones so far); the 8th bit is
// implied when the first seven are zero.
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <vector>
#include "../../include/pcc.h"
#include "rans_gate.h"

void pcc_set_error(const char* fmt, ...);

namespace {

constexpr uint32_t kProbBits = 12;
constexpr uint32_t kProbOne = 1u << kProbBits;
constexpr uint64_t kRansL = 1ull << 31;
constexpr int kCtxPerClass = 36;
constexpr int kClasses = 3;
constexpr int kHeader = 24;
constexpr int kMaxParts = 16;   // blob version 3: parts a decoder accepts (the encoder writes at most 8)

inline void adapt(uint16_t& p1, int bit) {
  const uint32_t p = p1, up = p + ((kProbOne - p) >> 4), dn = p - (p >> 4);
  p1 = (uint16_t)(bit ? up : dn);   // both sides computed: a conditional move, not a branch on a data bit
}
// first context of bit position j inside a level class (j (j + 1) / 2 + ones so far)
constexpr int kTri[8] = {0, 1, 3, 6, 10, 15, 21, 28};
inline int class_of(int depth, int level) {
  const int cls = depth - 1 - level;
  return cls > kClasses - 1 ? kClasses - 1 : cls;
}
// x / f for x < 2^63 and the 12-bit frequencies of the model without a division: multiply by ceil(2^(shift + 63) / f),
// shift = ceil(log2 f) (Alverson; the form rans_host.cpp uses for the y coder, checked there against x / f)
struct Recip {
  uint64_t rcp;
  uint32_t rs;
};
struct RecipTable {
  Recip r[kProbOne + 1];
  RecipTable() {
    for (uint32_t f = 1; f <= kProbOne; ++f) {
      if (f < 2) {
        r[f] = {~0ull, 0};   // q = (x * (2^64 - 1)) >> 64 = x - 1 for x >= 1: never used (a frequency is >= 15)
        continue;
      }
      uint32_t shift = 0;
      while (f > (1u << shift)) ++shift;
      r[f].rcp = (uint64_t)((((unsigned __int128)1 << (shift + 63)) + f - 1) / f);
      r[f].rs = shift - 1;
    }
    r[0] = {0, 0};
  }
};
const RecipTable kRecip;
inline void put_u32(uint8_t* p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }
inline uint32_t get_u32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// same bit interleave as csrc/common.h (x top bit of each triple), without the bias
inline uint32_t compact3(uint64_t x) {
  x &= 0x249249249249ull;
  x = (x | (x >> 2)) & 0x0C30C30C30C3ull;
  x = (x | (x >> 4)) & 0x00F00F00F00Full;
  x = (x | (x >> 8)) & 0x0000FF0000FFull;
  x = (x | (x >> 16)) & 0xFFFFull;
  return (uint32_t)x;
}

}  // namespace

static int octree_decode_cells(const uint8_t* h_in, int64_t len, std::vector<uint64_t>* cells, int64_t* h_level_n,
                               int32_t origin[3]);

extern "C" int pcc_octree_pack(const uint8_t* h_occ, const int64_t* h_level_n, int depth,
                               int64_t n_points, const int32_t* h_origin, uint8_t* h_out, int64_t cap,
                               int64_t* h_len) {
  if (!h_out || !h_len || depth < 0 || depth > 16 || n_points < 0 || cap < kHeader ||
      (n_points > 0 && (!h_occ || !h_level_n || !h_origin || depth < 1))) {
    pcc_set_error("pcc_octree_pack: bad argument");
    return PCC_E_ARG;
  }
  memset(h_out, 0, kHeader);
  h_out[0] = 'O';
  h_out[1] = 1;
  h_out[2] = (uint8_t)(n_points > 0 ? depth : 0);
  put_u32(h_out + 4, (uint32_t)n_points);
  if (n_points == 0) {
    *h_len = kHeader;
    return PCC_OK;
  }
  for (int a = 0; a < 3; ++a) put_u32(h_out + 8 + 4 * a, (uint32_t)h_origin[a]);

  // forward modelling pass
  int64_t total_nodes = 0;
  for (int L = 0; L < depth; ++L) total_nodes += h_level_n[L];
  // one record per decision: probability of a one | bit << 15 (the probability stays below 4096)
  std::unique_ptr<uint16_t[]> recs(new uint16_t[(size_t)total_nodes * 8 + 8]);
  size_t n_rec = 0;
  uint16_t model[kClasses * kCtxPerClass];
  for (auto& m : model) m = kProbOne / 2;
  int64_t pos = 0;
  for (int L = 0; L < depth; ++L) {
    uint16_t* mc = model + class_of(depth, L) * kCtxPerClass;
    for (int64_t i = 0; i < h_level_n[L]; ++i, ++pos) {
      const uint32_t byte = h_occ[pos];
      if (byte == 0) {
        pcc_set_error("pcc_octree_pack: empty occupancy byte at level %d node %lld", L, (long long)i);
        return PCC_E_ARG;
      }
      int ones = 0;
      const int last = (byte & 0x7Fu) ? 8 : 7;   // the eighth bit is implied behind seven zeros
      for (int j = 0; j < last; ++j) {
        const int bit = (byte >> j) & 1;
        uint16_t& m = mc[kTri[j] + ones];
        recs[n_rec++] = (uint16_t)(m | (bit << 15));
        adapt(m, bit);
        ones += bit;
      }
    }
  }
  // reverse rANS pass
  std::vector<uint32_t> words(n_rec / 2 + 8);
  uint32_t* end = words.data() + words.size();
  uint32_t* ptr = end;
  uint64_t x = kRansL;
  for (int64_t k = (int64_t)n_rec - 1; k >= 0; --k) {
    const uint32_t r = recs[(size_t)k], p1 = r & 0x7FFFu, bit = r >> 15;
    const uint32_t start = bit ? (kProbOne - p1) : 0u;
    const uint32_t freq = bit ? p1 : (kProbOne - p1);
    const uint64_t x_max = ((kRansL >> kProbBits) << 32) * freq;
    if (x >= x_max) {
      if (ptr == words.data()) { pcc_set_error("pcc_octree_pack: internal overflow"); return PCC_E_NOMEM; }
      *--ptr = (uint32_t)x;
      x >>= 32;
    }
    const Recip& rc = kRecip.r[freq];
    const uint64_t q = (uint64_t)(((unsigned __int128)rc.rcp * x) >> 64) >> rc.rs;   // == x / freq (x < 2^63)
    x = (q << kProbBits) + (x - q * freq) + start;
  }
  if (ptr - words.data() < 2) { pcc_set_error("pcc_octree_pack: internal overflow"); return PCC_E_NOMEM; }
  *--ptr = (uint32_t)(x >> 32);
  *--ptr = (uint32_t)x;
  const int64_t payload = (int64_t)(end - ptr) * 4;
  if (kHeader + payload > cap) {
    pcc_set_error("pcc_octree_pack: blob needs %lld bytes, capacity %lld", (long long)(kHeader + payload),
                  (long long)cap);
    return PCC_E_NOMEM;
  }
  put_u32(h_out + 20, (uint32_t)payload);
  memcpy(h_out + kHeader, ptr, (size_t)payload);
  *h_len = kHeader + payload;
  return PCC_OK;
}

extern "C" int pcc_octree_peek(const uint8_t* h_in, int64_t len, int64_t* h_n_points, int* h_depth,
                               int32_t* h_origin) {
  if (!h_in || len < kHeader || h_in[0] != 'O' || h_in[1] < 1 || h_in[1] > 3 || h_in[2] > 16 ||
      (h_in[1] == 3 && (h_in[3] < 2 || h_in[3] > kMaxParts))) {
    pcc_set_error("pcc_octree_peek: not an octree blob (len=%lld)", (long long)len);
    return PCC_E_STREAM;
  }
  const int64_t n = (int64_t)get_u32(h_in + 4);
  const int64_t payload = (int64_t)get_u32(h_in + 20);
  if (kHeader + payload > len || (n > 0 && (h_in[2] < 1 || payload < 8)) ||
      (h_in[1] == 3 && payload < (int64_t)h_in[3] * (4 + kHeader))) {
    pcc_set_error("pcc_octree_peek: truncated blob");
    return PCC_E_STREAM;
  }
  if (h_n_points) *h_n_points = n;
  if (h_depth) *h_depth = h_in[2];
  if (h_origin)
    for (int a = 0; a < 3; ++a) h_origin[a] = (int32_t)get_u32(h_in + 8 + 4 * a);
  return PCC_OK;
}

static int octree_unpack_impl(const uint8_t* h_in, int64_t len, int32_t* h_points, int64_t cap_points,
                              int64_t* h_level_n);

extern "C" int pcc_octree_unpack(const uint8_t* h_in, int64_t len, int32_t* h_points, int64_t cap_points) {
  return octree_unpack_impl(h_in, len, h_points, cap_points, nullptr);
}

// the same, and the node count of every level (h_level_n[L], L = 0 .. depth-1; needs 16 entries): level depth-1 is
// the set of the leaves' parents, depth-2 their grandparents — the decoder's stride-16 / stride-32 coordinate sets
extern "C" int pcc_octree_unpack_levels(const uint8_t* h_in, int64_t len, int32_t* h_points, int64_t cap_points,
                                        int64_t* h_level_n) {
  if (!h_level_n) {
    pcc_set_error("pcc_octree_unpack_levels: null h_level_n");
    return PCC_E_ARG;
  }
  for (int L = 0; L < 16; ++L) h_level_n[L] = 0;
  return octree_unpack_impl(h_in, len, h_points, cap_points, h_level_n);
}

// occupancy stream -> Morton cell indexes of the leaves (relative to the root cube), in coding order.  Memory grows
// with what the stream actually holds, never with the announced point count.
static int octree_decode_cells(const uint8_t* h_in, int64_t len, std::vector<uint64_t>* cells, int64_t* h_level_n,
                               int32_t origin[3]) {
  int64_t n = 0;
  int depth = 0;
  const int r = pcc_octree_peek(h_in, len, &n, &depth, origin);
  if (r != PCC_OK) return r;
  cells->clear();
  if (n == 0) return PCC_OK;
  if (h_in[1] == 3) {   // parts, one after the other here (codec.hip decodes them side by side)
    int K = 0;
    const uint8_t* pp[kMaxParts];
    int64_t pl[kMaxParts];
    const int rp = pcc_octree_parts(h_in, len, &K, pp, pl);
    if (rp != PCC_OK) return rp;
    std::vector<PccOctPart> parts((size_t)K);
    for (int k = 0; k < K; ++k) {
      const int rk = pcc_octree_unpack_part(pp[k], pl[k], &parts[(size_t)k]);
      if (rk != PCC_OK) return rk;
    }
    return pcc_octree_merge_parts(h_in, len, parts.data(), K, cells, h_level_n);
  }
  if (h_in[1] != 1) {   // blob version 2 is coded and decoded by the GPU (octree2.hip): there is no host decoder for it
    pcc_set_error("pcc_octree_unpack: blob version %d needs a context (pcc_octree_decode_ctx / pcc_octree_decode_dev)", h_in[1]);
    return PCC_E_STREAM;
  }
  const uint8_t* p = h_in + kHeader;
  const uint8_t* end = p + get_u32(h_in + 20);
  auto word = [&](bool& bad) -> uint32_t {
    if (end - p < 4) { bad = true; return 0; }
    const uint32_t w = get_u32(p);
    p += 4;
    return w;
  };
  bool bad = false;
  uint64_t x = word(bad);
  x |= (uint64_t)word(bad) << 32;
  uint16_t model[kClasses * kCtxPerClass];
  for (auto& m : model) m = kProbOne / 2;
  std::vector<uint64_t> cur(1, 0ull), nxt;
  for (int L = 0; L < depth; ++L) {
    if (h_level_n) h_level_n[L] = (int64_t)cur.size();
    // the next level has at most 8 nodes per node of this one (decoded, so real) and at most n + 7 before the count
    // check below fires: sized by what the stream has shown so far, written through a pointer
    const size_t room = (size_t)std::min<int64_t>(8 * (int64_t)cur.size(), n + 8);
    nxt.resize(room);
    uint64_t* out = nxt.data();
    size_t cnt = 0;
    uint16_t* mc = model + class_of(depth, L) * kCtxPerClass;
    for (size_t i = 0; i < cur.size(); ++i) {
      const uint64_t base = cur[i] << 3;
      int ones = 0;
      for (int j = 0; j < 7; ++j) {
        uint16_t& m = mc[kTri[j] + ones];
        const uint32_t p1 = m;
        const uint32_t cum = (uint32_t)(x & (kProbOne - 1));
        const int bit = cum >= (kProbOne - p1) ? 1 : 0;
        const uint32_t start = bit ? (kProbOne - p1) : 0u;
        const uint32_t freq = bit ? p1 : (kProbOne - p1);
        x = (uint64_t)freq * (x >> kProbBits) + cum - start;
        if (x < kRansL) x = (x << 32) | word(bad);
        adapt(m, bit);
        out[cnt] = base | (uint64_t)j;   // written always, kept when the bit is set
        cnt += (size_t)bit;
        ones += bit;
      }
      int bit7 = 1;   // implied behind seven zeros
      if (ones != 0) {
        uint16_t& m = mc[kTri[7] + ones];
        const uint32_t p1 = m;
        const uint32_t cum = (uint32_t)(x & (kProbOne - 1));
        bit7 = cum >= (kProbOne - p1) ? 1 : 0;
        const uint32_t start = bit7 ? (kProbOne - p1) : 0u;
        const uint32_t freq = bit7 ? p1 : (kProbOne - p1);
        x = (uint64_t)freq * (x >> kProbBits) + cum - start;
        if (x < kRansL) x = (x << 32) | word(bad);
        adapt(m, bit7);
      }
      out[cnt] = base | 7ull;
      cnt += (size_t)bit7;
      if (bad || (int64_t)cnt > n) {
        pcc_set_error("pcc_octree_unpack: corrupt stream at level %d", L);
        return PCC_E_STREAM;
      }
    }
    nxt.resize(cnt);
    cur.swap(nxt);
  }
  if ((int64_t)cur.size() != n) {
    pcc_set_error("pcc_octree_unpack: decoded %zu points, header says %lld", cur.size(), (long long)n);
    return PCC_E_STREAM;
  }
  cells->swap(cur);
  return PCC_OK;
}

// ---- blob version 3: the parts of a blob, one part decoded, the decoded parts put together ---------------------------
// (internal, rans_gate.h: codec.hip runs pcc_octree_unpack_part on its worker threads)
int pcc_octree_parts(const uint8_t* h_in, int64_t len, int* K, const uint8_t** part, int64_t* part_len) {
  int64_t n = 0;
  int depth = 0;
  int32_t origin[3];
  const int r = pcc_octree_peek(h_in, len, &n, &depth, origin);
  if (r != PCC_OK) return r;
  if (h_in[1] != 3) {
    *K = 1;
    part[0] = h_in;
    part_len[0] = len;
    return PCC_OK;
  }
  const int k_parts = h_in[3];
  const int64_t payload = (int64_t)get_u32(h_in + 20);
  int64_t pos = kHeader + 4 * (int64_t)k_parts, sum_n = 0;
  for (int k = 0; k < k_parts; ++k) {
    const int64_t pl = (int64_t)get_u32(h_in + kHeader + 4 * k);
    if (pl < kHeader || pos + pl > kHeader + payload) {
      pcc_set_error("pcc_octree_unpack: part %d of %d: %lld bytes at %lld of a %lld-byte payload", k, k_parts, (long long)pl,
                    (long long)(pos - kHeader), (long long)payload);
      return PCC_E_STREAM;
    }
    const uint8_t* pb = h_in + pos;
    int64_t pn = 0;
    int pd = 0;
    int32_t po[3];
    const int rk = pcc_octree_peek(pb, pl, &pn, &pd, po);
    if (rk != PCC_OK) return rk;
    if (pb[1] != 1 || (pn > 0 && (pd != depth || po[0] != origin[0] || po[1] != origin[1] || po[2] != origin[2]))) {
      pcc_set_error("pcc_octree_unpack: part %d is not a version-1 blob under the frame's root", k);
      return PCC_E_STREAM;
    }
    sum_n += pn;
    part[k] = pb;
    part_len[k] = pl;
    pos += pl;
  }
  if (sum_n != n || pos != kHeader + payload) {
    pcc_set_error("pcc_octree_unpack: parts announce %lld points in %lld bytes, the blob %lld in %lld", (long long)sum_n,
                  (long long)(pos - kHeader), (long long)n, (long long)payload);
    return PCC_E_STREAM;
  }
  *K = k_parts;
  return PCC_OK;
}

int pcc_octree_unpack_part(const uint8_t* h_in, int64_t len, PccOctPart* out) {
  for (int L = 0; L < 16; ++L) out->level_n[L] = 0;
  out->cells.clear();
  if (!h_in || len < kHeader || h_in[1] != 1) {
    pcc_set_error("pcc_octree_unpack: a part must be a version-1 blob");
    return PCC_E_STREAM;
  }
  int32_t origin[3];
  return octree_decode_cells(h_in, len, &out->cells, out->level_n, origin);
}

// parts -> the frame's cells (Morton order) and level counts; a part must begin in a later grandparent cell than the
// one the part in front of it ends in (only then are the leaves distinct, sorted, and the counts of the two lowest
// inner levels sums)
int pcc_octree_merge_parts(const uint8_t* h_in, int64_t len, PccOctPart* parts, int K, std::vector<uint64_t>* cells,
                           int64_t* h_level_n) {
  int64_t n = 0;
  int depth = 0;
  const int r = pcc_octree_peek(h_in, len, &n, &depth, nullptr);
  if (r != PCC_OK) return r;
  int64_t total = 0;
  bool any = false;
  uint64_t last = 0;
  for (int k = 0; k < K; ++k) {
    const std::vector<uint64_t>& c = parts[k].cells;
    if (c.empty()) continue;
    if (any && (c.front() >> 6) <= (last >> 6)) {
      pcc_set_error("pcc_octree_unpack: part %d begins inside or in front of the cell the part before it ends in", k);
      return PCC_E_STREAM;
    }
    any = true;
    last = c.back();
    total += (int64_t)c.size();
  }
  if (total != n) {
    pcc_set_error("pcc_octree_unpack: parts decoded %lld points, header says %lld", (long long)total, (long long)n);
    return PCC_E_STREAM;
  }
  if (h_level_n)
    for (int L = 0; L < 16; ++L) {
      h_level_n[L] = 0;
      for (int k = 0; k < K; ++k)
        if (!parts[k].cells.empty()) h_level_n[L] += parts[k].level_n[L];
    }
  if (K == 1) {
    cells->swap(parts[0].cells);
    return PCC_OK;
  }
  cells->clear();
  cells->reserve((size_t)n);
  for (int k = 0; k < K; ++k) cells->insert(cells->end(), parts[k].cells.begin(), parts[k].cells.end());
  return PCC_OK;
}

// cells -> points (internal: codec.hip, straight into its own arrays)
void pcc_octree_cells_to_points(const uint64_t* cells, int64_t n, const int32_t origin[3], int32_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    out[3 * i] = (int32_t)compact3(cells[i] >> 2) + origin[0];
    out[3 * i + 1] = (int32_t)compact3(cells[i] >> 1) + origin[1];
    out[3 * i + 2] = (int32_t)compact3(cells[i]) + origin[2];
  }
}

// the envelope of version 3 around K finished parts (version-1 blobs under the frame's root)
int pcc_octree_join_parts(int depth, const int32_t origin[3], int64_t n_points, const std::vector<uint8_t>* parts, int K,
                          uint8_t* h_out, int64_t cap, int64_t* h_len) {
  int64_t total = kHeader + 4 * (int64_t)K;
  for (int k = 0; k < K; ++k) total += (int64_t)parts[k].size();
  if (!h_out || !h_len || K < 2 || K > kMaxParts || depth < 1 || depth > 16 || total > cap) {
    pcc_set_error("pcc_octree_join_parts: bad argument (K=%d, %lld bytes, capacity %lld)", K, (long long)total, (long long)cap);
    return total > cap ? PCC_E_NOMEM : PCC_E_ARG;
  }
  memset(h_out, 0, kHeader);
  h_out[0] = 'O';
  h_out[1] = 3;
  h_out[2] = (uint8_t)depth;
  h_out[3] = (uint8_t)K;
  put_u32(h_out + 4, (uint32_t)n_points);
  for (int a = 0; a < 3; ++a) put_u32(h_out + 8 + 4 * a, (uint32_t)origin[a]);
  put_u32(h_out + 20, (uint32_t)(total - kHeader));
  int64_t pos = kHeader + 4 * (int64_t)K;
  for (int k = 0; k < K; ++k) {
    put_u32(h_out + kHeader + 4 * k, (uint32_t)parts[k].size());
    memcpy(h_out + pos, parts[k].data(), parts[k].size());
    pos += (int64_t)parts[k].size();
  }
  *h_len = total;
  return PCC_OK;
}

static inline void cell_point(uint64_t k, const int32_t origin[3], int32_t* out) {
  out[0] = (int32_t)compact3(k >> 2) + origin[0];
  out[1] = (int32_t)compact3(k >> 1) + origin[1];
  out[2] = (int32_t)compact3(k) + origin[2];
}

static int octree_unpack_impl(const uint8_t* h_in, int64_t len, int32_t* h_points, int64_t cap_points,
                              int64_t* h_level_n) {
  int64_t n = 0;
  int depth = 0;
  int32_t origin[3] = {0, 0, 0};
  const int r = pcc_octree_peek(h_in, len, &n, &depth, origin);
  if (r != PCC_OK) return r;
  if (n == 0) return PCC_OK;
  if (!h_points || cap_points < n) {
    pcc_set_error("pcc_octree_unpack: output capacity %lld < %lld points", (long long)cap_points,
                  (long long)n);
    return PCC_E_ARG;
  }
  std::vector<uint64_t> cells;
  const int rc = octree_decode_cells(h_in, len, &cells, h_level_n, origin);
  if (rc != PCC_OK) return rc;
  for (int64_t i = 0; i < n; ++i) cell_point(cells[(size_t)i], origin, h_points + 3 * i);
  return PCC_OK;
}

// internal (rans_gate.h): the same into a vector sized by what was DECODED — for callers holding untrusted blobs,
// which must not size anything from the announced count before the stream has confirmed it
int pcc_octree_unpack_vec(const uint8_t* h_in, int64_t len, std::vector<int32_t>* pts, int64_t* h_level_n /*[16]*/) {
  for (int L = 0; L < 16; ++L) h_level_n[L] = 0;
  std::vector<uint64_t> cells;
  int32_t origin[3] = {0, 0, 0};
  const int rc = octree_decode_cells(h_in, len, &cells, h_level_n, origin);
  if (rc != PCC_OK) return rc;
  pts->resize(cells.size() * 3);
  for (size_t i = 0; i < cells.size(); ++i) cell_point(cells[i], origin, pts->data() + 3 * i);
  return PCC_OK;
}
