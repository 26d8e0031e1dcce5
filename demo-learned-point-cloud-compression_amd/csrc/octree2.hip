// octree2.hip — blob version 2 of the geometry slot: the octree occupancy ENTROPY coder on the GPU.
//
// Replaces, for point sets above PCC_OCTREE_V2_MIN_LEAVES leaves, the serial host coder of blob version 1
// (octree_host.cpp) behind utils.gpcc_encode / gpcc_decode (shared/utils.py:169-240; tmc3 in the reference): rounds
// 1-3 formed the occupancy bytes on the GPU and coded them in one adaptive binary rANS stream on a host core — two
// 64-bit divisions per decision, 9.4 ms to code and 4.6 ms to decode BASELINE.json configs[2] (a 104k-point LiDAR
// sweep: 197k nodes, 1.5M decisions).  Version 2 keeps that model — context = (level class, bit position, ones so
// far), 12-bit probabilities, adaptation shift 4 — and deals the nodes (breadth-first, root first) in runs of S
// consecutive nodes to the 64 lanes of chunks, one wave per chunk, each lane with its own 32-bit rANS state (L = 2^16,
// 16-bit words) and its own copy of the model in LDS:
//
//   'O' 2 depth 0 | u32 n | i32 origin[3] | u32 payload_len |
//   u32 level_n[depth] | u32 S | u32 n_chunks | u16 p0[108] | u32 words[n_chunks] | chunk payloads (16-bit words)
//   chunk c, lane l : nodes [(64 c + l) S, (64 c + l + 1) S) below n_nodes = sum(level_n)
//   step t = 8 s + j: every lane codes bit j of its node s (nothing when the node does not exist or the bit is
//                     implied: j == 7 behind seven zeros)
//   payload         = 64 x (state lo, state hi) | u16 len[64] | words of lane 0 | words of lane 1 | ..: every lane has
//                     its own run of 16-bit renormalisation words, in the order its decoder consumes them
//
// (The first form of this round shared one word sequence per chunk — blocks per step, a lane's place by ballot + mbcnt,
// as the y / z coder of container version 1 does: 0.92 / 1.04 ms per coder launch for the sweep, i.e. ~0.25 us per
// step: a lone wave issues one instruction every ~5 cycles whether or not it depends on the one before, and a step was
// ~100 instructions — the refill's ballot, two ds_bpermute and window bookkeeping, a 64-bit-float division per
// decision in the encoder.  Per-lane runs need none of that: a refill is a shift and an LDS read of the lane's own next
// word, the division a lookup of 2^32 / freq in a 16-KB LDS table; 128 B of length table per chunk: +0.9 % bytes.
// With steps free of divergent branches and the encoder's records as 16-byte pieces per node: 0.53 / 0.52 ms to code /
// decode the sweep, blob on the host <-> keys / points in HBM; DESIGN.md 6c has the steps.)
//
// A lane's model starts from the frame's average probability per context (p0: a counting pass, 216 B of header)
// instead of 1/2, so that a run of 512 nodes does not pay for learning it again: +3.6 % bytes against version 1 on
// the sweep (1.5 % the 4-byte final states, 0.9 % the run lengths), +4.3 % on a 1M-point room.  The node count of every level
// is in the header because a decoder lane needs the level class of a node before the levels above it are decoded;
// with them ALL nodes decode in one launch, and the leaves follow from ONE exclusive scan of the nodes' child counts
// (breadth-first numbering: the first child of node i is node 1 + sum of the child counts in front of i), a pass that
// writes every child's (parent, octant) link and a pass in which every leaf walks up `depth` links.
// All integer: bit-exact against oracle/pcc_oracle.c (orc_octree2_encode / orc_octree_decode).
#include "common.h"

#include <string.h>

#include <algorithm>
#include <vector>

size_t pcc_octree_wave_scratch(int64_t n);
int pcc_octree_wave_async(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, int depth, uint8_t* d_occ,
                          int64_t cap, uint32_t* d_counts);
void octree_root(uint64_t first, uint64_t last, int key_shift, int* depth, int32_t origin[3]);

namespace {

constexpr int kLanes = 64;
constexpr int kCtx = 108;        // 3 level classes x 36 (bit position, ones so far)
constexpr int kSMax = 512;       // nodes per lane: a launch lasts 8 S dependent steps of one wave
constexpr int kHeader = 24;
constexpr uint32_t kL = 1u << 16;

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

__host__ __device__ inline uint32_t o2_p0(uint64_t c0, uint64_t c1) {
  const uint64_t p = (4096ull * (2 * c1 + 1)) / (2 * (c0 + c1 + 1));
  return (uint32_t)(p < 16 ? 16 : (p > 4080 ? 4080 : p));
}
// (both sides computed and merged by a mask: written as a conditional expression the compiler made it a divergent branch)
__device__ __forceinline__ uint32_t o2_adapt(uint32_t p, uint32_t bit) {
  const uint32_t up = p + ((4096u - p) >> 4), dn = p - (p >> 4), m = 0u - bit;
  return (up & m) | (dn & ~m);
}
__device__ __forceinline__ int lane_rank(unsigned long long bal) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t u) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
}

struct O2Layout {
  int64_t S, nc;
};
static O2Layout o2_layout(int64_t n_nodes) {
  int64_t c = (n_nodes + (int64_t)kLanes * kSMax - 1) / ((int64_t)kLanes * kSMax);
  if (c < 1) c = 1;
  int64_t s = (n_nodes + kLanes * c - 1) / (kLanes * c);
  s = (s + 3) / 4 * 4;
  if (s < 4) s = 4;
  return O2Layout{s, c};
}

// ---- encoder -----------------------------------------------------------------------------------------------------
// zeros and ones seen per context over the whole frame: cnt[2 ctx + bit]
__global__ __launch_bounds__(256) void k_o2_stats(const uint8_t* __restrict__ occ, int64_t n_nodes, int64_t start_last,
                                                  int64_t start_prev, uint32_t* __restrict__ cnt) {
  __shared__ uint32_t s_cnt[2 * kCtx];
  for (int i = threadIdx.x; i < 2 * kCtx; i += blockDim.x) s_cnt[i] = 0u;
  __syncthreads();
  for (int64_t node = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; node < n_nodes; node += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t byte = occ[node];
    const int cls = node >= start_last ? 0 : (node >= start_prev ? 1 : 2);
    int ones = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t bit = (byte >> j) & 1u;
      if (!(j == 7 && ones == 0)) atomicAdd(&s_cnt[2 * (cls * 36 + j * (j + 1) / 2 + ones) + bit], 1u);
      ones += (int)bit;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * kCtx; i += blockDim.x)
    if (s_cnt[i]) atomicAdd(&cnt[i], s_cnt[i]);
}

// floor(2^32 / f) for f = 1 .. 4095 (entry 0 and 1 unused: a frequency is 15 .. 4081): x / f for x < 2^32 is
// mulhi(x, rcp[f]) or one more (checked by the remainder)
struct O2Rcp {
  uint32_t v[4096];
};
__host__ __device__ constexpr O2Rcp o2_rcp_table() {
  O2Rcp t{};
  for (int f = 2; f < 4096; ++f) t.v[f] = (uint32_t)(0x100000000ull / (uint64_t)f);
  t.v[0] = 0;
  t.v[1] = 0xFFFFFFFFu;
  return t;
}
__device__ const O2Rcp kO2Rcp = o2_rcp_table();

// One wave per chunk.  Forward pass: every lane walks its S nodes with its own model and leaves one record per step
// (probability of a one | bit << 15, 0 = nothing coded) in rec[chunk][step][lane]; backward pass: the lane's rANS steps
// in reverse, every renormalisation word stored downwards from the end of the lane's own T-word region of `work`
// ([chunk][lane][T]: a step emits at most one word).  states[chunk][128] and lens[chunk][64] receive the final states
// and the word counts; words_out[chunk] = 192 + sum of the counts.  Every global access of the coding loop is
// unconditional (rans_gpu.hip's rule).
__global__ __launch_bounds__(64) void k_o2_enc(const uint32_t* __restrict__ occ32, int64_t n_nodes, int64_t start_last,
                                               int64_t start_prev, int S, const uint32_t* __restrict__ cnt,
                                               uint16_t* __restrict__ rec, uint16_t* __restrict__ work,
                                               uint16_t* __restrict__ states, uint16_t* __restrict__ lens,
                                               uint32_t* __restrict__ words_out, uint16_t* __restrict__ p0_out) {
  __shared__ uint16_t s_model[(kCtx + 1) * kLanes];   // [ctx][lane]; row kCtx takes the writes of decisions that are not coded
  __shared__ uint16_t s_p0[kCtx];
  __shared__ __attribute__((aligned(16))) uint32_t s_rcp[4096];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  for (int ctx = lane; ctx < kCtx; ctx += kLanes) {
    const uint32_t p = o2_p0(cnt[2 * ctx], cnt[2 * ctx + 1]);
    s_p0[ctx] = (uint16_t)p;
    if (c == 0) p0_out[ctx] = (uint16_t)p;
  }
  {
    const uint4* src = reinterpret_cast<const uint4*>(kO2Rcp.v);
    uint4* dst = reinterpret_cast<uint4*>(s_rcp);
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[lane + 64 * i] = src[lane + 64 * i];
  }
  __syncthreads();
  for (int ctx = 0; ctx < kCtx; ++ctx) s_model[ctx * kLanes + lane] = s_p0[ctx];
  const int64_t T = 8 * (int64_t)S;
  // records: one 16-byte piece per (node, lane) — the node's 8 decisions —, [chunk][node][lane]: the forward pass stores
  // one dwordx4 per node, the backward pass requests a node's piece four nodes (32 steps) ahead.  (As [step][lane]
  // 16-bit entries requested 8 steps ahead the backward pass waited for every one of them: a step is ~0.1 us, a load
  // that comes from L2 ~0.8 us.)
  uint4* rec4 = reinterpret_cast<uint4*>(rec) + c * (int64_t)S * kLanes;
  const int64_t node0 = (c * kLanes + lane) * S;   // a multiple of 4: four nodes per dword
  const int64_t last_dw = (n_nodes - 1) >> 2;
  uint32_t dw_next = occ32[(node0 >> 2) < last_dw ? (node0 >> 2) : last_dw];
  for (int s = 0; s < S; s += 4) {
    const uint32_t dw = dw_next;
    dw_next = occ32[((node0 + s + 4) >> 2) < last_dw ? ((node0 + s + 4) >> 2) : last_dw];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t node = node0 + s + q;
      const bool valid = node < n_nodes;
      const uint32_t byte = (dw >> (8 * q)) & 0xFFu;
      const int cls = node >= start_last ? 0 : (node >= start_prev ? 1 : 2);
      uint32_t p[8];
      int at[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        at[j] = (cls * 36 + j * (j + 1) / 2 + __popc(byte & ((1u << j) - 1u))) * kLanes + lane;
        p[j] = s_model[at[j]];
      }
      uint32_t r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t bit = (byte >> j) & 1u;
        const bool act = valid && !(j == 7 && (byte & 0x7Fu) == 0u);
        r[j] = act ? (p[j] | (bit << 15)) : 0u;
        s_model[act ? at[j] : kCtx * kLanes + lane] = (uint16_t)o2_adapt(p[j], bit);   // no branch: a dummy row for the rest
      }
      rec4[(int64_t)(s + q) * kLanes + lane] = make_uint4(r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16));
    }
  }

  // backward pass
  uint16_t* reg_w = reinterpret_cast<uint16_t*>(uniform_u64((uint64_t)(work + c * kLanes * T)));
  const __amdgpu_buffer_rsrc_t reg_rs = __builtin_amdgcn_make_buffer_rsrc(reg_w, 0, (int)(uint32_t)(kLanes * T * 2), 0x00027000);
  int wp = (int)T;            // words [wp, T) of the lane's region are written
  const uint32_t lane_off = (uint32_t)lane * (uint32_t)T * 2u;
  uint32_t x = kL;
  constexpr int kAhead = 4;   // nodes in flight (S is a multiple of 4)
  uint4 R[kAhead];
  auto fetch = [&](int sn) -> uint4 { return rec4[(int64_t)(sn >= 0 ? sn : 0) * kLanes + lane]; };
#pragma unroll
  for (int d = 0; d < kAhead; ++d) R[d] = fetch(S - 1 - d);
  struct Prep {
    uint32_t freq, start, rcp;
    bool act;
  };
  auto prep = [&](uint32_t r) -> Prep {   // one step ahead: the LDS lookup has a whole step to arrive in
    const uint32_t p1 = r & 0xFFFu, bit = r >> 15;
    Prep q;
    q.act = r != 0u;
    q.freq = bit ? p1 : 4096u - p1;
    q.start = bit ? 4096u - p1 : 0u;
    q.rcp = s_rcp[q.freq & 4095u];
    return q;
  };
  auto rec_of = [](const uint4& v, int j) -> uint32_t {
    const uint32_t w = j < 2 ? v.x : (j < 4 ? v.y : (j < 6 ? v.z : v.w));
    return (w >> (16 * (j & 1))) & 0xFFFFu;
  };
  Prep cur = prep(rec_of(R[0], 7));
  for (int s0 = S - 1; s0 >= 0; s0 -= kAhead) {
#pragma unroll
    for (int d = 0; d < kAhead; ++d) {
      const int sn = s0 - d;
      const uint4 Rc = R[d];
      const uint4 Rn = R[(d + 1) % kAhead];   // node sn - 1 (requested earlier; for d == kAhead - 1: at the top of this round)
#pragma unroll
      for (int j = 7; j >= 0; --j) {
        const Prep nxt = j > 0 ? prep(rec_of(Rc, j - 1)) : prep(sn > 0 ? rec_of(Rn, 7) : 0u);
        const bool need = cur.act && x >= (cur.freq << 20);   // ((L >> 12) << 16) * freq; freq <= 4081
        wp -= need ? 1 : 0;
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)x, reg_rs, need ? lane_off + (uint32_t)wp * 2u : 0xFFFFFFF0u, 0, 0);
        x = need ? x >> 16 : x;
        {
          // x / freq with x < 2^20 freq: mulhi by floor(2^32 / freq) is the quotient or one less.  Selects, no branch: a
          // divergent branch costs a lone wave more than the six instructions it would skip
          const uint32_t q0 = __umulhi(x, cur.rcp);
          const uint32_t r0 = x - q0 * cur.freq;
          const bool over = r0 >= cur.freq;
          const uint32_t qd = q0 + (over ? 1u : 0u), rem = r0 - (over ? cur.freq : 0u);
          x = cur.act ? (qd << 12) + rem + cur.start : x;
        }
        cur = nxt;
      }
      R[d] = fetch(sn - kAhead);   // the slot is free: node sn - kAhead into it
    }
  }
  states[c * 2 * kLanes + 2 * lane] = (uint16_t)x;
  states[c * 2 * kLanes + 2 * lane + 1] = (uint16_t)(x >> 16);
  const uint32_t len = (uint32_t)((int)T - wp);
  lens[c * kLanes + lane] = (uint16_t)len;
  uint32_t tot = len;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) tot += (uint32_t)__shfl_xor((int)tot, d, 64);
  if (lane == 0) words_out[c] = 3u * kLanes + tot;
}

struct O2Head {
  int depth;
  uint32_t n_points;
  int32_t origin[3];
  uint32_t S, nc;
};

// the blob, assembled where `out` points (pinned host memory: the bytes cross PCIe as the kernel writes them):
// workgroup c < nc moves chunk c (states, length table, the 64 runs), workgroup nc writes the header; *len_out = bytes,
// or -1 (cap)
__global__ __launch_bounds__(256) void k_o2_pack(const uint16_t* __restrict__ work, int64_t T,
                                                 const uint16_t* __restrict__ states, const uint16_t* __restrict__ lens,
                                                 const uint32_t* __restrict__ words, const uint32_t* __restrict__ counts,
                                                 const uint16_t* __restrict__ p0, O2Head h, uint8_t* __restrict__ out,
                                                 int64_t cap, long long* __restrict__ len_out) {
  __shared__ unsigned long long s_sum[256];
  __shared__ uint32_t s_off[kLanes + 1];
  const int64_t c = blockIdx.x, nc = h.nc;
  unsigned long long part = 0;
  const int64_t upto = c < nc ? c : nc;
  for (int64_t j = threadIdx.x; j < upto; j += blockDim.x) part += words[j];
  s_sum[threadIdx.x] = part;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) s_sum[threadIdx.x] += s_sum[threadIdx.x + d];
    __syncthreads();
  }
  const unsigned long long before = s_sum[0];
  const unsigned long long head = (unsigned long long)kHeader + 4ull * h.depth + 8ull + 2ull * kCtx + 4ull * nc;
  if (c == nc) {
    const unsigned long long total = head + before * 2;
    const bool fits = (long long)total <= cap;
    if (threadIdx.x == 0) {
      *len_out = fits ? (long long)total : -1;
      __threadfence_system();
    }
    if (!fits) return;
    uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
    if (threadIdx.x == 0) {
      o32[0] = (uint32_t)'O' | (2u << 8) | ((uint32_t)h.depth << 16);
      o32[1] = h.n_points;
      o32[2] = (uint32_t)h.origin[0];
      o32[3] = (uint32_t)h.origin[1];
      o32[4] = (uint32_t)h.origin[2];
      o32[5] = (uint32_t)(total - kHeader);
      o32[6 + h.depth] = h.S;
      o32[7 + h.depth] = h.nc;
    }
    if ((int)threadIdx.x < h.depth) o32[6 + threadIdx.x] = counts[threadIdx.x];
    uint16_t* o16 = reinterpret_cast<uint16_t*>(o32 + 8 + h.depth);
    if ((int)threadIdx.x < kCtx) o16[threadIdx.x] = p0[threadIdx.x];
    uint32_t* tab = o32 + 8 + h.depth + kCtx / 2;
    for (int64_t j = threadIdx.x; j < nc; j += blockDim.x) tab[j] = words[j];
    return;
  }
  const uint32_t cw = words[c];
  if ((long long)(head + (before + cw) * 2) > cap) return;   // the header block reports it
  uint16_t* dst = reinterpret_cast<uint16_t*>(out + head) + before;
  if (threadIdx.x == 0) {   // where every lane's run starts (words behind the states and the length table)
    uint32_t off = 3u * kLanes;
    for (int l = 0; l < kLanes; ++l) {
      s_off[l] = off;
      off += lens[c * kLanes + l];
    }
    s_off[kLanes] = off;
  }
  if (threadIdx.x < 2 * kLanes) dst[threadIdx.x] = states[c * 2 * kLanes + threadIdx.x];
  if (threadIdx.x < kLanes) dst[2 * kLanes + threadIdx.x] = lens[c * kLanes + threadIdx.x];
  __syncthreads();
  const int wave = threadIdx.x >> 6, ln = threadIdx.x & 63;
  for (int l = wave; l < kLanes; l += 4) {
    const uint32_t n_w = s_off[l + 1] - s_off[l];
    const uint16_t* src = work + (c * kLanes + l) * T + (T - n_w);
    uint16_t* d = dst + s_off[l];
    for (uint32_t j = ln; j < n_w; j += 64) d[j] = src[j];
  }
}

// ---- decoder -----------------------------------------------------------------------------------------------------
// status (int32): OR of 1 = a chunk ran out of words or did not use all of its words, 2 = an empty node,
// 8 = the child counts do not add up to the announced level sizes / point count
struct O2Offs {
  int64_t off[18];   // off[L] = nodes in front of level L; off[depth] = n_nodes; off[depth + 1] = n_nodes + n_points
};

// the chunk's words in LDS when they fit (a chunk is 64 S nodes: <= 32 KB of payload for S = 512 unless the stream was
// made to cost more than 8 bits per node), else read from the stream where they lie (slow, correct)
constexpr int kDecLdsWords = 24576;   // 48 KB beside the 13.9 KB of models

template <bool IN_LDS>
__device__ __forceinline__ void o2_decode_chunk(uint16_t* s_model, const uint16_t* __restrict__ s_words,
                                                const uint16_t* __restrict__ p /* the chunk in the stream */, uint32_t cw,
                                                int64_t c, int lane, int64_t n_nodes, int64_t start_last, int64_t start_prev,
                                                int S, uint32_t* __restrict__ occ32, int& bad) {
  uint32_t x = (uint32_t)p[2 * lane] | ((uint32_t)p[2 * lane + 1] << 16);
  // the lane's run: [rbase, rend) in 16-bit words from the chunk's start
  const uint32_t my_len = p[2 * kLanes + lane];
  uint32_t incl = my_len;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
    incl += lane >= d ? o : 0u;
  }
  const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
  if (3u * kLanes + total != cw) {   // wave-uniform
    bad |= 1;
    return;
  }
  const uint32_t rbase = 3u * kLanes + incl - my_len, rend = rbase + my_len;
  uint32_t pos = rbase;   // next word of the run
  auto word_at = [&](uint32_t i) -> uint32_t {
    const uint32_t ic = i < cw ? i : cw - 1;   // a lane at the end of the chunk's last run looks one word too far: never used
    if constexpr (IN_LDS) return s_words[ic];
    return p[ic];
  };
  uint32_t nextw = word_at(pos);
  const int64_t node0 = (c * kLanes + lane) * S;
  // the lane's bytes leave as dwords (node0 and S are multiples of 4) through a descriptor over the node array padded
  // to a whole dword: a lane past the end stores beyond it (dropped)
  const __amdgpu_buffer_rsrc_t occ_rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(uniform_u64((uint64_t)occ32)), 0,
      __builtin_amdgcn_readfirstlane((int)(uint32_t)(((n_nodes + 3) >> 2) << 2)), 0x00027000);
  auto cls_of = [&](int64_t node) -> int { return node >= start_last ? 0 : (node >= start_prev ? 1 : 2); };
  // the model entry of the NEXT decision is requested before the current one is decoded — both candidates (the ones
  // so far, and one more) — so that the LDS round trip is not on the chain from state to state
  uint32_t p_cur = s_model[(cls_of(node0) * 36) * kLanes + lane];
  const int a_dummy = kCtx * kLanes + lane;   // takes the model writes of decisions that are not coded (no branch)
  for (int s = 0; s < S; s += 4) {
    uint32_t dw = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t node = node0 + s + q;
      const bool valid = node < n_nodes;
      const int cbase = cls_of(node) * 36;
      int ones = 0;
      uint32_t byte = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int at = (cbase + j * (j + 1) / 2 + ones) * kLanes + lane;
        uint32_t c0, c1 = 0;
        if (j < 7) {
          const int an = (cbase + (j + 1) * (j + 2) / 2 + ones) * kLanes + lane;
          c0 = s_model[an];
          c1 = s_model[an + kLanes];
        } else {
          c0 = s_model[(cls_of(node + 1) * 36) * kLanes + lane];   // decision 0 of the next node
        }
        const bool act = valid && !(j == 7 && ones == 0);
        const uint32_t p1 = p_cur;
        const uint32_t cum = x & 4095u;
        const uint32_t dbit = cum >= 4096u - p1 ? 1u : 0u;
        const uint32_t start = dbit ? 4096u - p1 : 0u, freq = dbit ? p1 : 4096u - p1;
        const uint32_t xn = freq * (x >> 12) + cum - start;
        x = act ? xn : x;
        s_model[act ? at : a_dummy] = (uint16_t)o2_adapt(p1, dbit);
        const uint32_t bit = act ? dbit : (valid ? 1u : 0u);   // else: the implied bit
        const bool need = act && x < kL;
        bad |= (need && pos >= rend) ? 1 : 0;
        x = need ? (x << 16) | nextw : x;
        pos += need ? 1u : 0u;
        nextw = word_at(pos);   // every step (the same word again when nothing was consumed): no branch
        ones += (int)bit;
        byte |= bit << j;
        p_cur = (j < 7 && bit) ? c1 : c0;
      }
      bad |= (valid && byte == 0u) ? 2 : 0;
      dw |= byte << (8 * q);
    }
    __builtin_amdgcn_raw_buffer_store_b32(dw, occ_rs, (uint32_t)(node0 + s), 0, 0);
  }
  if (pos != rend) bad |= 1;   // every word of the run consumed
}

__global__ __launch_bounds__(64) void k_o2_dec(const uint16_t* __restrict__ p0, const uint32_t* __restrict__ table,
                                               const uint16_t* __restrict__ payload, int64_t n_nodes, int64_t start_last,
                                               int64_t start_prev, int S, uint32_t* __restrict__ occ32,
                                               int32_t* __restrict__ status) {
  __shared__ uint16_t s_model[(kCtx + 1) * kLanes];   // row kCtx: dummy (o2_decode_chunk)
  __shared__ __attribute__((aligned(16))) uint16_t s_words[kDecLdsWords];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  for (int ctx = 0; ctx < kCtx; ++ctx) s_model[ctx * kLanes + lane] = p0[ctx];
  unsigned long long before = 0;
  for (int64_t j = lane; j < c; j += kLanes) before += table[j];
  for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
  const uint32_t cw = (uint32_t)__builtin_amdgcn_readfirstlane((int)table[c]);
  const uint16_t* p = payload + before;
  int bad = 0;
  if (cw < 3 * kLanes) {
    if (lane == 0) atomicOr(status, 1);
    return;
  }
  if (cw <= (uint32_t)kDecLdsWords) {
    // the chunk into LDS: dwords from the aligned address at or below its first word
    const int mis = (int)(((uintptr_t)p >> 1) & 1);
    const uint32_t* p32 = reinterpret_cast<const uint32_t*>(p - mis);
    const uint32_t n_dw = (cw + (uint32_t)mis + 1u) >> 1;
    for (uint32_t i = lane; i < n_dw; i += kLanes) {
      const uint32_t v = p32[i];
      const int w0 = 2 * (int)i - mis;
      if (w0 >= 0 && w0 < (int)cw) s_words[w0] = (uint16_t)v;
      if (w0 + 1 >= 0 && w0 + 1 < (int)cw) s_words[w0 + 1] = (uint16_t)(v >> 16);
    }
    __syncthreads();
    o2_decode_chunk<true>(s_model, s_words, p, cw, c, lane, n_nodes, start_last, start_prev, S, occ32, bad);
  } else {
    o2_decode_chunk<false>(s_model, s_words, p, cw, c, lane, n_nodes, start_last, start_prev, S, occ32, bad);
  }
  const unsigned long long b1 = __ballot((bad & 1) != 0), b2 = __ballot((bad & 2) != 0);
  if (lane == 0 && (b1 | b2) != 0ull) atomicOr(status, (b1 ? 1 : 0) | (b2 ? 2 : 0));
}

__global__ __launch_bounds__(256) void k_o2_popc(const uint8_t* __restrict__ occ, int64_t n_nodes, uint32_t* __restrict__ pc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_nodes) pc[i] = (uint32_t)__popc((uint32_t)occ[i]);
}

// link[child] = parent << 3 | octant for every child the occupancy bytes announce (breadth-first numbering: the first
// child of node i is node 1 + excl[i]); the first node of every level checks that its level starts where the header
// says, node 0 checks the total
__global__ __launch_bounds__(256) void k_o2_link(const uint8_t* __restrict__ occ, const uint32_t* __restrict__ excl,
                                                 const uint32_t* __restrict__ total, int64_t n_nodes, int64_t n_all,
                                                 O2Offs offs, int depth, uint32_t* __restrict__ link,
                                                 int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const uint32_t byte = occ[i];
  const int64_t first = 1 + (int64_t)excl[i];
  int bad = 0;
  if (i == 0 && (int64_t)*total != n_all - 1) bad = 8;
  for (int L = 0; L < depth; ++L)
    if (i == offs.off[L] && first != offs.off[L + 1]) bad = 8;
  if (bad) atomicOr(status, bad);
  int k = 0;
  for (int j = 0; j < 8; ++j)
    if ((byte >> j) & 1u) {
      const int64_t child = first + k;
      if (child < n_all) link[child] = ((uint32_t)i << 3) | (uint32_t)j;
      ++k;
    }
}

// every leaf walks up its `depth` links: the octants on the way are its cell inside the root cube
__global__ __launch_bounds__(256) void k_o2_points(const uint32_t* __restrict__ link, int64_t n_nodes, int64_t n_points,
                                                   int depth, int ox, int oy, int oz, int32_t* __restrict__ points,
                                                   int32_t* __restrict__ status) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_points) return;
  uint64_t code = 0;
  uint32_t idx = (uint32_t)(n_nodes + e);
  for (int d = 0; d < depth; ++d) {
    const uint32_t l = link[idx];
    code |= (uint64_t)(l & 7u) << (3 * d);
    idx = l >> 3;
  }
  if (idx != 0u) atomicOr(status, 8);
  points[3 * e] = (int32_t)pcc_compact3(code >> 2) + ox;
  points[3 * e + 1] = (int32_t)pcc_compact3(code >> 1) + oy;
  points[3 * e + 2] = (int32_t)pcc_compact3(code) + oz;
}

inline uint32_t get_u32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

}  // namespace

// pinned staging of a context, grown on demand (blobs and decoded points cross PCIe through it)
static int o2_stage_reserve(pcc_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->stage_cap) return PCC_OK;
  size_t want = ctx->stage_cap ? ctx->stage_cap : ((size_t)1 << 20);
  while (want < bytes) want *= 2;
  PCC_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->stage) PCC_HIP(hipHostFree(ctx->stage));
  ctx->stage = nullptr;
  ctx->stage_cap = 0;
  PCC_HIP(hipHostMalloc(&ctx->stage, want, hipHostMallocDefault));
  ctx->stage_cap = want;
  return PCC_OK;
}

// ======================================================================== entry points (internal + C-ABI)

// blob version 2 of the rows d_keys[0 .. n) (Morton-sorted, one frame): two synchronisations (the level counts size
// the chunks; the blob's length), the blob itself is written into pinned memory by the packing kernel
int pcc_octree2_encode(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, int depth, const int32_t origin[3],
                       uint8_t* h_out, int64_t cap, int64_t* h_len) {
  PCC_REQUIRE(ctx && d_keys && h_out && h_len && n >= 1 && n < ((int64_t)1 << 27) && depth >= 1 && depth <= 16, PCC_E_ARG,
              "pcc_octree2_encode: bad argument");
  hipStream_t st = ctx->stream;
  const size_t cap_occ = pcc_align((size_t)n) * (size_t)depth;
  PCC_REQUIRE(cap_occ < ((size_t)1 << 31), PCC_E_ARG, "pcc_octree2_encode: %lld leaves at depth %d", (long long)n, depth);
  // The coder's scratch is sized by the node count, which is known after the first synchronisation, and a second
  // reservation would drop what the first holds: the arena is reserved for the node counts of surfaces and sweeps
  // (<= 2.5 nodes per leaf); a sparser set (up to `depth` nodes per leaf) takes a block of its own for this call.
  const int64_t nodes_guess = std::min<int64_t>((int64_t)cap_occ, 5 * n / 2 + 64 * depth + 4096);
  auto coder_bytes = [](int64_t nodes) -> size_t {
    const O2Layout l = o2_layout(nodes);
    const int64_t T = 8 * l.S;
    return (size_t)l.nc * ((size_t)T * kLanes * 2 * 2 + 4 + 3 * kLanes * 2) + 3 * 256 + 4096;   // records, word regions, tables
  };
  PCC_TRY(pcc_arena_reserve(ctx, cap_occ + pcc_octree_wave_scratch(n) + coder_bytes(nodes_guess) + 16384));
  uint8_t* occ = (uint8_t*)pcc_arena_alloc(ctx, cap_occ + 16);
  uint32_t* counts = (uint32_t*)pcc_arena_alloc(ctx, 32 * 4);
  uint32_t* cnt = (uint32_t*)pcc_arena_alloc(ctx, 2 * kCtx * 4);
  uint16_t* p0 = (uint16_t*)pcc_arena_alloc(ctx, kCtx * 2);
  if (!occ || !counts || !cnt || !p0) return PCC_E_NOMEM;
  PccProfScope prof(ctx, "octree2_encode", n, depth, 0, 0);
  if (n <= pcc_octree_small_max())
    PCC_TRY(pcc_octree_small_async(ctx, d_keys, n, key_shift, depth, occ, (int64_t)cap_occ, counts));
  else
    PCC_TRY(pcc_octree_wave_async(ctx, d_keys, n, key_shift, depth, occ, (int64_t)cap_occ, counts));
  uint32_t* hc = (uint32_t*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(hc, counts, (size_t)(depth + 1) * 4, hipMemcpyDeviceToHost, st));
  PCC_HIP(hipMemsetAsync(cnt, 0, 2 * kCtx * 4, st));
  PCC_HIP(hipStreamSynchronize(st));
  int64_t n_nodes = 0;
  for (int L = 0; L < depth; ++L) n_nodes += (int64_t)hc[L];
  PCC_REQUIRE(hc[0] == 1, PCC_E_ARG, "pcc_octree2_encode: keys exceed 3*depth bits (root level has %u nodes)", hc[0]);
  PCC_REQUIRE(n_nodes <= (int64_t)cap_occ, PCC_E_NOMEM, "pcc_octree2_encode: %lld nodes for %lld leaves", (long long)n_nodes, (long long)n);
  const int64_t start_last = n_nodes - (int64_t)hc[depth - 1];
  const int64_t start_prev = depth >= 2 ? start_last - (int64_t)hc[depth - 2] : 0;
  const O2Layout lay = o2_layout(n_nodes);
  const int64_t T = 8 * lay.S, cap_words = 3 * kLanes + kLanes * T;   // bound of a chunk in the blob
  struct Own {   // the rare block of its own, freed on every way out
    void* p = nullptr;
    ~Own() { if (p) (void)hipFree(p); }
  } own;
  const size_t rec_b = pcc_align((size_t)lay.nc * T * kLanes * 2), work_b = rec_b;   // [chunk][step][lane] / [chunk][lane][T]
  const size_t small_b = pcc_align((size_t)lay.nc * (4 + 2 * kLanes * 2 + kLanes * 2));   // words | states | lens
  uint16_t *rec, *work;
  char* small;
  if (pcc_align(ctx->arena_off) + rec_b + work_b + small_b + 1024 <= ctx->arena_cap) {
    rec = (uint16_t*)pcc_arena_alloc(ctx, rec_b);
    work = (uint16_t*)pcc_arena_alloc(ctx, work_b);
    small = (char*)pcc_arena_alloc(ctx, small_b);
  } else {
    PCC_HIP(hipMalloc(&own.p, rec_b + work_b + small_b));
    rec = (uint16_t*)own.p;
    work = (uint16_t*)((char*)own.p + rec_b);
    small = (char*)own.p + rec_b + work_b;
  }
  if (!rec || !work || !small) return PCC_E_NOMEM;
  uint32_t* words = (uint32_t*)small;
  uint16_t* states = (uint16_t*)(small + (size_t)lay.nc * 4);
  uint16_t* lens = states + (size_t)lay.nc * 2 * kLanes;
  const int64_t head = kHeader + 4 * depth + 8 + 2 * kCtx + 4 * lay.nc;
  const int64_t bound = head + 2 * lay.nc * cap_words;
  const int64_t cap_blob = std::min<int64_t>(bound, std::max<int64_t>(cap, head));
  PCC_TRY(o2_stage_reserve(ctx, (size_t)cap_blob + 64));
  uint8_t* stage = (uint8_t*)ctx->stage;
  long long* len_dev = (long long*)ctx->pinned + 64;   // bytes 512 .. of the 4-KB pinned block (device-visible)
  hipLaunchKernelGGL(k_o2_stats, dim3(std::min<unsigned>(nblk(n_nodes, 256), 256u)), dim3(256), 0, st, (const uint8_t*)occ, n_nodes,
                     start_last, start_prev, cnt);
  PCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_o2_enc, dim3((unsigned)lay.nc), dim3(64), 0, st, (const uint32_t*)occ, n_nodes, start_last, start_prev,
                     (int)lay.S, (const uint32_t*)cnt, rec, work, states, lens, words, p0);
  PCC_CHECK_LAUNCH();
  O2Head h;
  h.depth = depth;
  h.n_points = (uint32_t)n;
  for (int a = 0; a < 3; ++a) h.origin[a] = origin[a];
  h.S = (uint32_t)lay.S;
  h.nc = (uint32_t)lay.nc;
  hipLaunchKernelGGL(k_o2_pack, dim3((unsigned)lay.nc + 1), dim3(256), 0, st, (const uint16_t*)work, T, (const uint16_t*)states,
                     (const uint16_t*)lens, (const uint32_t*)words, (const uint32_t*)counts, (const uint16_t*)p0, h, stage,
                     cap_blob, len_dev);
  PCC_CHECK_LAUNCH();
  PCC_HIP(hipStreamSynchronize(st));
  const long long total = *(volatile long long*)len_dev;
  PCC_REQUIRE(total >= 0 && total <= cap, PCC_E_NOMEM, "pcc_octree2_encode: blob does not fit %lld bytes", (long long)cap);
  memcpy(h_out, stage, (size_t)total);
  *h_len = total;
  return PCC_OK;
}

// header of a version-2 blob, checked against its length: everything a decoder sizes from
struct O2Info {
  int depth;
  int64_t n, n_nodes, S, nc, level_n[16];
  int32_t origin[3];
  int64_t off_p0, off_table, off_payload, payload_words;
};
static int o2_parse(const uint8_t* h_in, int64_t len, O2Info* o) {
  PCC_REQUIRE(h_in && len >= kHeader && h_in[0] == 'O' && h_in[1] == 2, PCC_E_STREAM, "octree blob v2: bad header");
  o->depth = h_in[2];
  o->n = (int64_t)get_u32(h_in + 4);
  for (int a = 0; a < 3; ++a) o->origin[a] = (int32_t)get_u32(h_in + 8 + 4 * a);
  const int64_t payload = (int64_t)get_u32(h_in + 20);
  PCC_REQUIRE(kHeader + payload <= len, PCC_E_STREAM, "octree blob v2: truncated");
  if (o->n == 0) {
    o->n_nodes = 0;
    return PCC_OK;
  }
  const int d = o->depth;
  PCC_REQUIRE(d >= 1 && d <= 16 && payload >= 4 * d + 8 + 2 * kCtx + 4, PCC_E_STREAM, "octree blob v2: depth %d, payload %lld", d,
              (long long)payload);
  const uint8_t* q = h_in + kHeader;
  o->n_nodes = 0;
  for (int L = 0; L < d; ++L, q += 4) {
    o->level_n[L] = (int64_t)get_u32(q);
    o->n_nodes += o->level_n[L];
    PCC_REQUIRE(o->level_n[L] >= 1 && (L == 0 ? o->level_n[0] == 1 : o->level_n[L] <= 8 * o->level_n[L - 1]) && o->level_n[L] <= o->n,
                PCC_E_STREAM, "octree blob v2: level %d has %lld nodes", L, (long long)o->level_n[L]);
  }
  PCC_REQUIRE(o->n <= 8 * o->level_n[d - 1] && o->n >= o->level_n[d - 1] && o->n_nodes < ((int64_t)1 << 28), PCC_E_STREAM,
              "octree blob v2: %lld points under %lld nodes", (long long)o->n, (long long)o->level_n[d - 1]);
  o->S = (int64_t)get_u32(q);
  o->nc = (int64_t)get_u32(q + 4);
  q += 8;
  PCC_REQUIRE(o->S >= 4 && o->S % 4 == 0 && o->S <= 4096 && o->nc >= 1 && kLanes * o->S * o->nc >= o->n_nodes &&
                  kLanes * o->S * (o->nc - 1) < o->n_nodes,
              PCC_E_STREAM, "octree blob v2: %lld nodes in %lld chunks of 64 x %lld", (long long)o->n_nodes, (long long)o->nc,
              (long long)o->S);
  o->off_p0 = q - h_in;
  for (int i = 0; i < kCtx; ++i, q += 2) {
    const uint32_t p = (uint32_t)q[0] | ((uint32_t)q[1] << 8);
    PCC_REQUIRE(p >= 16 && p <= 4080, PCC_E_STREAM, "octree blob v2: initial probability %u", p);
  }
  o->off_table = q - h_in;
  PCC_REQUIRE(kHeader + payload - o->off_table >= 4 * o->nc, PCC_E_STREAM, "octree blob v2: truncated chunk table");
  int64_t words = 0;
  for (int64_t c = 0; c < o->nc; ++c) {
    const int64_t cw = (int64_t)get_u32(q + 4 * c);
    PCC_REQUIRE(cw >= 3 * kLanes, PCC_E_STREAM, "octree blob v2: chunk %lld has no states", (long long)c);
    words += cw;
  }
  o->off_payload = o->off_table + 4 * o->nc;
  o->payload_words = words;
  PCC_REQUIRE(o->off_payload + 2 * words == kHeader + payload, PCC_E_STREAM, "octree blob v2: chunks take %lld bytes, blob has %lld",
              (long long)(2 * words), (long long)(kHeader + payload - o->off_payload));
  return PCC_OK;
}

// version-2 blob -> Morton-ordered points int32 [n, 3] (origin added): on the device (d_points) and / or on the host
// (h_points).  h_level_n (16 entries, nullable) receives the node counts of the levels.  One synchronisation.
int pcc_octree2_decode(pcc_ctx* ctx, const uint8_t* h_in, int64_t len, int32_t* d_points, int32_t* h_points, int64_t cap_points,
                       int64_t* h_n_points, int64_t* h_level_n) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_octree2_decode: null ctx");
  O2Info o;
  PCC_TRY(o2_parse(h_in, len, &o));
  if (h_n_points) *h_n_points = o.n;
  if (h_level_n) {
    for (int L = 0; L < 16; ++L) h_level_n[L] = 0;
    if (o.n)
      for (int L = 0; L < o.depth; ++L) h_level_n[L] = o.level_n[L];
  }
  if (o.n == 0 || (!d_points && !h_points)) return PCC_OK;
  PCC_REQUIRE(cap_points >= o.n, PCC_E_NOMEM, "pcc_octree2_decode: %lld points, capacity %lld", (long long)o.n, (long long)cap_points);
  hipStream_t st = ctx->stream;
  const int64_t n_all = o.n_nodes + o.n;
  const int64_t body = len - o.off_p0;   // p0 | table | payload: 4-byte aligned inside the blob (header 24 + 4 depth + 8)
  const size_t occ_bytes = (size_t)(kLanes * o.S * o.nc) + 16;
  PCC_TRY(pcc_arena_reserve(ctx, pcc_align((size_t)body + 16) + pcc_align(occ_bytes) + 2 * pcc_align((size_t)o.n_nodes * 4) +
                                     pcc_align((size_t)n_all * 4) + pcc_align((size_t)o.n * 12) +
                                     pcc_scan_scratch_bytes(o.n_nodes) + 8192));
  uint8_t* d_body = (uint8_t*)pcc_arena_alloc(ctx, (size_t)body + 16);
  uint8_t* occ = (uint8_t*)pcc_arena_alloc(ctx, occ_bytes);
  uint32_t* pc = (uint32_t*)pcc_arena_alloc(ctx, (size_t)o.n_nodes * 4);
  uint32_t* excl = (uint32_t*)pcc_arena_alloc(ctx, (size_t)o.n_nodes * 4);
  uint32_t* link = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n_all * 4);
  int32_t* pts = d_points ? d_points : (int32_t*)pcc_arena_alloc(ctx, (size_t)o.n * 12);
  uint32_t* small = (uint32_t*)pcc_arena_alloc(ctx, 64);   // status | total
  if (!d_body || !occ || !pc || !excl || !link || !pts || !small) return PCC_E_NOMEM;
  PccProfScope prof(ctx, "octree2_decode", o.n, o.depth, o.n_nodes, o.nc);
  const size_t out_bytes = h_points ? (size_t)o.n * 12 : 0;
  PCC_TRY(o2_stage_reserve(ctx, pcc_align((size_t)body) + out_bytes + 64));
  uint8_t* stage = (uint8_t*)ctx->stage;
  memcpy(stage, h_in + o.off_p0, (size_t)body);
  PCC_HIP(hipMemcpyAsync(d_body, stage, (size_t)body, hipMemcpyHostToDevice, st));
  PCC_HIP(hipMemsetAsync(small, 0, 64, st));
  PCC_HIP(hipMemsetAsync(link, 0, (size_t)n_all * 4, st));
  int32_t* status = (int32_t*)small;
  const int64_t start_last = o.n_nodes - o.level_n[o.depth - 1];
  const int64_t start_prev = o.depth >= 2 ? start_last - o.level_n[o.depth - 2] : 0;
  const uint16_t* d_p0 = (const uint16_t*)d_body;
  const uint32_t* d_table = (const uint32_t*)(d_body + (o.off_table - o.off_p0));
  const uint16_t* d_payload = (const uint16_t*)(d_body + (o.off_payload - o.off_p0));
  hipLaunchKernelGGL(k_o2_dec, dim3((unsigned)o.nc), dim3(64), 0, st, d_p0, d_table, d_payload, o.n_nodes, start_last, start_prev,
                     (int)o.S, (uint32_t*)occ, status);
  PCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_o2_popc, dim3(nblk(o.n_nodes, 256)), dim3(256), 0, st, (const uint8_t*)occ, o.n_nodes, pc);
  PCC_CHECK_LAUNCH();
  PCC_TRY(pcc_scan_exclusive_u32(ctx, pc, excl, o.n_nodes, small + 1));
  O2Offs offs;
  int64_t run = 0;
  for (int L = 0; L < o.depth; ++L) {
    offs.off[L] = run;
    run += o.level_n[L];
  }
  offs.off[o.depth] = run;
  offs.off[o.depth + 1] = n_all;
  hipLaunchKernelGGL(k_o2_link, dim3(nblk(o.n_nodes, 256)), dim3(256), 0, st, (const uint8_t*)occ, (const uint32_t*)excl,
                     (const uint32_t*)(small + 1), o.n_nodes, n_all, offs, o.depth, link, status);
  PCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_o2_points, dim3(nblk(o.n, 256)), dim3(256), 0, st, (const uint32_t*)link, o.n_nodes, o.n, o.depth,
                     o.origin[0], o.origin[1], o.origin[2], pts, status);
  PCC_CHECK_LAUNCH();
  uint8_t* stage_out = stage + pcc_align((size_t)body);
  if (h_points) PCC_HIP(hipMemcpyAsync(stage_out, pts, out_bytes, hipMemcpyDeviceToHost, st));
  int32_t* h_status = (int32_t*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(h_status, status, 4, hipMemcpyDeviceToHost, st));
  PCC_HIP(hipStreamSynchronize(st));
  PCC_REQUIRE(*h_status == 0, PCC_E_STREAM, "octree blob v2: corrupt stream (status %d: 1 = words, 2 = empty node, 8 = counts)",
              *h_status);
  if (h_points) memcpy(h_points, stage_out, out_bytes);
  return PCC_OK;
}

// ---- C-ABI: the geometry slot, one call each (utils.gpcc_encode / gpcc_decode, shared/utils.py:169-240) -------------
extern "C" int pcc_octree_encode_version(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, int version,
                                         uint8_t* h_out, int64_t cap, int64_t* h_len) {
  PCC_REQUIRE(ctx && h_out && h_len && n >= 0 && (n == 0 || d_keys) && version >= 0 && version <= 3, PCC_E_ARG,
              "pcc_octree_encode: bad argument");
  if (version == 0) version = n > PCC_OCTREE_V2_MIN_LEAVES ? 2 : (n >= PCC_OCTREE_V3_MIN_LEAVES ? 3 : 1);
  PCC_REQUIRE(version != 3 || (n >= 2 && n <= 8 * (int64_t)pcc_octree_small_max()), PCC_E_ARG,
              "pcc_octree_encode: blob version 3 takes 2 .. %lld leaves (n=%lld)", 8 * (long long)pcc_octree_small_max(), (long long)n);
  const int64_t zero = 0;
  const int32_t org0[3] = {0, 0, 0};
  if (n == 0) {
    PCC_TRY(pcc_octree_pack(nullptr, &zero, 0, 0, org0, h_out, cap, h_len));
    h_out[1] = (uint8_t)version;
    return PCC_OK;
  }
  uint64_t* ends = (uint64_t*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(&ends[0], d_keys, 8, hipMemcpyDeviceToHost, ctx->stream));
  PCC_HIP(hipMemcpyAsync(&ends[1], d_keys + (n - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
  PCC_HIP(hipStreamSynchronize(ctx->stream));
  int depth;
  int32_t origin[3];
  octree_root(ends[0], ends[1], key_shift, &depth, origin);
  if (version == 2) return pcc_octree2_encode(ctx, d_keys, n, key_shift, depth, origin, h_out, cap, h_len);
  if (version == 3) {   // parts under the frame's root: one workgroup each on the GPU, one after the other on this thread
    const int K = pcc_octree_parts_for(n);
    PCC_REQUIRE(n <= (int64_t)K * pcc_octree_small_max() / 2, PCC_E_ARG, "pcc_octree_encode: %lld leaves in %d parts", (long long)n, K);
    constexpr int kStride = 20;
    const int64_t cap_occ = n * depth + 4 * K + 4;
    uint8_t* d_buf = nullptr;
    PCC_HIP(hipMalloc((void**)&d_buf, (size_t)cap_occ + 256 + (size_t)K * kStride * 4));
    uint32_t* d_counts = (uint32_t*)(d_buf + (((size_t)cap_occ + 255) & ~(size_t)255));
    std::vector<uint8_t> occ((size_t)cap_occ);
    std::vector<uint32_t> counts((size_t)K * kStride);
    int rc = pcc_octree_parts_async(ctx, d_keys, n, key_shift, depth, K, d_buf, cap_occ, d_counts, kStride);
    if (rc == PCC_OK && (hipMemcpyAsync(occ.data(), d_buf, (size_t)cap_occ, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                         hipMemcpyAsync(counts.data(), d_counts, counts.size() * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                         hipStreamSynchronize(ctx->stream) != hipSuccess)) {
      pcc_set_error("pcc_octree_encode: reading the parts back failed");
      rc = PCC_E_HIP;
    }
    (void)hipFree(d_buf);
    PCC_TRY(rc);
    std::vector<std::vector<uint8_t>> parts((size_t)K);
    int64_t start = 0;
    for (int k = 0; k < K; ++k) {
      const uint32_t* c = counts.data() + (size_t)k * kStride;
      const int64_t np = (int64_t)c[depth];
      std::vector<int64_t> level_n((size_t)depth, 0);
      int64_t nodes = 0;
      for (int L = 0; L < depth; ++L) { level_n[(size_t)L] = np ? (int64_t)c[L] : 0; nodes += level_n[(size_t)L]; }
      PCC_REQUIRE(start + np <= n && nodes <= np * depth, PCC_E_ARG, "pcc_octree_encode: part %d: %lld leaves behind %lld of %lld, %lld nodes",
                  k, (long long)np, (long long)start, (long long)n, (long long)nodes);
      const size_t off = ((((size_t)start * depth) + 3) & ~(size_t)3) + 4 * (size_t)k;
      parts[(size_t)k].resize((size_t)(64 + 2 * nodes + 16));
      int64_t len = 0;
      PCC_TRY(pcc_octree_pack(np ? occ.data() + off : nullptr, np ? level_n.data() : &zero, np ? depth : 0, np, np ? origin : org0,
                              parts[(size_t)k].data(), (int64_t)parts[(size_t)k].size(), &len));
      parts[(size_t)k].resize((size_t)len);
      start += np;
    }
    PCC_REQUIRE(start == n, PCC_E_ARG, "pcc_octree_encode: the parts hold %lld of %lld leaves", (long long)start, (long long)n);
    return pcc_octree_join_parts(depth, origin, n, parts.data(), K, h_out, cap, h_len);
  }
  uint8_t* d_occ = nullptr;
  PCC_HIP(hipMalloc((void**)&d_occ, (size_t)n * depth));
  std::vector<int64_t> level_n((size_t)depth, 0);
  int rc = pcc_octree_levels(ctx, d_keys, n, key_shift, depth, d_occ, n * depth, level_n.data());
  std::vector<uint8_t> occ;
  if (rc == PCC_OK) {
    int64_t tot = 0;
    for (int64_t v : level_n) tot += v;
    occ.resize((size_t)std::max<int64_t>(tot, 1));
    if (hipMemcpy(occ.data(), d_occ, (size_t)tot, hipMemcpyDeviceToHost) != hipSuccess) rc = PCC_E_HIP;
  }
  (void)hipFree(d_occ);
  PCC_TRY(rc);
  return pcc_octree_pack(occ.data(), level_n.data(), depth, n, origin, h_out, cap, h_len);
}

extern "C" int pcc_octree_encode(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, uint8_t* h_out,
                                 int64_t cap, int64_t* h_len) {
  return pcc_octree_encode_version(ctx, d_keys, n, key_shift, 0, h_out, cap, h_len);
}

extern "C" int pcc_octree_blob_version(const uint8_t* h_in, int64_t len) {
  if (!h_in || len < kHeader || h_in[0] != 'O' || h_in[1] < 1 || h_in[1] > 3) {
    pcc_set_error("not an octree blob (len=%lld)", (long long)len);
    return PCC_E_STREAM;
  }
  return h_in[1];
}

extern "C" int pcc_octree_decode_ctx(pcc_ctx* ctx, const uint8_t* h_in, int64_t len, int32_t* h_points, int64_t cap_points,
                                     int64_t* h_n_points) {
  const int v = pcc_octree_blob_version(h_in, len);
  if (v < 0) return v;
  if (v != 2) return pcc_octree_decode(h_in, len, h_points, cap_points, h_n_points);   // versions 1 and 3: host decoders
  return pcc_octree2_decode(ctx, h_in, len, nullptr, h_points, cap_points, h_n_points, nullptr);
}

extern "C" int pcc_octree_decode_dev(pcc_ctx* ctx, const uint8_t* h_in, int64_t len, int32_t* d_points, int64_t cap_points,
                                     int64_t* h_n_points, int64_t* h_level_n) {
  const int v = pcc_octree_blob_version(h_in, len);
  if (v < 0) return v;
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_octree_decode_dev: null ctx");
  if (v == 2) return pcc_octree2_decode(ctx, h_in, len, d_points, nullptr, cap_points, h_n_points, h_level_n);
  // version 1: the serial host decoder, then one upload
  int64_t n = 0;
  PCC_TRY(pcc_octree_peek(h_in, len, &n, nullptr, nullptr));
  if (h_n_points) *h_n_points = n;
  int64_t level_n[16];
  for (int L = 0; L < 16; ++L) level_n[L] = 0;
  if (n && d_points) {
    PCC_REQUIRE(cap_points >= n, PCC_E_NOMEM, "pcc_octree_decode_dev: %lld points, capacity %lld", (long long)n, (long long)cap_points);
    std::vector<int32_t> pts;
    PCC_TRY(pcc_octree_unpack_vec(h_in, len, &pts, level_n));
    PCC_REQUIRE((int64_t)pts.size() == 3 * n, PCC_E_STREAM, "pcc_octree_decode_dev: decoded %zu points, announced %lld", pts.size() / 3,
                (long long)n);
    PCC_TRY(o2_stage_reserve(ctx, (size_t)n * 12));
    memcpy(ctx->stage, pts.data(), (size_t)n * 12);
    PCC_HIP(hipMemcpyAsync(d_points, ctx->stage, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    PCC_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (h_level_n)
    for (int L = 0; L < 16; ++L) h_level_n[L] = level_n[L];
  return PCC_OK;
}
