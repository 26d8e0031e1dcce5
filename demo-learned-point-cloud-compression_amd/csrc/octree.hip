// octree.hip — occupancy bytes of every octree level from Morton-sorted leaves.
//
// Device half of the replacement for utils.gpcc_encode (shared/utils.py:169):
// the reference hands the latent coordinates to the tmc3 subprocess through an
// ASCII PLY file; here the coordinates never leave HBM until they are
// occupancy bytes.  All levels follow from the sorted leaves directly (see
// k_oct_small): with h = the highest bit in which leaf e differs from leaf
// e-1, e opens a new node at every level L >= lmin(e) = depth - h / 3, so
//   nodes at level L          = #{e : lmin(e) <= L}
//   node of leaf e at level L = #{e' <= e : lmin(e') <= L} - 1
// and the leaf that opens a node at level L+1 sets bit (leaf >> 3 (depth-L-1)) & 7
// in its level-L node.  Latent-sized inputs (<= 65536 leaves): one workgroup,
// one launch.  Larger inputs (round 4; rounds 1-3 walked the levels bottom-up,
// five launches per level): per-wave counts of lmin <= L, ONE scan over the
// [level][wave] table, and one pass that ranks the leaves per level with
// ballots and ORs the octants into the root-first packed array — six launches
// whatever the depth.
#include "common.h"

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

// ---- single-workgroup form (n <= OCT_SMALL_MAX): every level in one launch, from the leaves alone -----------
// Latent frames are ~1e3..3e4 leaves; per-level launches are pure launch latency for them, and walking the
// levels bottom-up inside one workgroup still pays a dependent global round trip per level (117 us for the
// bench latent).  All levels follow from the sorted leaves directly: with h = the highest bit in which leaf e
// differs from leaf e-1, e opens a new node at every level L with 3 (depth - L) <= h, i.e. L >= lmin(e) =
// depth - h / 3 (leaf 0 opens one everywhere).  So
//   nodes at level L          = #{e : lmin(e) <= L}
//   node of leaf e at level L = #{e' <= e : lmin(e') <= L} - 1
// and the leaf that opens a node at level L+1 sets bit (leaf >> 3 (depth-L-1)) & 7 in its level-L node.
// Sweep 1 histograms lmin (level counts -> offsets of the root-first packed array); sweep 2 ranks the leaves
// per level with wave ballots (one per level per 64 leaves) and ORs the bits in.
#define OCT_SMALL_MAX 65536
#define OCT_T 1024
#define OCT_W (OCT_T / 64)
#define OCT_MAXD 16
#define OCT_LDS_WORDS 24576
#define OCT_U 8

__device__ __forceinline__ int oct_lmin(uint64_t prev, uint64_t cur, bool first, int depth) {
  if (first) return 0;
  const uint64_t x = prev ^ cur;
  if (x == 0) return depth + 1;  // repeated leaf: opens nothing, not even a leaf
  return depth - (63 - __builtin_clzll(x)) / 3;
}

__device__ __forceinline__ void oct_small_body(const uint64_t* __restrict__ keys, int n, int shift, uint64_t mask, int depth,
                                               uint32_t* __restrict__ occ32, int cap, uint32_t* __restrict__ counts) {
  // Each wave owns a contiguous chunk of leaves in both sweeps, so after the one exchange of per-wave level
  // counts the waves never wait for each other and their loads overlap.
  __shared__ uint32_t s_off[OCT_MAXD + 1];
  __shared__ uint32_t s_wcnt[OCT_W][OCT_MAXD + 4];  // [wave][v]: leaves of the chunk with lmin == v
  __shared__ uint32_t s_run[OCT_W][OCT_MAXD];       // [wave][L]: level-L nodes opened before the wave's chunk
  // the occupancy bytes are assembled in LDS when they fit (a latent has ~1.3 bytes per leaf): ORing them into
  // HBM costs one L2 atomic transaction per node, ~10 ns each from a single CU (measured: 310 us for 35k nodes)
  __shared__ uint32_t s_occ[OCT_LDS_WORDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t lanes_le = ~0ull >> (63 - lane);
  const int chunk = ((n + OCT_W - 1) / OCT_W + 63) & ~63;
  const int c0 = min(wave * chunk, n), c1 = min(c0 + chunk, n);

  // sweep 1: per-wave histogram of lmin, counted per lane in 16-bit fields (a lane sees <= chunk / 64 <= 64 leaves
  // of a 65536-leaf input; a wave total <= 4096 still fits) and summed across the wave once
  unsigned long long f[5] = {0ull, 0ull, 0ull, 0ull, 0ull};  // field v lives in f[v >> 2], bits 16 (v & 3)
  // OCT_U x 64 leaves in flight per wave (one workgroup has nobody else to hide the latency behind); the previous
  // leaf comes from the neighbouring lane, across steps from lane 63 of the step before
  const uint64_t k_before = c0 < c1 ? (keys[max(c0 - 1, 0)] >> shift) & mask : 0ull;
  uint64_t tail = k_before;
  for (int e0 = c0; e0 < c1; e0 += 64 * OCT_U) {  // wave-uniform
    uint64_t kc[OCT_U];
#pragma unroll
    for (int u = 0; u < OCT_U; ++u) kc[u] = (keys[min(e0 + u * 64 + lane, n - 1)] >> shift) & mask;
#pragma unroll
    for (int u = 0; u < OCT_U; ++u) {
      const int e = e0 + u * 64 + lane;
      const uint64_t up = __shfl_up((unsigned long long)kc[u], 1, 64);
      const uint64_t kp = lane == 0 ? tail : up;
      tail = __shfl((unsigned long long)kc[u], 63, 64);
      if (e < c1) {
        const int lm = oct_lmin(kp, kc[u], e == 0, depth);  // 0 .. depth + 1 <= 17
        const unsigned long long one = 1ull << (16 * (lm & 3));
#pragma unroll
        for (int w = 0; w < 5; ++w) f[w] += (lm >> 2) == w ? one : 0ull;
      }
    }
  }
#pragma unroll
  for (int w = 0; w < 5; ++w) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) f[w] += __shfl_xor(f[w], d, 64);
  }
  if (lane < OCT_MAXD + 2) {
    unsigned long long word = f[0];
#pragma unroll
    for (int w = 1; w < 5; ++w) word = (lane >> 2) == w ? f[w] : word;
    s_wcnt[wave][lane] = (uint32_t)(word >> (16 * (lane & 3))) & 0xFFFFu;
  }
  __syncthreads();
  if (tid < depth) {  // thread L: level-L nodes opened by each wave, exclusive prefix over the waves
    uint32_t run = 0;
    for (int w = 0; w < OCT_W; ++w) {
      uint32_t c = 0;
      for (int v = 0; v <= tid; ++v) c += s_wcnt[w][v];
      s_run[w][tid] = run;
      run += c;
    }
    counts[tid] = run;
    s_off[tid] = run;  // node count for now, offsets below
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t off = 0;
    for (int L = 0; L < depth; ++L) {
      const uint32_t c = s_off[L];
      s_off[L] = off;
      off += c;
    }
    s_off[depth] = off;  // total bytes
    counts[depth] = (uint32_t)n;
  }
  __syncthreads();
  const uint32_t tot = s_off[depth];
  if (tot > (uint32_t)cap) return;  // the host sees the counts and reports the capacity error
  const uint32_t words = (tot + 3u) / 4u;
  const bool in_lds = words <= OCT_LDS_WORDS;  // block-uniform
  if (in_lds) {
    for (uint32_t j = tid; j < words; j += OCT_T) s_occ[j] = 0u;
  } else {
    for (uint32_t j = tid; j < words; j += OCT_T) occ32[j] = 0u;
    __threadfence();
  }
  __syncthreads();

  // sweep 2: rank the chunk's leaves per level with ballots and OR the octants into the parents' bytes.  The
  // running node count of level L lives in lane L's register (read with v_readlane, no LDS round trip per level)
  uint32_t run_reg = lane < depth ? s_run[wave][lane] : 0u;
  tail = k_before;
  for (int e0 = c0; e0 < c1; e0 += 64 * OCT_U) {
    uint64_t kc[OCT_U];
#pragma unroll
    for (int u = 0; u < OCT_U; ++u) kc[u] = (keys[min(e0 + u * 64 + lane, n - 1)] >> shift) & mask;
#pragma unroll
    for (int u = 0; u < OCT_U; ++u) {
      if (e0 + u * 64 >= c1) break;  // wave-uniform
      const int e = e0 + u * 64 + lane;
      const bool valid = e < c1;
      const uint64_t up = __shfl_up((unsigned long long)kc[u], 1, 64);
      const uint64_t kp = lane == 0 ? tail : up;
      tail = __shfl((unsigned long long)kc[u], 63, 64);
      const int lm = valid ? oct_lmin(kp, kc[u], e == 0, depth) : OCT_MAXD + 2;
      // 64 consecutive Morton-sorted leaves share their upper levels: below the wave's smallest lmin - 1 no lane
      // opens a node or a child, so those levels are skipped (typically 3-4 of 13 remain)
      int lo = lm;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) lo = min(lo, __shfl_xor(lo, d, 64));
      for (int L = max(lo - 1, 0); L < depth; ++L) {
        const uint64_t m = __ballot(lm <= L);
        const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)run_reg, L);
        if (valid && lm <= L + 1) {  // opens a node at level L+1: its octant goes into its level-L node
          const uint32_t at = s_off[L] + before + (uint32_t)__popcll(m & lanes_le) - 1u;
          const uint32_t oct = (uint32_t)(kc[u] >> (3 * (depth - L - 1))) & 7u;
          const uint32_t bits = (1u << oct) << (8u * (at & 3u));
          if (in_lds) atomicOr(&s_occ[at >> 2], bits);
          else atomicOr(&occ32[at >> 2], bits);
        }
        if (lane == L) run_reg += (uint32_t)__popcll(m);
      }
    }
  }
  __syncthreads();
  if (in_lds)
    for (uint32_t j = tid; j < words; j += OCT_T) occ32[j] = s_occ[j];
}

__global__ __launch_bounds__(OCT_T) void k_oct_small(const uint64_t* __restrict__ keys, int n, int shift,
                                                     uint64_t mask, int depth, uint32_t* __restrict__ occ32,
                                                     int cap, uint32_t* __restrict__ counts) {
  oct_small_body(keys, n, shift, mask, depth, occ32, cap, counts);
}

// Blob version 3 (octree_host.cpp): the frame's leaves in gridDim.x parts under the frame's root, one workgroup each.
// Part k = leaves [cut(n k / K), cut(n (k + 1) / K)), cut(t) = the first leaf at or behind t whose grandparent cell
// (leaf >> 6) differs from its predecessor's (0 for t = 0, n behind the last such leaf).  Its occupancy bytes go to
// occ + 4-aligned (start * depth) + 4 k — a part has at most (leaves x depth) nodes, so the regions are disjoint and
// the host finds them from the leaf counts alone —, its level counts and leaf count to counts + stride * k (all zero
// for an empty part).
__global__ __launch_bounds__(OCT_T) void k_oct_parts(const uint64_t* __restrict__ keys, int n, int shift, uint64_t mask,
                                                     int depth, uint8_t* __restrict__ occ, uint32_t* __restrict__ counts,
                                                     int counts_stride) {
  __shared__ int s_cut[2];
  const int K = (int)gridDim.x, k = (int)blockIdx.x, lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {   // wave 0 finds both cuts, 64 candidates at a time
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int t = (int)((int64_t)n * (k + which) / K);
      int found = t <= 0 ? 0 : n;
      for (int pos = t; t > 0 && pos < n; pos += 64) {   // wave-uniform
        const int e = pos + lane;
        bool hit = false;
        if (e < n) hit = ((((keys[e] >> shift) & mask) >> 6) != (((keys[e - 1] >> shift) & mask) >> 6));
        const unsigned long long bal = __ballot(hit);
        if (bal) {
          found = pos + (int)__builtin_ctzll(bal);
          break;
        }
      }
      if (lane == 0) s_cut[which] = found;
    }
  }
  __syncthreads();
  const int lo = s_cut[0], hi = s_cut[1];
  uint32_t* my_counts = counts + (size_t)counts_stride * k;
  if (hi <= lo) {
    if (threadIdx.x <= (unsigned)depth) my_counts[threadIdx.x] = 0u;
    return;
  }
  const uint32_t off = (((uint32_t)lo * (uint32_t)depth + 3u) & ~3u) + 4u * (uint32_t)k;
  oct_small_body(keys + lo, hi - lo, shift, mask, depth, reinterpret_cast<uint32_t*>(occ + off), (hi - lo) * depth + 3, my_counts);
}

// Internal (codec.hip): the single-workgroup form alone and nothing read back — the levels packed root-first into
// d_occ (4-byte aligned, cap_s bytes), the node counts of levels 0 .. depth-1 and the leaf count into
// d_counts[depth + 1].  A total above cap_s leaves d_occ untouched; the caller sees it in the counts.
int pcc_octree_small_max() { return OCT_SMALL_MAX; }
int pcc_octree_small_async(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, int depth, uint8_t* d_occ,
                           int64_t cap_s, uint32_t* d_counts) {
  PCC_REQUIRE(ctx && d_keys && d_occ && d_counts && n >= 1 && n <= OCT_SMALL_MAX && depth >= 1 && depth <= 16 &&
                  key_shift >= 0 && key_shift % 3 == 0 && key_shift + 3 * depth <= 48 && cap_s >= 4 &&
                  cap_s < ((int64_t)1 << 31) && (uintptr_t)d_occ % 4 == 0,
              PCC_E_ARG, "pcc_octree_small_async: bad argument (n=%lld depth=%d)", (long long)n, depth);
  PccProfScope prof(ctx, "octree_levels", n, depth, 0, 0);
  const uint64_t leaf_mask = (depth == 16 ? ~0ull >> 16 : ((1ull << (3 * depth)) - 1));
  hipLaunchKernelGGL(k_oct_small, dim3(1), dim3(OCT_T), 0, ctx->stream, d_keys, (int)n, key_shift, leaf_mask, depth,
                     (uint32_t*)d_occ, (int)cap_s, d_counts);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// the same for blob version 3: K parts (2 .. 16), d_occ of (n * depth + 4 K + 4) bytes, d_counts of K * counts_stride
// words (counts_stride >= depth + 1)
int pcc_octree_parts_async(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, int depth, int K, uint8_t* d_occ,
                           int64_t cap_s, uint32_t* d_counts, int counts_stride) {
  PCC_REQUIRE(ctx && d_keys && d_occ && d_counts && n >= 2 && n <= (int64_t)OCT_SMALL_MAX * 8 && depth >= 1 && depth <= 16 &&
                  K >= 2 && K <= 16 && key_shift >= 0 && key_shift % 3 == 0 && key_shift + 3 * depth <= 48 &&
                  cap_s >= n * depth + 4 * K + 4 && cap_s < ((int64_t)1 << 31) && (uintptr_t)d_occ % 4 == 0 &&
                  counts_stride >= depth + 1,
              PCC_E_ARG, "pcc_octree_parts_async: bad argument (n=%lld depth=%d K=%d)", (long long)n, depth, K);
  PccProfScope prof(ctx, "octree_levels", n, depth, K, 0);
  const uint64_t leaf_mask = (depth == 16 ? ~0ull >> 16 : ((1ull << (3 * depth)) - 1));
  hipLaunchKernelGGL(k_oct_parts, dim3((unsigned)K), dim3(OCT_T), 0, ctx->stream, d_keys, (int)n, key_shift, leaf_mask, depth,
                     d_occ, d_counts, counts_stride);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// ---- many-workgroup form (n > OCT_SMALL_MAX): one wave per 64 consecutive leaves -----------------------------
// hist[L][w] = leaves of wave w that open a level-L node (lmin <= L); rows L >= depth stay 0
__global__ __launch_bounds__(256) void k_octw_hist(const uint64_t* __restrict__ keys, int64_t n, int shift, uint64_t mask,
                                                   int depth, int64_t n_waves, uint32_t* __restrict__ hist) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_waves) return;
  const int64_t e = w * 64 + lane, ec = e < n ? e : n - 1;
  const uint64_t kc = (keys[ec] >> shift) & mask, kp = ec > 0 ? (keys[ec - 1] >> shift) & mask : 0ull;
  const int lm = e < n ? oct_lmin(kp, kc, e == 0, depth) : OCT_MAXD + 2;
  uint32_t mine = 0;
  for (int L = 0; L < depth; ++L) {
    const uint32_t c = (uint32_t)__popcll(__ballot(lm <= L));
    mine = lane == L ? c : mine;
  }
  if (lane < OCT_MAXD) hist[(int64_t)lane * n_waves + w] = mine;
}

// from the exclusive scan of hist (flattened [16][n_waves]): counts[L] = nodes of level L (counts[depth] = leaves),
// offs[L] = bytes in front of level L in the root-first packed array (offs[depth] = all nodes)
__global__ void k_octw_offsets(const uint32_t* __restrict__ scan, const uint32_t* __restrict__ total, int64_t n_waves,
                               int depth, uint32_t n, uint32_t* __restrict__ counts, uint32_t* __restrict__ offs) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t off = 0;
  for (int L = 0; L < depth; ++L) {
    const uint32_t lo = scan[(int64_t)L * n_waves], hi = L + 1 < OCT_MAXD ? scan[(int64_t)(L + 1) * n_waves] : *total;
    counts[L] = hi - lo;
    offs[L] = off;
    off += hi - lo;
  }
  counts[depth] = n;
  offs[depth] = off;
}

__global__ __launch_bounds__(256) void k_octw_zero(uint32_t* __restrict__ occ32, const uint32_t* __restrict__ offs, int depth,
                                                   int64_t cap_words) {
  int64_t words = ((int64_t)offs[depth] + 3) / 4;
  words = words < cap_words ? words : cap_words;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < words; j += (int64_t)gridDim.x * blockDim.x) occ32[j] = 0u;
}

// ranks the wave's leaves per level and ORs the octants in.  The bits of lanes that hit the same 32-bit word are
// merged inside the wave first (the children of a node and the nodes of a word are neighbouring lanes: a segmented OR
// over runs of equal word index), so that a word receives one atomic per wave that touches it
__global__ __launch_bounds__(256) void k_octw_emit(const uint64_t* __restrict__ keys, int64_t n, int shift, uint64_t mask,
                                                   int depth, int64_t n_waves, const uint32_t* __restrict__ scan,
                                                   const uint32_t* __restrict__ offs, uint32_t* __restrict__ occ32,
                                                   int64_t cap) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_waves) return;
  if ((int64_t)offs[depth] > cap) return;   // the host sees the counts and reports the capacity error
  const uint64_t lanes_le = ~0ull >> (63 - lane);
  const int64_t e = w * 64 + lane, ec = e < n ? e : n - 1;
  const bool valid = e < n;
  const uint64_t kc = (keys[ec] >> shift) & mask, kp = ec > 0 ? (keys[ec - 1] >> shift) & mask : 0ull;
  const int lm = valid ? oct_lmin(kp, kc, e == 0, depth) : OCT_MAXD + 2;
  // level-L nodes opened before this wave, and the level's offset, in lane L's registers
  uint32_t before_reg = 0, off_reg = 0;
  if (lane < depth) {
    before_reg = scan[(int64_t)lane * n_waves + w] - scan[(int64_t)lane * n_waves];
    off_reg = offs[lane];
  }
  int lo = lm;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) lo = min(lo, __shfl_xor(lo, d, 64));
  for (int L = max(lo - 1, 0); L < depth; ++L) {
    const uint64_t m = __ballot(lm <= L);
    const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)before_reg, L);
    const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)off_reg, L);
    const bool opens = valid && lm <= L + 1;   // opens a node at level L+1: its octant goes into its level-L node
    const uint32_t at = off + before + (uint32_t)__popcll(m & lanes_le) - 1u;
    const uint32_t oct = (uint32_t)(kc >> (3 * (depth - L - 1))) & 7u;
    // every lane knows the level-L node of its leaf (`at`), the lanes that open a child carry its bit: segmented OR
    // towards the first lane of every run of equal word index (`at` ascends with the lane: runs are contiguous)
    const uint32_t word = at >> 2;
    uint32_t bits = opens ? (1u << oct) << (8u * (at & 3u)) : 0u;
#pragma unroll
    for (int d = 1; d <= 32; d <<= 1) {
      const uint32_t ob = (uint32_t)__shfl_down((int)bits, d, 64), ow = (uint32_t)__shfl_down((int)word, d, 64);
      bits |= (lane + d < 64 && ow == word) ? ob : 0u;
    }
    const uint32_t pw = (uint32_t)__shfl_up((int)word, 1, 64);
    if (bits != 0u && (lane == 0 || pw != word)) atomicOr(&occ32[word], bits);
  }
}

// Internal (octree2.hip): the levels of a large input and nothing read back — root-first packed into d_occ (4-byte
// aligned, cap bytes), node counts of levels 0 .. depth-1 and the leaf count into d_counts[depth + 1].  A total above
// cap leaves d_occ untouched.  Scratch from the arena behind what the caller has allocated (the caller reserves
// pcc_octree_wave_scratch(n) bytes for it).
size_t pcc_octree_wave_scratch(int64_t n) {
  const int64_t n_waves = (n + 63) / 64;
  return 2 * pcc_align((size_t)n_waves * OCT_MAXD * 4) + pcc_scan_scratch_bytes(n_waves * OCT_MAXD) + 4096;
}
int pcc_octree_wave_async(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, int depth, uint8_t* d_occ,
                          int64_t cap, uint32_t* d_counts) {
  PCC_REQUIRE(ctx && d_keys && d_occ && d_counts && n >= 1 && n < ((int64_t)1 << 27) && depth >= 1 && depth <= 16 &&
                  key_shift >= 0 && key_shift % 3 == 0 && key_shift + 3 * depth <= 48 && cap >= 4 &&
                  (uintptr_t)d_occ % 4 == 0,
              PCC_E_ARG, "pcc_octree_wave_async: bad argument (n=%lld depth=%d)", (long long)n, depth);
  hipStream_t st = ctx->stream;
  const int64_t n_waves = (n + 63) / 64, cells = n_waves * OCT_MAXD;
  uint32_t* hist = (uint32_t*)pcc_arena_alloc(ctx, (size_t)cells * 4);
  uint32_t* scan = (uint32_t*)pcc_arena_alloc(ctx, (size_t)cells * 4);
  uint32_t* small = (uint32_t*)pcc_arena_alloc(ctx, 256);   // total | offs[17]
  if (!hist || !scan || !small) return PCC_E_NOMEM;
  uint32_t* total = small;
  uint32_t* offs = small + 1;
  const uint64_t leaf_mask = (depth == 16 ? ~0ull >> 16 : ((1ull << (3 * depth)) - 1));
  hipLaunchKernelGGL(k_octw_hist, dim3(nblk(n_waves, 4)), dim3(256), 0, st, d_keys, n, key_shift, leaf_mask, depth, n_waves, hist);
  PCC_CHECK_LAUNCH();
  PCC_TRY(pcc_scan_exclusive_u32(ctx, hist, scan, cells, total));
  hipLaunchKernelGGL(k_octw_offsets, dim3(1), dim3(64), 0, st, (const uint32_t*)scan, (const uint32_t*)total, n_waves, depth,
                     (uint32_t)n, d_counts, offs);
  PCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_octw_zero, dim3(512), dim3(256), 0, st, (uint32_t*)d_occ, (const uint32_t*)offs, depth, cap / 4);
  PCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_octw_emit, dim3(nblk(n_waves, 4)), dim3(256), 0, st, d_keys, n, key_shift, leaf_mask, depth, n_waves,
                     (const uint32_t*)scan, (const uint32_t*)offs, (uint32_t*)d_occ, cap / 4 * 4);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_octree_levels(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift,
                                 int depth, uint8_t* d_occ, int64_t cap, int64_t* h_level_n) {
  PCC_REQUIRE(ctx && h_level_n, PCC_E_ARG, "pcc_octree_levels: null arg");
  PCC_REQUIRE(depth >= 1 && depth <= 16, PCC_E_ARG, "pcc_octree_levels: depth=%d", depth);
  PCC_REQUIRE(n >= 1 && n < ((int64_t)1 << 27), PCC_E_ARG, "pcc_octree_levels: n=%lld", (long long)n);
  PCC_REQUIRE(d_keys && d_occ, PCC_E_ARG, "pcc_octree_levels: null buffers");
  PCC_REQUIRE(key_shift >= 0 && key_shift % 3 == 0 && key_shift + 3 * depth <= 48, PCC_E_ARG,
              "pcc_octree_levels: key_shift=%d depth=%d", key_shift, depth);
  hipStream_t st = ctx->stream;
  const size_t n1 = pcc_align((size_t)n);
  const size_t cap_s = (size_t)depth * n1;   // bytes of the assembled levels (n * depth always suffices)
  PCC_TRY(pcc_arena_reserve(ctx, cap_s + pcc_octree_wave_scratch(n) + 4096));
  uint8_t* occ_lv = (uint8_t*)pcc_arena_alloc(ctx, cap_s);
  uint32_t* counts = (uint32_t*)pcc_arena_alloc(ctx, (size_t)(depth + 1) * 4);  // counts[L] = nodes at level L
  if (!occ_lv || !counts) return PCC_E_NOMEM;
  PccProfScope prof(ctx, "octree_levels", n, depth, 0, 0);
  if (n <= OCT_SMALL_MAX) {
    const uint64_t leaf_mask = (depth == 16 ? ~0ull >> 16 : ((1ull << (3 * depth)) - 1));
    hipLaunchKernelGGL(k_oct_small, dim3(1), dim3(OCT_T), 0, st, d_keys, (int)n, key_shift, leaf_mask, depth,
                       (uint32_t*)occ_lv, (int)cap_s, counts);
    PCC_CHECK_LAUNCH();
  } else {
    PCC_REQUIRE(cap_s < ((size_t)1 << 32), PCC_E_ARG, "pcc_octree_levels: %lld leaves at depth %d", (long long)n, depth);
    PCC_TRY(pcc_octree_wave_async(ctx, d_keys, n, key_shift, depth, occ_lv, (int64_t)cap_s, counts));
  }
  // one read-back (the counts), one copy (the caller's array need not be aligned)
  uint32_t* hc = (uint32_t*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(hc, counts, (size_t)(depth + 1) * 4, hipMemcpyDeviceToHost, st));
  PCC_HIP(hipStreamSynchronize(st));
  int64_t tot = 0;
  for (int L = 0; L < depth; ++L) {
    h_level_n[L] = (int64_t)hc[L];
    tot += h_level_n[L];
  }
  PCC_REQUIRE(h_level_n[0] == 1, PCC_E_ARG,
              "pcc_octree_levels: keys exceed 3*depth bits (root level has %lld nodes)",
              (long long)h_level_n[0]);
  PCC_REQUIRE(tot <= cap, PCC_E_ARG, "pcc_octree_levels: d_occ capacity %lld < %lld", (long long)cap,
              (long long)tot);
  PCC_REQUIRE(tot <= (int64_t)cap_s, PCC_E_ARG, "pcc_octree_levels: %lld nodes for %lld leaves", (long long)tot, (long long)n);
  PCC_HIP(hipMemcpyAsync(d_occ, occ_lv, (size_t)tot, hipMemcpyDeviceToDevice, st));
  return PCC_OK;
}
