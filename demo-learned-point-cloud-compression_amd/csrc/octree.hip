// octree.hip — occupancy bytes of every octree level from Morton-sorted leaves.
//
// Device half of the replacement for utils.gpcc_encode (shared/utils.py:169):
// the reference hands the latent coordinates to the tmc3 subprocess through an
// ASCII PLY file; here the coordinates never leave HBM until they are
// occupancy bytes.  Leaves are Morton-sorted, so the children of a node are
// adjacent: each level is an adjacent-unique pass (flag, prefix scan, emit)
// and the occupancy byte of a node is the OR of (1 << octant) over <= 8
// consecutive entries.  Node counts stay on the device until one read-back at
// the end; the level arrays are then packed root-first into d_occ.
#include "common.h"

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

__global__ void k_oct_flags(const uint64_t* __restrict__ cur, const uint32_t* __restrict__ n_cur_p,
                            int64_t n_max, uint32_t* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_max) return;
  const int64_t n_cur = (int64_t)*n_cur_p;
  uint32_t f = 0;
  if (i < n_cur) f = (i == 0 || (cur[i - 1] >> 3) != (cur[i] >> 3)) ? 1u : 0u;
  flags[i] = f;
}

__global__ void k_oct_emit(const uint64_t* __restrict__ cur, const uint32_t* __restrict__ n_cur_p,
                           const uint32_t* __restrict__ flags, const uint32_t* __restrict__ excl,
                           uint64_t* __restrict__ parents, uint8_t* __restrict__ occ) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_cur = (int64_t)*n_cur_p;
  if (i >= n_cur || !flags[i]) return;
  const uint64_t pk = cur[i] >> 3;
  uint32_t byte = 0;
  for (int64_t j = i; j < n_cur && j < i + 8; ++j) {
    const uint64_t k = cur[j];
    if ((k >> 3) != pk) break;
    byte |= 1u << (uint32_t)(k & 7ull);
  }
  const uint32_t p = excl[i];
  parents[p] = pk;
  occ[p] = (uint8_t)byte;
}

__global__ void k_set_u32(uint32_t* p, uint32_t v) { *p = v; }

__global__ void k_oct_leaves(const uint64_t* __restrict__ keys, int64_t n, int shift, uint64_t mask,
                             uint64_t* __restrict__ leaves) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) leaves[i] = (keys[i] >> shift) & mask;
}

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

__global__ __launch_bounds__(OCT_T) void k_oct_small(const uint64_t* __restrict__ keys, int n, int shift,
                                                     uint64_t mask, int depth, uint32_t* __restrict__ occ32,
                                                     int cap, uint32_t* __restrict__ counts) {
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

extern "C" int pcc_octree_levels(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift,
                                 int depth, uint8_t* d_occ, int64_t cap, int64_t* h_level_n) {
  PCC_REQUIRE(ctx && h_level_n, PCC_E_ARG, "pcc_octree_levels: null arg");
  PCC_REQUIRE(depth >= 1 && depth <= 16, PCC_E_ARG, "pcc_octree_levels: depth=%d", depth);
  PCC_REQUIRE(n >= 1 && n < ((int64_t)1 << 31), PCC_E_ARG, "pcc_octree_levels: n=%lld", (long long)n);
  PCC_REQUIRE(d_keys && d_occ, PCC_E_ARG, "pcc_octree_levels: null buffers");
  PCC_REQUIRE(key_shift >= 0 && key_shift % 3 == 0 && key_shift + 3 * depth <= 48, PCC_E_ARG,
              "pcc_octree_levels: key_shift=%d depth=%d", key_shift, depth);
  hipStream_t st = ctx->stream;
  const size_t n8 = pcc_align((size_t)n * 8), n4 = pcc_align((size_t)n * 4), n1 = pcc_align((size_t)n);
  PCC_TRY(pcc_arena_reserve(ctx, 2 * n8 + 2 * n4 + (size_t)depth * n1 + pcc_scan_scratch_bytes(n) + 4096));
  uint64_t* buf_a = (uint64_t*)pcc_arena_alloc(ctx, (size_t)n * 8);
  uint64_t* buf_b = (uint64_t*)pcc_arena_alloc(ctx, (size_t)n * 8);
  uint32_t* flags = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 4);
  uint32_t* excl = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 4);
  uint8_t* occ_lv = (uint8_t*)pcc_arena_alloc(ctx, (size_t)depth * n1);
  uint32_t* counts = (uint32_t*)pcc_arena_alloc(ctx, (size_t)(depth + 1) * 4);  // counts[L] = nodes at level L
  if (!buf_a || !buf_b || !flags || !excl || !occ_lv || !counts) return PCC_E_NOMEM;
  const size_t mark = ctx->arena_off;
  PccProfScope prof(ctx, "octree_levels", n, depth, 0, 0);
  const uint64_t leaf_mask = (depth == 16 ? ~0ull >> 16 : ((1ull << (3 * depth)) - 1));

  if (n <= OCT_SMALL_MAX) {
    // one launch; levels packed root-first from occ_lv[0], one read-back, one copy
    const int cap_s = (int)((size_t)depth * n1);
    hipLaunchKernelGGL(k_oct_small, dim3(1), dim3(OCT_T), 0, st, d_keys, (int)n, key_shift, leaf_mask, depth,
                       (uint32_t*)occ_lv, cap_s, counts);
    PCC_CHECK_LAUNCH();
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
    PCC_REQUIRE(tot <= cap_s, PCC_E_ARG, "pcc_octree_levels: %lld nodes for %lld leaves", (long long)tot, (long long)n);
    PCC_HIP(hipMemcpyAsync(d_occ, occ_lv, (size_t)tot, hipMemcpyDeviceToDevice, st));
    return PCC_OK;
  }

  hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, counts + depth, (uint32_t)n);
  PCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_oct_leaves, dim3(nblk(n, 256)), dim3(256), 0, st, d_keys, n, key_shift,
                     (depth == 16 ? ~0ull >> 16 : ((1ull << (3 * depth)) - 1)), buf_b);
  PCC_CHECK_LAUNCH();
  const uint64_t* cur = buf_b;
  uint64_t* nxt = buf_a;
  for (int L = depth - 1; L >= 0; --L) {
    // children live at level L+1 (count counts[L+1]); their parents are level L
    hipLaunchKernelGGL(k_oct_flags, dim3(nblk(n, 256)), dim3(256), 0, st, cur,
                       (const uint32_t*)(counts + L + 1), n, flags);
    PCC_CHECK_LAUNCH();
    ctx->arena_off = mark;
    PCC_TRY(pcc_scan_exclusive_u32(ctx, flags, excl, n, counts + L));
    hipLaunchKernelGGL(k_oct_emit, dim3(nblk(n, 256)), dim3(256), 0, st, cur,
                       (const uint32_t*)(counts + L + 1), (const uint32_t*)flags,
                       (const uint32_t*)excl, nxt, occ_lv + (size_t)L * n1);
    PCC_CHECK_LAUNCH();
    cur = nxt;
    nxt = (nxt == buf_a) ? buf_b : buf_a;
  }
  uint32_t* h = (uint32_t*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(h, counts, (size_t)(depth + 1) * 4, hipMemcpyDeviceToHost, st));
  PCC_HIP(hipStreamSynchronize(st));
  int64_t total = 0;
  for (int L = 0; L < depth; ++L) {
    h_level_n[L] = (int64_t)h[L];
    total += h_level_n[L];
  }
  PCC_REQUIRE(h_level_n[0] == 1, PCC_E_ARG,
              "pcc_octree_levels: keys exceed 3*depth bits (root level has %lld nodes)",
              (long long)h_level_n[0]);
  PCC_REQUIRE(total <= cap, PCC_E_ARG, "pcc_octree_levels: d_occ capacity %lld < %lld",
              (long long)cap, (long long)total);
  int64_t off = 0;
  for (int L = 0; L < depth; ++L) {
    PCC_HIP(hipMemcpyAsync(d_occ + off, occ_lv + (size_t)L * n1, (size_t)h_level_n[L],
                           hipMemcpyDeviceToDevice, st));
    off += h_level_n[L];
  }
  return PCC_OK;
}
