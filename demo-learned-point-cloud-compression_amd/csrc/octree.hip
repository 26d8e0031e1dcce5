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

// ---- single-workgroup form (n <= OCT_SMALL_MAX): every level in one launch -----------------
// Latent frames are ~1e3..3e4 leaves; the per-level launches above are pure launch latency for
// them.  One 1024-thread block walks the levels bottom-up; per level it counts the parents
// (block reduce), then scans tile by tile with a running carry and emits parents + occupancy
// bytes.  Levels are packed back to front in `occ` so that the finished array is contiguous
// root-first and ends at occ + cap.
#define OCT_SMALL_MAX 65536
#define OCT_T 1024

__device__ __forceinline__ uint32_t oct_block_scan(uint32_t v, uint32_t* total, uint32_t* s_w) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < OCT_T / 64; ++w) {
    const uint32_t s = s_w[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(OCT_T) void k_oct_small(const uint64_t* __restrict__ keys, int n, int shift,
                                                     uint64_t mask, int depth, uint64_t* __restrict__ buf_a,
                                                     uint64_t* __restrict__ buf_b, uint8_t* __restrict__ occ,
                                                     int cap, uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_w[OCT_T / 64];
  const int tid = threadIdx.x;
  for (int e = tid; e < n; e += OCT_T) buf_a[e] = (keys[e] >> shift) & mask;
  __threadfence_block();
  __syncthreads();
  uint64_t* cur = buf_a;
  uint64_t* nxt = buf_b;
  int n_cur = n, end = cap;
  if (tid == 0) counts[depth] = (uint32_t)n;
  for (int L = depth - 1; L >= 0; --L) {
    // pass A: number of parents
    uint32_t c = 0;
    for (int e = tid; e < n_cur; e += OCT_T) c += (e == 0 || (cur[e - 1] >> 3) != (cur[e] >> 3)) ? 1u : 0u;
    uint32_t m;
    oct_block_scan(c, &m, s_w);
    const int base_out = end - (int)m;
    // pass B: scan tiles with a carry, emit
    uint32_t carry = 0;
    for (int t0 = 0; t0 < n_cur; t0 += OCT_T) {
      const int e = t0 + tid;
      uint32_t f = 0;
      uint64_t pk = 0;
      if (e < n_cur) {
        pk = cur[e] >> 3;
        f = (e == 0 || (cur[e - 1] >> 3) != pk) ? 1u : 0u;
      }
      uint32_t tile_tot;
      const uint32_t ex = oct_block_scan(f, &tile_tot, s_w);
      if (f) {
        uint32_t byte = 0;
        for (int j = e; j < n_cur && j < e + 8; ++j) {
          const uint64_t k = cur[j];
          if ((k >> 3) != pk) break;
          byte |= 1u << (uint32_t)(k & 7ull);
        }
        const uint32_t p = carry + ex;
        nxt[p] = pk;
        occ[base_out + (int)p] = (uint8_t)byte;
      }
      carry += tile_tot;
    }
    if (tid == 0) counts[L] = m;
    __threadfence_block();
    __syncthreads();
    uint64_t* t = cur; cur = nxt; nxt = t;
    n_cur = (int)m;
    end = base_out;
  }
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
    // one launch; levels packed back to front in occ_lv[0 .. depth*n1), one read-back, one copy
    const int cap_s = (int)((size_t)depth * n1);
    hipLaunchKernelGGL(k_oct_small, dim3(1), dim3(OCT_T), 0, st, d_keys, (int)n, key_shift, leaf_mask, depth,
                       buf_a, buf_b, occ_lv, cap_s, counts);
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
    PCC_HIP(hipMemcpyAsync(d_occ, occ_lv + (cap_s - tot), (size_t)tot, hipMemcpyDeviceToDevice, st));
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
