// pyramid.hip — coordinate maps of the stride-2 stages.
//
// Because every tensor's rows are Morton-sorted, the parents of a stride-2
// kernel-2 convolution are an adjacent-unique pass over (key >> 3) and the
// <=8 children of a parent are consecutive input rows: the kernel-2 rule book
// is written in the same pass, without a hash table.  The generative
// transposed convolution is the inverse: parent row p spawns rows 8p..8p+7,
// already in Morton order.
//   down: g_a / h_a stride-2 stages, g_s.down_conv (codec_parallel.py:302-303)
//   up  : h_s / g_s generative up stages (codec_pipeline.py:354, codec_parallel.py:376,469)
#include "common.h"

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

__global__ void k_parent_flags(const uint64_t* __restrict__ keys, int64_t n, int pshift,
                               uint32_t* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t pk = keys[i] >> pshift;
  flags[i] = (i == 0 || (keys[i - 1] >> pshift) != pk) ? 1u : 0u;
}

__global__ void k_fill_i32(int32_t* __restrict__ p, int64_t n, int32_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void k_emit_parents(const uint64_t* __restrict__ keys, int64_t n, int cshift,
                               const uint32_t* __restrict__ flags, const uint32_t* __restrict__ excl,
                               uint64_t* __restrict__ pkeys, int32_t* __restrict__ nbr8, int64_t m,
                               int32_t* __restrict__ parent_of /*nullable*/) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  const uint32_t f = flags[i];
  const int64_t p = (int64_t)excl[i] + f - 1;
  if (p >= m) return;  // a caller-supplied m smaller than the real parent count (pcc_down_coords_known): never write past it
  const int o = (int)((k >> cshift) & 7ull);
  if (f) pkeys[p] = (k >> (cshift + 3)) << (cshift + 3);
  nbr8[(int64_t)o * m + p] = (int32_t)i;
  if (parent_of) parent_of[i] = (int32_t)p;
}

__global__ void k_up_keys(const uint64_t* __restrict__ keys, int64_t n, int cshift,
                          uint64_t* __restrict__ ckeys) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 8) return;
  ckeys[t] = keys[t >> 3] | ((uint64_t)(t & 7) << cshift);
}

// the keys of the listed generative children alone (rows[i] = 8p + o): what a pruning keeps of an up stage's 8N
// candidates, without the 8N keys ever being written
__global__ void k_up_keys_rows(const uint64_t* __restrict__ keys, int cshift, const uint32_t* __restrict__ rows, int64_t m,
                               uint64_t* __restrict__ ckeys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t r = rows[i];
  ckeys[i] = keys[r >> 3] | ((uint64_t)(r & 7u) << cshift);
}

// Sizes of the whole pyramid above a sorted key set in one pass: with h = the highest bit in which key e differs
// from key e-1, the two keys have different parents at shift s iff h >= s, so the number of distinct (key >> s) is
// 1 + #{e >= 1 : h_e >= s}.  Histogram of h (64 bins) + a bin for equal neighbours (duplicate rows).
__global__ __launch_bounds__(256) void k_diff_bit_hist(const uint64_t* __restrict__ keys, int64_t n,
                                                       uint32_t* __restrict__ hist /*[65]*/) {
  __shared__ uint32_t s_h[65];
  if (threadIdx.x < 65) s_h[threadIdx.x] = 0u;
  __syncthreads();
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t x = keys[e - 1] ^ keys[e];
    atomicAdd(&s_h[x ? 63 - __builtin_clzll(x) : 64], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 65 && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}

extern "C" int pcc_level_counts(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int child_shift, int levels,
                                int64_t* h_counts, int* h_dup) {
  PCC_REQUIRE(ctx && h_counts && levels >= 1 && levels <= 16, PCC_E_ARG, "pcc_level_counts: bad argument");
  PCC_REQUIRE(child_shift >= 0 && child_shift % 3 == 0 && child_shift + 3 * levels <= 48, PCC_E_ARG,
              "pcc_level_counts: child_shift=%d levels=%d", child_shift, levels);
  for (int l = 0; l < levels; ++l) h_counts[l] = 0;
  if (h_dup) *h_dup = 0;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_keys && n < ((int64_t)1 << 31), PCC_E_ARG, "pcc_level_counts: bad keys");
  hipStream_t st = ctx->stream;
  PCC_TRY(pcc_arena_reserve(ctx, 1024));
  uint32_t* hist = (uint32_t*)pcc_arena_alloc(ctx, 65 * 4);
  if (!hist) return PCC_E_NOMEM;
  PccProfScope prof(ctx, "level_counts", n, child_shift, levels, 0);
  PCC_HIP(hipMemsetAsync(hist, 0, 65 * 4, st));
  unsigned g = nblk(n, 256 * 8);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_diff_bit_hist, dim3(g), dim3(256), 0, st, d_keys, n, hist);
  PCC_CHECK_LAUNCH();
  uint32_t* h = (uint32_t*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(h, hist, 65 * 4, hipMemcpyDeviceToHost, st));
  PCC_HIP(hipStreamSynchronize(st));
  for (int l = 0; l < levels; ++l) {
    const int s = child_shift + 3 * (l + 1);
    int64_t c = 1;
    for (int b = s; b < 64; ++b) c += h[b];
    h_counts[l] = c;
  }
  if (h_dup) *h_dup = h[64] != 0;
  return PCC_OK;
}

static int down_coords_impl(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int child_shift, uint64_t* d_pkeys,
                            int32_t* d_nbr8, int64_t n_cap, int32_t* d_parent_of, int64_t m_known,
                            int64_t* h_n_out);

extern "C" int pcc_down_coords(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int child_shift,
                               uint64_t* d_pkeys, int32_t* d_nbr8, int64_t n_cap,
                               int32_t* d_parent_of, int64_t* h_n_out) {
  return down_coords_impl(ctx, d_keys, n, child_shift, d_pkeys, d_nbr8, n_cap, d_parent_of, -1, h_n_out);
}

extern "C" int pcc_down_coords_known(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int child_shift,
                                     uint64_t* d_pkeys, int32_t* d_nbr8, int64_t n_cap, int32_t* d_parent_of,
                                     int64_t m) {
  PCC_REQUIRE(m >= (n > 0 ? 1 : 0) && m <= n, PCC_E_ARG, "pcc_down_coords_known: m=%lld for n=%lld", (long long)m,
              (long long)n);
  int64_t got = 0;
  return down_coords_impl(ctx, d_keys, n, child_shift, d_pkeys, d_nbr8, n_cap, d_parent_of, m, &got);
}

// ---- the same with the parent count known in advance: two launches, no flag / offset arrays.
// Tile = 2048 keys (256 threads x 8).  Pass 1 counts the parents that START in each tile (a key starts a parent when
// its predecessor has another parent key); pass 2 recomputes those flags, adds up the counts of the tiles in front of
// its own (at most 2048 tiles: 4M keys), scans, and emits parent keys, the kernel-2 rule book and parent_of directly.
#define DC_THREADS 256
#define DC_ITEMS 8
#define DC_TILE (DC_THREADS * DC_ITEMS)

__device__ __forceinline__ uint32_t dc_wave_incl(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint32_t dc_block_excl(uint32_t v, uint32_t* total, uint32_t* lds /*[4]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = dc_wave_incl(v);
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < DC_THREADS / 64; ++w) {
    const uint32_t sv = lds[w];
    if (w < wave) base += sv;
    tot += sv;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// (also presets the kernel-2 rule book, 8 m entries, to -1 = "octant without a child": the emit launch follows)
__global__ __launch_bounds__(DC_THREADS) void k_parent_tile_counts(const uint64_t* __restrict__ keys, int64_t n, int pshift,
                                                                   uint32_t* __restrict__ sums, int32_t* __restrict__ nbr8,
                                                                   int64_t nbr8_n) {
  __shared__ uint32_t lds[4];
  for (int64_t i = (int64_t)blockIdx.x * DC_THREADS + threadIdx.x; i < nbr8_n; i += (int64_t)gridDim.x * DC_THREADS) nbr8[i] = -1;
  const int64_t base = (int64_t)blockIdx.x * DC_TILE + (int64_t)threadIdx.x * DC_ITEMS;
  uint32_t c = 0;
  uint64_t prev = base > 0 && base - 1 < n ? keys[base - 1] >> pshift : 0ull;
#pragma unroll
  for (int j = 0; j < DC_ITEMS; ++j) {
    const int64_t i = base + j;
    if (i < n) {
      const uint64_t pk = keys[i] >> pshift;
      c += (i == 0 || pk != prev) ? 1u : 0u;
      prev = pk;
    }
  }
  uint32_t tot;
  dc_block_excl(c, &tot, lds);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(DC_THREADS) void k_parent_scan_emit(const uint64_t* __restrict__ keys, int64_t n, int cshift,
                                                                 const uint32_t* __restrict__ sums,
                                                                 uint64_t* __restrict__ pkeys, int32_t* __restrict__ nbr8,
                                                                 int64_t m, int32_t* __restrict__ parent_of /*nullable*/) {
  __shared__ uint32_t lds[4];
  const int pshift = cshift + 3;
  const int64_t base = (int64_t)blockIdx.x * DC_TILE + (int64_t)threadIdx.x * DC_ITEMS;
  uint32_t part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += DC_THREADS) part += sums[i];
  uint32_t tile_off;
  dc_block_excl(part, &tile_off, lds);
  uint64_t kv[DC_ITEMS];
  uint32_t f[DC_ITEMS];
  uint32_t c = 0;
  uint64_t prev = base > 0 && base - 1 < n ? keys[base - 1] >> pshift : 0ull;
#pragma unroll
  for (int j = 0; j < DC_ITEMS; ++j) {
    const int64_t i = base + j;
    kv[j] = i < n ? keys[i] : 0ull;
    const uint64_t pk = kv[j] >> pshift;
    f[j] = (i < n && (i == 0 || pk != prev)) ? 1u : 0u;
    prev = pk;
    c += f[j];
  }
  uint32_t tot;
  uint32_t ex = dc_block_excl(c, &tot, lds) + tile_off;   // parents started in front of this thread's first key
#pragma unroll
  for (int j = 0; j < DC_ITEMS; ++j) {
    const int64_t i = base + j;
    ex += f[j];
    if (i >= n) continue;
    const int64_t p = (int64_t)ex - 1;
    if (p >= m) continue;   // a caller-supplied m smaller than the real parent count: never write past it
    const int o = (int)((kv[j] >> cshift) & 7ull);
    if (f[j]) pkeys[p] = (kv[j] >> pshift) << pshift;
    nbr8[(int64_t)o * m + p] = (int32_t)i;
    if (parent_of) parent_of[i] = (int32_t)p;
  }
}

static int down_coords_impl(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int child_shift, uint64_t* d_pkeys,
                            int32_t* d_nbr8, int64_t n_cap, int32_t* d_parent_of, int64_t m_known,
                            int64_t* h_n_out) {
  PCC_REQUIRE(ctx && h_n_out, PCC_E_ARG, "pcc_down_coords: null arg");
  PCC_REQUIRE(child_shift >= 0 && child_shift <= 42 && child_shift % 3 == 0, PCC_E_ARG,
              "pcc_down_coords: child_shift=%d", child_shift);
  *h_n_out = 0;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_keys && d_pkeys && d_nbr8 && n_cap >= n, PCC_E_ARG, "pcc_down_coords: bad buffers");
  PCC_REQUIRE(n < ((int64_t)1 << 31), PCC_E_ARG, "pcc_down_coords: n too large");
  hipStream_t st = ctx->stream;
  const int64_t tiles = (n + DC_TILE - 1) / DC_TILE;
  if (m_known >= 0 && tiles <= DC_TILE) {
    PCC_TRY(pcc_arena_reserve(ctx, pcc_align((size_t)tiles * 4) + 512));
    uint32_t* sums = (uint32_t*)pcc_arena_alloc(ctx, (size_t)tiles * 4);
    if (!sums) return PCC_E_NOMEM;
    PccProfScope prof(ctx, "down_coords", n, child_shift, 0, 0);
    hipLaunchKernelGGL(k_parent_tile_counts, dim3((unsigned)tiles), dim3(DC_THREADS), 0, st, d_keys, n, child_shift + 3, sums,
                       d_nbr8, 8 * m_known);
    PCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_parent_scan_emit, dim3((unsigned)tiles), dim3(DC_THREADS), 0, st, d_keys, n, child_shift,
                       (const uint32_t*)sums, d_pkeys, d_nbr8, m_known, d_parent_of);
    PCC_CHECK_LAUNCH();
    *h_n_out = m_known;
    return PCC_OK;
  }
  PCC_TRY(pcc_arena_reserve(ctx, 2 * pcc_align((size_t)n * 4) + pcc_scan_scratch_bytes(n) + 512));
  uint32_t* flags = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 4);
  uint32_t* excl = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 4);
  uint32_t* total = (uint32_t*)pcc_arena_alloc(ctx, 4);
  if (!flags || !excl || !total) return PCC_E_NOMEM;
  PccProfScope prof(ctx, "down_coords", n, child_shift, 0, 0);
  hipLaunchKernelGGL(k_parent_flags, dim3(nblk(n, 256)), dim3(256), 0, st, d_keys, n,
                     child_shift + 3, flags);
  PCC_CHECK_LAUNCH();
  PCC_TRY(pcc_scan_exclusive_u32(ctx, flags, excl, n, total));
  int64_t m = m_known;
  if (m < 0) {
    uint32_t* h = (uint32_t*)ctx->pinned;
    PCC_HIP(hipMemcpyAsync(h, total, 4, hipMemcpyDeviceToHost, st));
    PCC_HIP(hipStreamSynchronize(st));
    m = (int64_t)h[0];
  }
  hipLaunchKernelGGL(k_fill_i32, dim3(nblk(8 * m, 256)), dim3(256), 0, st, d_nbr8, 8 * m, -1);
  PCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_emit_parents, dim3(nblk(n, 256)), dim3(256), 0, st, d_keys, n, child_shift,
                     (const uint32_t*)flags, (const uint32_t*)excl, d_pkeys, d_nbr8, m, d_parent_of);
  PCC_CHECK_LAUNCH();
  *h_n_out = m;
  return PCC_OK;
}

extern "C" int pcc_up_coords(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int child_shift,
                             uint64_t* d_ckeys) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_up_coords: null ctx");
  PCC_REQUIRE(child_shift >= 0 && child_shift <= 45 && child_shift % 3 == 0, PCC_E_ARG,
              "pcc_up_coords: child_shift=%d", child_shift);
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_keys && d_ckeys, PCC_E_ARG, "pcc_up_coords: null buffers");
  hipLaunchKernelGGL(k_up_keys, dim3(nblk(n * 8, 256)), dim3(256), 0, ctx->stream, d_keys, n,
                     child_shift, d_ckeys);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_up_coords_rows(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int child_shift, const uint32_t* d_rows,
                                  int64_t m, uint64_t* d_ckeys) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_up_coords_rows: null ctx");
  PCC_REQUIRE(child_shift >= 0 && child_shift <= 45 && child_shift % 3 == 0, PCC_E_ARG,
              "pcc_up_coords_rows: child_shift=%d", child_shift);
  PCC_REQUIRE(n >= 0 && m >= 0 && 8 * n <= 0xFFFFFFFFll, PCC_E_ARG, "pcc_up_coords_rows: bad sizes (n=%lld m=%lld)",
              (long long)n, (long long)m);
  if (m == 0) return PCC_OK;
  PCC_REQUIRE(n > 0 && d_keys && d_rows && d_ckeys, PCC_E_ARG, "pcc_up_coords_rows: null buffers");
  hipLaunchKernelGGL(k_up_keys_rows, dim3(nblk(m, 256)), dim3(256), 0, ctx->stream, d_keys, child_shift, d_rows, m, d_ckeys);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}
