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
