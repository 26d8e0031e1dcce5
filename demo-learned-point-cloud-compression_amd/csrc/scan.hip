// scan.hip — device-wide exclusive prefix sum (uint32), reduce-then-scan.
//
// Used by every compaction on the path (parent-coordinate dedup, top-k prune,
// radix-sort digit offsets, octree level build).  Two launches up to 4M elements
// (block sums, then every block adds up the sums before its own and scans its
// tile), three per level of recursion above; the tile is 2048 elements (256 threads x 8, two dwordx4 loads per
// lane), wave64 shuffles for the in-wave part, one LDS exchange per block.
//
// Tried and dropped: one launch per scan with the tiles chained by decoupled
// look-back (ticketed tile order, epoch-tagged status words).  With every tile of
// these sizes resident at once nobody but tile 0 holds an inclusive prefix, so a
// tile sums aggregates 64 at a time all the way back — tile_count / 64 dependent
// cross-XCD round trips: 0.5 ms slower per 1M-point step for all scans, no faster
// than the three short launches even when limited to <= 256 tiles.
#include "common.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix
// and the block total through *total (valid in every thread).
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total,
                                                    uint32_t* lds /*[8]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = wave_incl_scan(v);
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    uint32_t s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(
    const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t lds[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
  if (base + SCAN_ITEMS <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
    uint4 a = p[0], b = p[1];
    s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  } else {
    for (int i = 0; i < SCAN_ITEMS; ++i)
      if (base + i < n) s += in[base + i];
  }
  uint32_t tot;
  block_excl_scan(s, &tot, lds);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// in/out may alias (in-place scan): no __restrict__ on them.
// block_offs: exclusive offsets of the tiles (scanned block sums), or — SUMS_RAW — the raw block sums themselves:
// the block then adds up the sums of the tiles before its own (<= 2048 values) instead of a middle launch doing it.
template <bool SUMS_RAW>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(
    const uint32_t* in, uint32_t* out, int64_t n,
    const uint32_t* __restrict__ block_offs /*nullable*/,
    uint32_t* __restrict__ total /*nullable*/) {
  __shared__ uint32_t lds[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  if (base + SCAN_ITEMS <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
    uint4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = (base + i < n) ? in[base + i] : 0u;
  }
  uint32_t tile_off = 0;
  if constexpr (SUMS_RAW) {
    uint32_t part = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_THREADS) part += block_offs[i];
    block_excl_scan(part, &tile_off, lds);
  } else {
    tile_off = block_offs ? block_offs[blockIdx.x] : 0u;
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) s += v[i];
  uint32_t tot;
  uint32_t ex = block_excl_scan(s, &tot, lds);
  ex += tile_off;
  uint32_t o[SCAN_ITEMS];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { o[i] = ex; ex += v[i]; }
  if (base + SCAN_ITEMS <= n) {
    uint4* q = reinterpret_cast<uint4*>(out + base);
    q[0] = make_uint4(o[0], o[1], o[2], o[3]);
    q[1] = make_uint4(o[4], o[5], o[6], o[7]);
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
      if (base + i < n) out[base + i] = o[i];
  }
  // the thread that owns element n-1 publishes the grand total
  if (total && n > 0 && base <= n - 1 && n - 1 < base + SCAN_ITEMS) *total = ex;
}

size_t pcc_scan_scratch_bytes(int64_t n) {
  size_t bytes = 0;
  int64_t m = n;
  while (m > SCAN_TILE) {
    m = (m + SCAN_TILE - 1) / SCAN_TILE;
    bytes += pcc_align((size_t)m * 4);
  }
  return bytes + 256;
}

// d_in and d_out may alias.  Arrays must be 16-byte aligned.
int pcc_scan_exclusive_u32(pcc_ctx* ctx, const uint32_t* d_in, uint32_t* d_out,
                           int64_t n, uint32_t* d_total) {
  if (n <= 0) {
    if (d_total) PCC_HIP(hipMemsetAsync(d_total, 0, 4, ctx->stream));
    return PCC_OK;
  }
  const int64_t nblk = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nblk == 1) {
    hipLaunchKernelGGL((k_scan_apply<false>), dim3(1), dim3(SCAN_THREADS), 0, ctx->stream,
                       d_in, d_out, n, (const uint32_t*)nullptr, d_total);
    PCC_CHECK_LAUNCH();
    return PCC_OK;
  }
  uint32_t* sums = (uint32_t*)pcc_arena_alloc(ctx, (size_t)nblk * 4);
  if (!sums) return PCC_E_NOMEM;
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nblk), dim3(SCAN_THREADS), 0,
                     ctx->stream, d_in, n, sums);
  PCC_CHECK_LAUNCH();
  if (nblk <= SCAN_TILE) {
    // up to 2048 tiles (4M elements): every block sums the tiles before its own — two launches instead of three
    hipLaunchKernelGGL((k_scan_apply<true>), dim3((unsigned)nblk), dim3(SCAN_THREADS), 0, ctx->stream, d_in, d_out, n,
                       (const uint32_t*)sums, d_total);
    PCC_CHECK_LAUNCH();
    return PCC_OK;
  }
  PCC_TRY(pcc_scan_exclusive_u32(ctx, sums, sums, nblk, nullptr));
  hipLaunchKernelGGL((k_scan_apply<false>), dim3((unsigned)nblk), dim3(SCAN_THREADS), 0,
                     ctx->stream, d_in, d_out, n, (const uint32_t*)sums, d_total);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// Building block exported for the op-level tests (every compaction of the path goes through it).
extern "C" int pcc_exclusive_scan_u32(pcc_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n,
                                      uint32_t* d_total) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_exclusive_scan_u32: null ctx");
  PCC_REQUIRE(n <= 0 || (d_in && d_out), PCC_E_ARG, "pcc_exclusive_scan_u32: null buffer");
  PCC_REQUIRE((uintptr_t)d_in % 16 == 0 && (uintptr_t)d_out % 16 == 0, PCC_E_ARG,
              "pcc_exclusive_scan_u32: buffers must be 16-byte aligned");
  pcc_arena_reset(ctx);
  PCC_TRY(pcc_arena_reserve(ctx, pcc_scan_scratch_bytes(n)));
  PccProfScope prof(ctx, "scan", n, 0, 0, 0);
  return pcc_scan_exclusive_u32(ctx, d_in, d_out, n, d_total);
}
