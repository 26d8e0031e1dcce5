// sort.hip — coordinate keys, stable LSD radix sort, row gathers.
//
// Serves: ME.SparseTensor construction (rows are kept in Morton-key order so
// that stride-2 parents are an adjacent-unique pass and generative children
// are produced already sorted) and utils.sort_tensor / utils.sort_points
// (shared/utils.py:116-165) through the exact linear int64 key.
//
// Radix sort: 8-bit digits, one wave owns a tile of 64*RS_ROUNDS keys; ranks
// inside a tile come from wave64 match-ballots (no LDS atomics in the scatter,
// stable by construction).  Passes whose digit is constant over all keys are
// skipped (a 512x512x256 room only varies in ~27 Morton bits).
#include "common.h"

#define RS_ROUNDS 16
#define RS_WAVE_TILE (64 * RS_ROUNDS)
#define RS_THREADS 256
#define RS_WAVES (RS_THREADS / 64)

// ---------------------------------------------------------------- key kernels
// n_batch: batch indexes must lie below it (65535 = any legal index)
__global__ void k_morton_keys(const int4* __restrict__ coords, int64_t n,
                              uint64_t* __restrict__ keys, int32_t* __restrict__ flag, int n_batch = 65535) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];  // (b,x,y,z)
  bool bad = (c.x < 0) | (c.x > 65534) | (c.x >= n_batch) | (c.y < -32768) | (c.y > 32767) |
             (c.z < -32768) | (c.z > 32767) | (c.w < -32768) | (c.w > 32767);
  if (bad) atomicOr(flag, 1);
  keys[i] = pcc_morton(c.x, c.y, c.z, c.w);
}

__global__ void k_keys_to_coords(const uint64_t* __restrict__ keys, int64_t n,
                                 int4* __restrict__ coords) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b, x, y, z;
  pcc_unmorton(keys[i], &b, &x, &y, &z);
  coords[i] = make_int4(b, x, y, z);
}

__global__ void k_linear_keys(const int4* __restrict__ coords, int64_t n,
                              int64_t* __restrict__ keys) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  // shared/utils.py:131-132: (C * [1e15,1e10,1e5,1]).sum(dim=1) in int64
  keys[i] = (int64_t)c.x * 1000000000000000ll + (int64_t)c.y * 10000000000ll +
            (int64_t)c.z * 100000ll + (int64_t)c.w;
}

// ---------------------------------------------------------------- radix sort
// OR / AND of all keys -> which digits vary
__global__ __launch_bounds__(256) void k_key_bits(const uint64_t* __restrict__ keys,
                                                  int64_t n, unsigned long long* __restrict__ orand) {
  uint64_t o = 0, a = ~0ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t k = keys[i];
    o |= k;
    a &= k;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    o |= __shfl_xor((unsigned long long)o, d, 64);
    a &= __shfl_xor((unsigned long long)a, d, 64);
  }
  // one atomic pair per block (same-address atomics serialise at the memory side)
  __shared__ unsigned long long s_o[4], s_a[4];
  if ((threadIdx.x & 63) == 0) { s_o[threadIdx.x >> 6] = o; s_a[threadIdx.x >> 6] = a; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicOr(&orand[0], s_o[0] | s_o[1] | s_o[2] | s_o[3]);
    atomicAnd(&orand[1], s_a[0] & s_a[1] & s_a[2] & s_a[3]);
  }
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_count(
    const uint64_t* __restrict__ keys, int64_t n, int shift, uint64_t flip,
    uint32_t* __restrict__ counts, int64_t nw) {
  __shared__ uint32_t cnt[RS_WAVES][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * RS_WAVES + wave;
  for (int d = lane; d < 256; d += 64) cnt[wave][d] = 0;
  __builtin_amdgcn_wave_barrier();
  if (wid < nw) {
    const int64_t t0 = wid * RS_WAVE_TILE;
    uint64_t kk[RS_ROUNDS];  // all of the tile's keys in flight before the first LDS atomic
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      const int64_t e = t0 + r * 64 + lane;
      kk[r] = e < n ? keys[e] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      const int64_t e = t0 + r * 64 + lane;
      if (e < n) atomicAdd(&cnt[wave][(uint32_t)(((kk[r] ^ flip) >> shift) & 255u)], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int d = lane; d < 256; d += 64) counts[(int64_t)d * nw + wid] = cnt[wave][d];
  }
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in /*null => iota*/,
    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t n,
    int shift, uint64_t flip, const uint32_t* __restrict__ offsets, int64_t nw) {
  __shared__ uint32_t base_s[RS_WAVES][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * RS_WAVES + wave;
  if (wid >= nw) return;
  volatile uint32_t* base = base_s[wave];
  for (int d = lane; d < 256; d += 64) base[d] = offsets[(int64_t)d * nw + wid];
  __builtin_amdgcn_wave_barrier();
  const int64_t t0 = wid * RS_WAVE_TILE;
  const uint64_t lanes_below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // the tile's keys and values are requested up front: the rounds below are serial through base[] (LDS), and with one
  // load per round each of them paid a full memory latency (16 per wave; the launch has ~1 wave per SIMD)
  uint64_t keys[RS_ROUNDS];
  uint32_t vals[RS_ROUNDS];
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int64_t e = t0 + r * 64 + lane;
    keys[r] = e < n ? keys_in[e] : 0ull;
    vals[r] = (e < n && vals_in) ? vals_in[e] : (uint32_t)e;
  }
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int64_t e = t0 + r * 64 + lane;
    const bool valid = e < n;
    const uint64_t key = keys[r];
    const uint32_t val = vals[r];
    const uint32_t d = (uint32_t)(((key ^ flip) >> shift) & 255u);
    uint64_t mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      mask &= bit ? bal : ~bal;
    }
    const uint32_t rank = (uint32_t)__popcll(mask & lanes_below);
    const uint32_t cnt = (uint32_t)__popcll(mask);
    uint32_t pos = 0;
    if (valid) pos = base[d] + rank;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) base[d] = pos + cnt;
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
  }
}

__global__ void k_iota(uint32_t* __restrict__ p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

static size_t sort_scratch_bytes(int64_t n) {
  const int64_t nw = (n + RS_WAVE_TILE - 1) / RS_WAVE_TILE;
  return pcc_align((size_t)n * 8) + pcc_align((size_t)n * 4) +
         pcc_align((size_t)256 * nw * 4) + pcc_scan_scratch_bytes(256 * nw) + 1024;
}

// ---- single-workgroup sort for the latent-sized tensors (n <= SS_MAX) ----------------------
// The y / z tensors are 1e3..3e4 rows; the multi-kernel path above costs ~30 launches and a
// host read-back for them.  Here one 1024-thread block runs every pass: each of its 16 waves
// owns a contiguous chunk (stable order = wave-major), counts its digits into its LDS row,
// thread d turns the 16x256 counts into exclusive offsets, and each wave scatters its chunk with
// the same match-ballot ranking as k_radix_scatter.  Constant digits are detected in the kernel
// and skipped; the result always ends in (keys, perm), so no host synchronisation is needed.
#define SS_THREADS 1024
#define SS_WAVES (SS_THREADS / 64)
#define SS_MAX 65536

__global__ __launch_bounds__(SS_THREADS) void k_sort_small(uint64_t* __restrict__ keys,
                                                           uint32_t* __restrict__ perm,
                                                           uint64_t* __restrict__ tmp_k,
                                                           uint32_t* __restrict__ tmp_v, int n, uint64_t flip,
                                                           const int* __restrict__ done /*nullable*/,
                                                           const int4* __restrict__ rows_in = nullptr,
                                                           int4* __restrict__ sorted_out = nullptr) {
  // rows_in / sorted_out (both or neither): the rows behind the sorted keys, written out in sorted order at the end
  // (pcc_sort_keys_canonical)
  if (done && *done) return;  // k_compact_coord_keys already wrote the permutation (block-uniform)
  __shared__ uint32_t cnt[SS_WAVES][256];
  __shared__ unsigned long long s_or, s_and;
  __shared__ uint32_t s_wsum[SS_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = ((n + SS_WAVES - 1) / SS_WAVES + 63) & ~63;  // elements per wave, multiple of 64
  const int c0 = wave * chunk, c1 = min(n, c0 + chunk);
  const uint64_t lanes_below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  // which digits vary at all?
  if (tid == 0) { s_or = 0ull; s_and = ~0ull; }
  __syncthreads();
  {
    // one workgroup has nobody to hide a load's latency behind: keep 4 independent loads in flight per lane
    uint64_t o = 0, a = ~0ull;
    for (int e = tid; e < n; e += 4 * SS_THREADS) {
      uint64_t k[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) k[u] = keys[min(e + u * SS_THREADS, n - 1)];  // clamped: duplicates are harmless
#pragma unroll
      for (int u = 0; u < 4; ++u) { o |= k[u]; a &= k[u]; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      o |= __shfl_xor((unsigned long long)o, d, 64);
      a &= __shfl_xor((unsigned long long)a, d, 64);
    }
    if (lane == 0) { atomicOr(&s_or, (unsigned long long)o); atomicAnd(&s_and, (unsigned long long)a); }
  }
  __syncthreads();
  const uint64_t varying = s_or & ~s_and;

  // already sorted (e.g. coordinates that come out of the octree decoder in Morton order):
  // identity permutation, no pass at all
  {
    bool ok = true;
    for (int e = tid + 1; e < n; e += 4 * SS_THREADS) {
      uint64_t lo[4], hi[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = min(e + u * SS_THREADS, n - 1);
        lo[u] = keys[q - 1];
        hi[u] = keys[q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) ok &= (lo[u] ^ flip) <= (hi[u] ^ flip);
    }
    const bool all_ok = __syncthreads_and(ok);
    if (all_ok) {
      for (int e = tid; e < n; e += SS_THREADS) {
        perm[e] = (uint32_t)e;
        if (sorted_out) sorted_out[e] = rows_in[e];
      }
      return;
    }
  }

  uint64_t* kin = keys;
  uint64_t* kout = tmp_k;
  uint32_t* vin = nullptr;  // identity on the first executed pass
  uint32_t* vout = tmp_v;
  for (int p = 0; p < 8; ++p) {
    if (((varying >> (8 * p)) & 0xFFull) == 0) continue;  // block-uniform
    const int shift = 8 * p;
    // (a) per-wave digit counts
    for (int d = lane; d < 256; d += 64) cnt[wave][d] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int e = c0 + lane; e < c1; e += 256) {
      uint64_t k[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) k[u] = kin[min(e + u * 64, n - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e + u * 64 < c1) atomicAdd(&cnt[wave][(uint32_t)(((k[u] ^ flip) >> shift) & 255u)], 1u);
    }
    __syncthreads();
    // (b) counts -> exclusive offsets: digit-major, wave-minor
    uint32_t tot = 0;
    if (tid < 256) {
      for (int w = 0; w < SS_WAVES; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = tot; tot += c; }
    }
    uint32_t inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    if (tid < 256) {
      uint32_t base = inc - tot;
      for (int w = 0; w < wave; ++w) base += s_wsum[w];
      for (int w = 0; w < SS_WAVES; ++w) cnt[w][tid] += base;
    }
    __syncthreads();
    // (c) stable scatter of this wave's chunk
    volatile uint32_t* base = cnt[wave];
    for (int eb = c0; eb < c1; eb += 256) {
      uint64_t pk[4];
      uint32_t pv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = min(eb + u * 64 + lane, n - 1);
        pk[u] = kin[q];
        pv[u] = vin ? vin[q] : (uint32_t)q;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
      const int e0 = eb + u * 64;
      if (e0 >= c1) break;  // wave-uniform
      const int e = e0 + lane;
      const bool valid = e < c1;
      const uint64_t key = valid ? pk[u] : 0ull;
      const uint32_t val = valid ? pv[u] : 0u;
      const uint32_t d = (uint32_t)(((key ^ flip) >> shift) & 255u);
      uint64_t mask = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        mask &= bit ? bal : ~bal;
      }
      const uint32_t rank = (uint32_t)__popcll(mask & lanes_below);
      const uint32_t c = (uint32_t)__popcll(mask);
      uint32_t pos = 0;
      if (valid) pos = base[d] + rank;
      __builtin_amdgcn_wave_barrier();
      if (valid && rank == 0) base[d] = pos + c;
      __builtin_amdgcn_wave_barrier();
      if (valid) { kout[pos] = key; vout[pos] = val; }
      }
    }
    __threadfence_block();
    __syncthreads();
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* nv = (vout == tmp_v) ? perm : tmp_v;
    vin = vout;
    vout = nv;
  }
  // land the result in (keys, perm)
  if (vin == nullptr) {
    for (int e = tid; e < n; e += SS_THREADS) perm[e] = (uint32_t)e;
  } else {
    if (kin != keys) for (int e = tid; e < n; e += SS_THREADS) keys[e] = kin[e];
    if (vin != perm) for (int e = tid; e < n; e += SS_THREADS) perm[e] = vin[e];
  }
  if (sorted_out)
    for (int e = tid; e < n; e += SS_THREADS) sorted_out[e] = rows_in[vin ? vin[e] : (uint32_t)e];
}

// Sort with scratch already reserved in the arena.
// byte_mask (0 = find out): the key bytes to sort on, when the caller knows which can vary — saves the pass over the
// keys that finds the varying bytes and the host round trip behind it; a pass on a byte that happens to be constant
// is a stable no-op, so a superset is always correct.
static int sort_pairs_impl(pcc_ctx* ctx, uint64_t* d_keys, uint32_t* d_perm, int64_t n,
                           int is_signed, const int* d_done = nullptr, const int4* d_rows = nullptr,
                           int4* d_sorted = nullptr, unsigned byte_mask = 0) {
  hipStream_t st = ctx->stream;
  if (n <= 0) return PCC_OK;
  if (n == 1) {
    hipLaunchKernelGGL(k_iota, dim3(1), dim3(64), 0, st, d_perm, n);
    PCC_CHECK_LAUNCH();
    return PCC_OK;
  }
  const int64_t nw = (n + RS_WAVE_TILE - 1) / RS_WAVE_TILE;
  uint64_t* tmp_k = (uint64_t*)pcc_arena_alloc(ctx, (size_t)n * 8);
  uint32_t* tmp_v = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 4);
  if (n <= SS_MAX) {
    if (!tmp_k || !tmp_v) return PCC_E_NOMEM;
    hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(SS_THREADS), 0, st, d_keys, d_perm, tmp_k, tmp_v, (int)n,
                       is_signed ? (1ull << 63) : 0ull, d_done, d_rows, d_sorted);
    PCC_CHECK_LAUNCH();
    return PCC_OK;
  }
  uint32_t* counts = (uint32_t*)pcc_arena_alloc(ctx, (size_t)256 * nw * 4);
  unsigned long long* orand = (unsigned long long*)pcc_arena_alloc(ctx, 16);
  if (!tmp_k || !tmp_v || !counts || !orand) return PCC_E_NOMEM;
  const size_t arena_mark = ctx->arena_off;

  // which digits vary?
  uint64_t varying = 0;
  if (byte_mask) {
    for (int p = 0; p < 8; ++p)
      if ((byte_mask >> p) & 1u) varying |= 0xFFull << (8 * p);
  } else {
    unsigned long long init[2] = {0ull, ~0ull};
    unsigned long long* h = (unsigned long long*)ctx->pinned;
    h[0] = init[0];
    h[1] = init[1];
    PCC_HIP(hipMemcpyAsync(orand, h, 16, hipMemcpyHostToDevice, st));
    unsigned g = nblk(n, 256);
    if (g > 512) g = 512;
    hipLaunchKernelGGL(k_key_bits, dim3(g), dim3(256), 0, st, d_keys, n, orand);
    PCC_CHECK_LAUNCH();
    PCC_HIP(hipMemcpyAsync(h, orand, 16, hipMemcpyDeviceToHost, st));
    PCC_HIP(hipStreamSynchronize(st));
    varying = h[0] & ~h[1];
  }
  const uint64_t flip = is_signed ? (1ull << 63) : 0ull;

  uint64_t* kin = d_keys;
  uint64_t* kout = tmp_k;
  uint32_t* vin = nullptr;  // iota on the first executed pass
  uint32_t* vout = tmp_v;
  // ping-pong targets for values: d_perm and tmp_v
  int executed = 0;
  const unsigned gblk = nblk(nw, RS_WAVES);
  for (int p = 0; p < 8; ++p) {
    if (((varying >> (8 * p)) & 0xFFull) == 0) continue;
    const int shift = 8 * p;
    ctx->arena_off = arena_mark;  // scan scratch is reusable per pass
    hipLaunchKernelGGL(k_radix_count, dim3(gblk), dim3(RS_THREADS), 0, st, kin, n, shift,
                       flip, counts, nw);
    PCC_CHECK_LAUNCH();
    PCC_TRY(pcc_scan_exclusive_u32(ctx, counts, counts, 256 * nw, nullptr));
    vout = (executed % 2 == 0) ? tmp_v : d_perm;
    hipLaunchKernelGGL(k_radix_scatter, dim3(gblk), dim3(RS_THREADS), 0, st, kin,
                       (const uint32_t*)vin, kout, vout, n, shift, flip,
                       (const uint32_t*)counts, nw);
    PCC_CHECK_LAUNCH();
    // swap
    uint64_t* tk = kin; kin = kout; kout = tk;
    vin = vout;
    ++executed;
  }
  if (executed == 0) {
    hipLaunchKernelGGL(k_iota, dim3(nblk(n, 256)), dim3(256), 0, st, d_perm, n);
    PCC_CHECK_LAUNCH();
    return PCC_OK;
  }
  if (kin != d_keys)
    PCC_HIP(hipMemcpyAsync(d_keys, kin, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  if (vin != d_perm)
    PCC_HIP(hipMemcpyAsync(d_perm, vin, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  return PCC_OK;
}

// ---------------------------------------------------------------- misc kernels
__global__ void k_gather_rows16(const uint4* __restrict__ src, const uint32_t* __restrict__ perm,
                                int64_t n, int vec, uint4* __restrict__ dst) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = t / vec;
  int j = (int)(t - row * vec);
  if (row >= n) return;
  dst[row * vec + j] = src[(int64_t)perm[row] * vec + j];
}
__global__ void k_gather_rows4(const uint32_t* __restrict__ src, const uint32_t* __restrict__ perm,
                               int64_t n, int vec, uint32_t* __restrict__ dst) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = t / vec;
  int j = (int)(t - row * vec);
  if (row >= n) return;
  dst[row * vec + j] = src[(int64_t)perm[row] * vec + j];
}
__global__ void k_gather_rows_or_zero(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                      int64_t m, int c, float* __restrict__ dst) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = t / c;
  int j = (int)(t - row * c);
  if (row >= m) return;
  int32_t r = rows[row];
  dst[t] = r >= 0 ? src[(int64_t)r * c + j] : 0.0f;
}

__global__ void k_check_unique(const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 < n && keys[i] == keys[i + 1]) atomicOr(flag, 1);
}

// offsets[b] = lower_bound(keys, b<<48)
__global__ void k_batch_offsets(const uint64_t* __restrict__ keys, int64_t n, int n_batch,
                                int64_t* __restrict__ offs) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_batch) return;
  const uint64_t target = (uint64_t)b << 48;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  offs[b] = lo;
}

// ---------------------------------------------------------------- C-ABI
extern "C" int pcc_morton_keys(pcc_ctx* ctx, const int32_t* d_coords, int64_t n,
                               uint64_t* d_keys, int32_t* d_flag) {
  PCC_REQUIRE(ctx && (n == 0 || (d_coords && d_keys && d_flag)), PCC_E_ARG, "pcc_morton_keys: null arg");
  if (n <= 0) return PCC_OK;
  hipLaunchKernelGGL(k_morton_keys, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream,
                     (const int4*)d_coords, n, d_keys, d_flag, 65535);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// internal (common.h): pcc_morton_keys that also flags batch indexes >= n_batch — the whole-GOP encoder picks the radix
// passes of its input sort from the frame count alone, which is only right when every index lies below it
int pcc_morton_keys_batch(pcc_ctx* ctx, const int32_t* d_coords, int64_t n, int n_batch, uint64_t* d_keys, int32_t* d_flag) {
  PCC_REQUIRE(ctx && (n == 0 || (d_coords && d_keys && d_flag)) && n_batch >= 1, PCC_E_ARG, "pcc_morton_keys: null arg");
  if (n <= 0) return PCC_OK;
  hipLaunchKernelGGL(k_morton_keys, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream,
                     (const int4*)d_coords, n, d_keys, d_flag, n_batch);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_keys_to_coords(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                                  int32_t* d_coords) {
  PCC_REQUIRE(ctx && (n == 0 || (d_coords && d_keys)), PCC_E_ARG, "pcc_keys_to_coords: null arg");
  if (n <= 0) return PCC_OK;
  hipLaunchKernelGGL(k_keys_to_coords, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_keys, n,
                     (int4*)d_coords);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_linear_keys(pcc_ctx* ctx, const int32_t* d_coords, int64_t n,
                               int64_t* d_keys) {
  PCC_REQUIRE(ctx && (n == 0 || (d_coords && d_keys)), PCC_E_ARG, "pcc_linear_keys: null arg");
  if (n <= 0) return PCC_OK;
  hipLaunchKernelGGL(k_linear_keys, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream,
                     (const int4*)d_coords, n, d_keys);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// internal (common.h): pcc_sort_pairs on the key bytes of byte_mask only (bit p = byte p), without looking at the keys
// first — for callers that know which bytes can differ (codec.hip: Morton keys of int16 coordinates vary in bytes 0-5,
// the frame index sits above them)
int pcc_sort_pairs_bytes(pcc_ctx* ctx, uint64_t* d_keys, uint32_t* d_perm, int64_t n, unsigned byte_mask) {
  PCC_REQUIRE(ctx && (n == 0 || (d_keys && d_perm)) && byte_mask != 0 && byte_mask < 256, PCC_E_ARG,
              "pcc_sort_pairs_bytes: bad argument");
  PCC_REQUIRE(n < ((int64_t)1 << 31), PCC_E_ARG, "pcc_sort_pairs_bytes: n too large");
  PCC_TRY(pcc_arena_reserve(ctx, sort_scratch_bytes(n)));
  PccProfScope prof(ctx, "sort_pairs", n, 0, 0, 0);
  return sort_pairs_impl(ctx, d_keys, d_perm, n, 0, nullptr, nullptr, nullptr, byte_mask);
}

extern "C" int pcc_sort_pairs(pcc_ctx* ctx, uint64_t* d_keys, uint32_t* d_perm, int64_t n,
                              int is_signed) {
  PCC_REQUIRE(ctx && (n == 0 || (d_keys && d_perm)), PCC_E_ARG, "pcc_sort_pairs: null arg");
  PCC_REQUIRE(n < ((int64_t)1 << 31), PCC_E_ARG, "pcc_sort_pairs: n too large");
  PCC_TRY(pcc_arena_reserve(ctx, sort_scratch_bytes(n)));
  PccProfScope prof(ctx, "sort_pairs", n, is_signed, 0, 0);
  return sort_pairs_impl(ctx, d_keys, d_perm, n, is_signed);
}

// Canonical order of latent-sized tensors.  For |x|,|y|,|z| < 50000 the reference key
// b*1e15 + x*1e10 + y*1e5 + z orders rows exactly like the tuple (b,x,y,z) (each decimal "digit" stays below
// half its base), so the order may be taken on a compact key instead: per field, subtract the minimum, drop the
// trailing zero bits every row shares (coordinates are multiples of the tensor stride) and concatenate only the
// bits that remain — ~18 bits for a 26k-voxel latent instead of the ~50 varying bits of the decimal key.
//
// Rows of a coordinate set are distinct, so with a key domain of <= 2^20 values no sort is needed at all: the
// position of a row is the number of occupied keys below its own.  The kernel sets one bit per row in an LDS
// bitmap (<= 128 KB), scans the word popcounts and writes perm[rank(key)] = row — five sweeps over the rows in
// one workgroup instead of three radix passes (137 -> ~30 us for the bench latent).  A repeated key (atomicOr
// finds its bit set) or a wider domain falls back to k_sort_small on the compact key (3 passes instead of 7);
// out-of-range input to the decimal key itself (sign-flipped for unsigned order).  `done` tells k_sort_small,
// which is launched behind this kernel either way, whether there is anything left to do.
#define CK_BITMAP_BITS 20
#define CK_U 8

// sorted_out (nullable): the rows in canonical order are written as well (sorted_out[rank] = row), which saves the
// gather of the coordinates by the permutation behind this kernel (codec.hip view_of, four times per GOP).
__global__ __launch_bounds__(SS_THREADS) void k_compact_coord_keys(const int4* __restrict__ coords, int n,
                                                                   uint64_t* __restrict__ keys,
                                                                   uint32_t* __restrict__ perm,
                                                                   int* __restrict__ done,
                                                                   int4* __restrict__ sorted_out) {
  __shared__ int s_mn[4], s_mx[4], s_bad, s_dup;
  __shared__ unsigned s_or[4];
  __shared__ uint32_t s_chunk[SS_THREADS], s_wsum[SS_WAVES];
  __shared__ uint32_t bm[(1 << CK_BITMAP_BITS) / 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 4) { s_mn[tid] = 0x7fffffff; s_mx[tid] = (int)0x80000000; s_or[tid] = 0u; }
  if (tid == 0) { s_bad = 0; s_dup = 0; }
  __syncthreads();
  // one sweep: min / max per field, range check, and the bits in which rows differ from row 0 (the lowest such
  // bit is the lowest set bit of OR(c - min): both say "all rows are congruent mod 2^tz")
  const int4 c0 = coords[0];
  int mn[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  int mx[4] = {(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  unsigned orv[4] = {0u, 0u, 0u, 0u};
  bool bad = false;
  for (int e0 = tid; e0 < n; e0 += CK_U * SS_THREADS) {
    int4 cc[CK_U];  // CK_U independent loads in flight per lane (clamped index: duplicates do not change min / max / or)
#pragma unroll
    for (int u = 0; u < CK_U; ++u) cc[u] = coords[min(e0 + u * SS_THREADS, n - 1)];
#pragma unroll
    for (int u = 0; u < CK_U; ++u) {
      const int4 c = cc[u];
      const int v[4] = {c.x, c.y, c.z, c.w};
      const int r[4] = {c0.x, c0.y, c0.z, c0.w};
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        mn[f] = min(mn[f], v[f]);
        mx[f] = max(mx[f], v[f]);
        orv[f] |= (unsigned)(v[f] ^ r[f]);
      }
      bad |= (c.x < 0) | (c.y <= -50000) | (c.y >= 50000) | (c.z <= -50000) | (c.z >= 50000) | (c.w <= -50000) |
             (c.w >= 50000);
    }
  }
  // reduce in the wave first: 1024 lanes hitting the same 12 LDS words serialise (that alone was ~30 us)
#pragma unroll
  for (int f = 0; f < 4; ++f) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      mn[f] = min(mn[f], __shfl_xor(mn[f], d, 64));
      mx[f] = max(mx[f], __shfl_xor(mx[f], d, 64));
      orv[f] |= (unsigned)__shfl_xor((int)orv[f], d, 64);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      atomicMin(&s_mn[f], mn[f]);
      atomicMax(&s_mx[f], mx[f]);
      atomicOr(&s_or[f], orv[f]);
    }
  }
  if (__any(bad) && lane == 0) atomicOr(&s_bad, 1);
  __syncthreads();
  int tz[4], w[4], total = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const unsigned o = s_or[f];
    tz[f] = o ? __builtin_ctz(o) : 0;
    const unsigned span = ((unsigned)(s_mx[f] - s_mn[f])) >> tz[f];
    w[f] = span ? 32 - __builtin_clz(span) : 0;
    total += w[f];
  }
  const bool compact = !s_bad && total <= 63;
  const int m0 = s_mn[0], m1 = s_mn[1], m2 = s_mn[2], m3 = s_mn[3];
  auto compact_key = [&](const int4 c) -> uint64_t {
    uint64_t k = (uint64_t)((unsigned)(c.x - m0) >> tz[0]);
    k = (k << w[1]) | (uint64_t)((unsigned)(c.y - m1) >> tz[1]);
    k = (k << w[2]) | (uint64_t)((unsigned)(c.z - m2) >> tz[2]);
    k = (k << w[3]) | (uint64_t)((unsigned)(c.w - m3) >> tz[3]);
    return k;
  };

  if (compact && total <= CK_BITMAP_BITS) {  // block-uniform
    const int nwords = max((1 << total) >> 5, 1);
    for (int j = tid; j < nwords; j += SS_THREADS) bm[j] = 0u;
    __syncthreads();
    bool dup = false;
    for (int e0 = tid; e0 < n; e0 += CK_U * SS_THREADS) {
      int4 cc[CK_U];
#pragma unroll
      for (int u = 0; u < CK_U; ++u) cc[u] = coords[min(e0 + u * SS_THREADS, n - 1)];
#pragma unroll
      for (int u = 0; u < CK_U; ++u) {
        if (e0 + u * SS_THREADS < n) {
          const uint32_t k = (uint32_t)compact_key(cc[u]);
          const uint32_t bit = 1u << (k & 31u);
          dup |= (atomicOr(&bm[k >> 5], bit) & bit) != 0u;
        }
      }
    }
    if (__any(dup) && lane == 0) atomicOr(&s_dup, 1);
    __syncthreads();
    if (!s_dup) {
      // occupied keys below each chunk of nw words (thread t owns words [t*nw, (t+1)*nw))
      const int nw = (nwords + SS_THREADS - 1) / SS_THREADS;
      uint32_t sum = 0;
      for (int j = tid * nw; j < min((tid + 1) * nw, nwords); ++j) sum += (uint32_t)__popc(bm[j]);
      uint32_t inc = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
      }
      if (lane == 63) s_wsum[wave] = inc;
      __syncthreads();
      uint32_t base = inc - sum;
      for (int q = 0; q < wave; ++q) base += s_wsum[q];
      s_chunk[tid] = base;
      __syncthreads();
      for (int e0 = tid; e0 < n; e0 += CK_U * SS_THREADS) {
        int4 cc[CK_U];
#pragma unroll
        for (int u = 0; u < CK_U; ++u) cc[u] = coords[min(e0 + u * SS_THREADS, n - 1)];
#pragma unroll
        for (int u = 0; u < CK_U; ++u) {
          const int e = e0 + u * SS_THREADS;
          if (e < n) {
            const uint32_t k = (uint32_t)compact_key(cc[u]);
            const int wi = (int)(k >> 5), ch = wi / nw;
            uint32_t r = s_chunk[ch];
            for (int j = ch * nw; j < wi; ++j) r += (uint32_t)__popc(bm[j]);
            r += (uint32_t)__popc(bm[wi] & ((1u << (k & 31u)) - 1u));
            perm[r] = (uint32_t)e;
            if (sorted_out) sorted_out[r] = cc[u];
          }
        }
      }
      if (tid == 0) *done = 1;
      return;
    }
  }
  if (tid == 0) *done = 0;
  for (int e0 = tid; e0 < n; e0 += CK_U * SS_THREADS) {
    int4 cc[CK_U];
#pragma unroll
    for (int u = 0; u < CK_U; ++u) cc[u] = coords[min(e0 + u * SS_THREADS, n - 1)];
#pragma unroll
    for (int u = 0; u < CK_U; ++u) {
      const int e = e0 + u * SS_THREADS;
      const int4 c = cc[u];
      uint64_t k;
      if (compact) {
        k = compact_key(c);
      } else {
        const int64_t lin = (int64_t)c.x * 1000000000000000ll + (int64_t)c.y * 10000000000ll +
                            (int64_t)c.z * 100000ll + (int64_t)c.w;
        k = (uint64_t)lin ^ (1ull << 63);
      }
      if (e < n) keys[e] = k;
    }
  }
}

// ---- the same canonical order over several workgroups (latents of CKM_MIN rows and more) --------------------------
// One workgroup sweeps the rows three times with its own latency in front of every sweep (45 us for a 26k-row latent,
// four times per GOP step).  Four launches of many workgroups instead, each a plain parallel pass:
//   k_ckm_range : Morton key -> row (b, x, y, z), per-workgroup min / max / varying bits of every field; clears the bitmap
//   k_ckm_bits  : compact key of every row, one bit per key in a global bitmap (a bit found set = a repeated row)
//   k_ckm_scan  : one workgroup: occupied keys in front of every bitmap word
//   k_ckm_place : perm[rank(key)] = row, ordered rows; or, when the domain is too wide / a row repeats, the compact
//                 (or decimal) key of every row for k_sort_small, which is launched behind either way (`done`)
// Every workgroup derives the field widths from the per-workgroup ranges for itself (ckm_params).
constexpr int CKM_THREADS = 256;
constexpr int CKM_ROWS = 1024;          // rows per workgroup
constexpr int64_t CKM_MIN = 8192;       // below: the single-workgroup kernel (10-13 us at 2-3k rows)
struct CkmPart { int mn[4], mx[4]; unsigned orv[4]; int bad; };
struct CkmParams { int m[4], tz[4], w[4], total; bool compact; };

__device__ __forceinline__ CkmParams ckm_params(const CkmPart* __restrict__ part, int n_part, int* lds /* [16] */) {
  // n_part <= 64: one wave reduces the partial ranges, everybody reads the result
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    CkmPart p;
    if (lane < n_part) p = part[lane];
    else {
#pragma unroll
      for (int f = 0; f < 4; ++f) { p.mn[f] = 0x7fffffff; p.mx[f] = (int)0x80000000; p.orv[f] = 0u; }
      p.bad = 0;
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        p.mn[f] = min(p.mn[f], __shfl_xor(p.mn[f], d, 64));
        p.mx[f] = max(p.mx[f], __shfl_xor(p.mx[f], d, 64));
        p.orv[f] |= (unsigned)__shfl_xor((int)p.orv[f], d, 64);
      }
    }
    const int bad = __any(p.bad != 0) ? 1 : 0;
    if (lane == 0) {
#pragma unroll
      for (int f = 0; f < 4; ++f) { lds[f] = p.mn[f]; lds[4 + f] = p.mx[f]; lds[8 + f] = (int)p.orv[f]; }
      lds[12] = bad;
    }
  }
  __syncthreads();
  CkmParams q;
  q.total = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const unsigned o = (unsigned)lds[8 + f];
    q.m[f] = lds[f];
    q.tz[f] = o ? __builtin_ctz(o) : 0;
    const unsigned span = ((unsigned)(lds[4 + f] - lds[f])) >> q.tz[f];
    q.w[f] = span ? 32 - __builtin_clz(span) : 0;
    q.total += q.w[f];
  }
  q.compact = lds[12] == 0 && q.total <= 63;
  return q;
}
__device__ __forceinline__ uint64_t ckm_key(const CkmParams& q, const int4 c) {
  uint64_t k = (uint64_t)((unsigned)(c.x - q.m[0]) >> q.tz[0]);
  k = (k << q.w[1]) | (uint64_t)((unsigned)(c.y - q.m[1]) >> q.tz[1]);
  k = (k << q.w[2]) | (uint64_t)((unsigned)(c.z - q.m[2]) >> q.tz[2]);
  k = (k << q.w[3]) | (uint64_t)((unsigned)(c.w - q.m[3]) >> q.tz[3]);
  return k;
}

__global__ __launch_bounds__(CKM_THREADS) void k_ckm_range(const uint64_t* __restrict__ mkeys, int n, int4* __restrict__ coords,
                                                           CkmPart* __restrict__ part, uint32_t* __restrict__ bitmap,
                                                           int* __restrict__ flags /* [0] dup */) {
  __shared__ int s_mn[4], s_mx[4], s_bad;
  __shared__ unsigned s_or[4];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 4) { s_mn[tid] = 0x7fffffff; s_mx[tid] = (int)0x80000000; s_or[tid] = 0u; }
  if (tid == 0) s_bad = 0;
  if (blockIdx.x == 0 && tid == 0) flags[0] = 0;
  // this workgroup's share of the 2^20-bit bitmap
  const int words = (1 << CK_BITMAP_BITS) / 32, per = (words + gridDim.x - 1) / gridDim.x;
  for (int j = blockIdx.x * per + tid; j < min((int)(blockIdx.x + 1) * per, words); j += CKM_THREADS) bitmap[j] = 0u;
  __syncthreads();
  int b0, x0, y0, z0;
  pcc_unmorton(mkeys[0], &b0, &x0, &y0, &z0);
  const int r[4] = {b0, x0, y0, z0};
  int mn[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  int mx[4] = {(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  unsigned orv[4] = {0u, 0u, 0u, 0u};
  bool bad = false;
  const int e_lo = blockIdx.x * CKM_ROWS, e_hi = min(e_lo + CKM_ROWS, n);
  for (int e = e_lo + tid; e < e_hi; e += CKM_THREADS) {
    int b, x, y, z;
    pcc_unmorton(mkeys[e], &b, &x, &y, &z);
    coords[e] = make_int4(b, x, y, z);
    const int v[4] = {b, x, y, z};
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      mn[f] = min(mn[f], v[f]);
      mx[f] = max(mx[f], v[f]);
      orv[f] |= (unsigned)(v[f] ^ r[f]);
    }
    bad |= (b < 0) | (x <= -50000) | (x >= 50000) | (y <= -50000) | (y >= 50000) | (z <= -50000) | (z >= 50000);
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      mn[f] = min(mn[f], __shfl_xor(mn[f], d, 64));
      mx[f] = max(mx[f], __shfl_xor(mx[f], d, 64));
      orv[f] |= (unsigned)__shfl_xor((int)orv[f], d, 64);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int f = 0; f < 4; ++f) { atomicMin(&s_mn[f], mn[f]); atomicMax(&s_mx[f], mx[f]); atomicOr(&s_or[f], orv[f]); }
  }
  if (__any(bad) && lane == 0) atomicOr(&s_bad, 1);
  __syncthreads();
  if (tid == 0) {
    CkmPart p;
#pragma unroll
    for (int f = 0; f < 4; ++f) { p.mn[f] = s_mn[f]; p.mx[f] = s_mx[f]; p.orv[f] = s_or[f]; }
    p.bad = s_bad;
    part[blockIdx.x] = p;
  }
}

__global__ __launch_bounds__(CKM_THREADS) void k_ckm_bits(const int4* __restrict__ coords, int n, const CkmPart* __restrict__ part,
                                                          uint32_t* __restrict__ bitmap, int* __restrict__ flags) {
  __shared__ int lds[16];
  const CkmParams q = ckm_params(part, gridDim.x, lds);
  if (!(q.compact && q.total <= CK_BITMAP_BITS)) return;   // block-uniform
  const int e_lo = blockIdx.x * CKM_ROWS, e_hi = min(e_lo + CKM_ROWS, n);
  bool dup = false;
  for (int e = e_lo + threadIdx.x; e < e_hi; e += CKM_THREADS) {
    const uint32_t k = (uint32_t)ckm_key(q, coords[e]);
    const uint32_t bit = 1u << (k & 31u);
    dup |= (atomicOr(&bitmap[k >> 5], bit) & bit) != 0u;
  }
  if (__any(dup) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1);
}

// one workgroup of SS_THREADS: prefix[j] = occupied keys in the words in front of word j; *done = the order is made here.
// Thread t holds words t, t + 1024, t + 2048, ... (coalesced loads, all requested at once and before the widths are
// known: one memory round trip for the kernel).  Row i of 1024 consecutive words is scanned inside its 16 waves by
// shuffles; the 16 wave totals of every row are scanned once, in word order, by the first wave.  No large LDS array: a
// workgroup with a 128-KB bitmap image in LDS took 16 us here whatever it did, this one takes a plain launch.
__global__ __launch_bounds__(SS_THREADS) void k_ckm_scan(const CkmPart* __restrict__ part, int n_part,
                                                         const uint32_t* __restrict__ bitmap, uint32_t* __restrict__ prefix,
                                                         const int* __restrict__ flags, int* __restrict__ done) {
  constexpr int kPer = (1 << CK_BITMAP_BITS) / 32 / SS_THREADS;   // rows of 1024 words: 32
  __shared__ int lds[16];
  __shared__ uint32_t s_tot[kPer * SS_WAVES];   // totals of (row, wave), row-major = word order
  uint32_t pc[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) pc[i] = (uint32_t)__popc(bitmap[threadIdx.x + i * SS_THREADS]);
  const int dup = flags[0];
  const CkmParams q = ckm_params(part, n_part, lds);
  const bool ok = q.compact && q.total <= CK_BITMAP_BITS && dup == 0;
  if (threadIdx.x == 0) *done = ok ? 1 : 0;
  if (!ok) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwords = max((1 << q.total) >> 5, 1);
  const int rows = (nwords + SS_THREADS - 1) / SS_THREADS;   // block-uniform
  uint32_t ex[kPer];   // exclusive count inside the wave
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    if (i < rows) {
      const uint32_t c = tid + i * SS_THREADS < nwords ? pc[i] : 0u;
      uint32_t inc = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
      }
      ex[i] = inc - c;
      if (lane == 63) s_tot[i * SS_WAVES + wave] = inc;
    }
  }
  __syncthreads();
  if (wave == 0) {   // exclusive scan of the rows * 16 totals, 64 at a time
    uint32_t carry = 0;
    for (int b0 = 0; b0 < rows * SS_WAVES; b0 += 64) {
      const uint32_t c = s_tot[b0 + lane];
      uint32_t inc = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
      }
      s_tot[b0 + lane] = carry + inc - c;
      carry += __shfl(inc, 63, 64);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPer; ++i)
    if (i < rows && tid + i * SS_THREADS < nwords) prefix[tid + i * SS_THREADS] = s_tot[i * SS_WAVES + wave] + ex[i];
}

__global__ __launch_bounds__(CKM_THREADS) void k_ckm_place(const int4* __restrict__ coords, int n, const CkmPart* __restrict__ part,
                                                           const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ prefix,
                                                           const int* __restrict__ done, uint64_t* __restrict__ keys,
                                                           uint32_t* __restrict__ perm, int4* __restrict__ sorted_out) {
  __shared__ int lds[16];
  const CkmParams q = ckm_params(part, gridDim.x, lds);
  const bool placed = *done != 0;
  const int e_lo = blockIdx.x * CKM_ROWS, e_hi = min(e_lo + CKM_ROWS, n);
  for (int e = e_lo + threadIdx.x; e < e_hi; e += CKM_THREADS) {
    const int4 c = coords[e];
    if (placed) {
      const uint32_t k = (uint32_t)ckm_key(q, c);
      const uint32_t r = prefix[k >> 5] + (uint32_t)__popc(bitmap[k >> 5] & ((1u << (k & 31u)) - 1u));
      perm[r] = (uint32_t)e;
      if (sorted_out) sorted_out[r] = c;
    } else if (q.compact) {
      keys[e] = ckm_key(q, c);
    } else {
      const int64_t lin = (int64_t)c.x * 1000000000000000ll + (int64_t)c.y * 10000000000ll + (int64_t)c.z * 100000ll +
                          (int64_t)c.w;
      keys[e] = (uint64_t)lin ^ (1ull << 63);
    }
  }
}

extern "C" int pcc_sort_coords(pcc_ctx* ctx, const int32_t* d_coords, int64_t n,
                               uint32_t* d_perm) {
  PCC_REQUIRE(ctx && (n == 0 || (d_coords && d_perm)), PCC_E_ARG, "pcc_sort_coords: null arg");
  PCC_REQUIRE(n < ((int64_t)1 << 31), PCC_E_ARG, "pcc_sort_coords: n too large");
  if (n <= 0) return PCC_OK;
  PCC_TRY(pcc_arena_reserve(ctx, sort_scratch_bytes(n) + pcc_align((size_t)n * 8) + 256));
  int64_t* lk = (int64_t*)pcc_arena_alloc(ctx, (size_t)n * 8);
  int* done = (int*)pcc_arena_alloc(ctx, 4);
  if (!lk || !done) return PCC_E_NOMEM;
  PccProfScope prof(ctx, "sort_coords", n, 0, 0, 0);
  if (n <= SS_MAX) {
    hipLaunchKernelGGL(k_compact_coord_keys, dim3(1), dim3(SS_THREADS), 0, ctx->stream, (const int4*)d_coords, (int)n,
                       (uint64_t*)lk, d_perm, done, (int4*)nullptr);
    PCC_CHECK_LAUNCH();
    return sort_pairs_impl(ctx, (uint64_t*)lk, d_perm, n, 0, done);
  }
  hipLaunchKernelGGL(k_linear_keys, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream,
                     (const int4*)d_coords, n, lk);
  PCC_CHECK_LAUNCH();
  return sort_pairs_impl(ctx, (uint64_t*)lk, d_perm, n, 1);
}

// Internal (codec.hip view_of): canonical order of a coordinate set given by its Morton keys — the permutation AND the
// rows [n,4] (b,x,y,z) in that order, for sets within the single-workgroup kernels (n <= pcc_sort_small_max()): the
// keys are turned into coordinates by a plain parallel kernel (inside the one-workgroup kernel, which sweeps the rows
// three times, the bit de-interleaving cost 28 us for 26k rows), the order kernel writes the ordered rows itself.
int64_t pcc_sort_small_max() { return SS_MAX; }
int pcc_sort_keys_canonical(pcc_ctx* ctx, const uint64_t* d_mkeys, int64_t n, uint32_t* d_perm, int32_t* d_sorted_coords) {
  PCC_REQUIRE(ctx && d_mkeys && d_perm && d_sorted_coords && n >= 1 && n <= SS_MAX, PCC_E_ARG,
              "pcc_sort_keys_canonical: bad argument (n=%lld)", (long long)n);
  PCC_TRY(pcc_arena_reserve(ctx, sort_scratch_bytes(n) + pcc_align((size_t)n * 8) + pcc_align((size_t)n * 16) + 256 +
                                     2 * pcc_align((size_t)(1 << CK_BITMAP_BITS) / 8) + pcc_align(sizeof(CkmPart) * 64) + 512));
  int64_t* lk = (int64_t*)pcc_arena_alloc(ctx, (size_t)n * 8);
  int4* c = (int4*)pcc_arena_alloc(ctx, (size_t)n * 16);
  int* done = (int*)pcc_arena_alloc(ctx, 4);
  if (!lk || !c || !done) return PCC_E_NOMEM;
  PccProfScope prof(ctx, "sort_coords", n, 0, 0, 0);
  if (n >= CKM_MIN) {
    const int nb = (int)nblk(n, CKM_ROWS);   // <= 64 (n <= SS_MAX)
    CkmPart* part = (CkmPart*)pcc_arena_alloc(ctx, sizeof(CkmPart) * 64);
    uint32_t* bitmap = (uint32_t*)pcc_arena_alloc(ctx, (size_t)(1 << CK_BITMAP_BITS) / 8);
    uint32_t* prefix = (uint32_t*)pcc_arena_alloc(ctx, (size_t)(1 << CK_BITMAP_BITS) / 8);
    int* flags = (int*)pcc_arena_alloc(ctx, 16);
    if (!part || !bitmap || !prefix || !flags) return PCC_E_NOMEM;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_ckm_range, dim3(nb), dim3(CKM_THREADS), 0, st, d_mkeys, (int)n, c, part, bitmap, flags);
    PCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ckm_bits, dim3(nb), dim3(CKM_THREADS), 0, st, (const int4*)c, (int)n, (const CkmPart*)part, bitmap, flags);
    PCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ckm_scan, dim3(1), dim3(SS_THREADS), 0, st, (const CkmPart*)part, nb, (const uint32_t*)bitmap, prefix,
                       (const int*)flags, done);
    PCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ckm_place, dim3(nb), dim3(CKM_THREADS), 0, st, (const int4*)c, (int)n, (const CkmPart*)part,
                       (const uint32_t*)bitmap, (const uint32_t*)prefix, (const int*)done, (uint64_t*)lk, d_perm,
                       (int4*)d_sorted_coords);
    PCC_CHECK_LAUNCH();
    return sort_pairs_impl(ctx, (uint64_t*)lk, d_perm, n, 0, done, (const int4*)c, (int4*)d_sorted_coords);
  }
  hipLaunchKernelGGL(k_keys_to_coords, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_mkeys, n, c);
  PCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_compact_coord_keys, dim3(1), dim3(SS_THREADS), 0, ctx->stream, (const int4*)c, (int)n, (uint64_t*)lk,
                     d_perm, done, (int4*)d_sorted_coords);
  PCC_CHECK_LAUNCH();
  return sort_pairs_impl(ctx, (uint64_t*)lk, d_perm, n, 0, done, (const int4*)c, (int4*)d_sorted_coords);
}

extern "C" int pcc_gather_rows(pcc_ctx* ctx, const void* d_src, const uint32_t* d_perm,
                               int64_t n, int row_bytes, void* d_dst) {
  PCC_REQUIRE(ctx && (n == 0 || (d_src && d_perm && d_dst)), PCC_E_ARG, "pcc_gather_rows: null arg");
  PCC_REQUIRE(row_bytes > 0 && row_bytes % 4 == 0, PCC_E_ARG, "pcc_gather_rows: row_bytes %d", row_bytes);
  if (n <= 0) return PCC_OK;
  PccProfScope prof(ctx, "gather_rows", n, row_bytes, 0, 0);
  if (row_bytes % 16 == 0 && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0)) {
    int vec = row_bytes / 16;
    hipLaunchKernelGGL(k_gather_rows16, dim3(nblk(n * vec, 256)), dim3(256), 0, ctx->stream,
                       (const uint4*)d_src, d_perm, n, vec, (uint4*)d_dst);
  } else {
    int vec = row_bytes / 4;
    hipLaunchKernelGGL(k_gather_rows4, dim3(nblk(n * vec, 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)d_src, d_perm, n, vec, (uint32_t*)d_dst);
  }
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// internal (codec.hip): dst[perm[i]] = src[i] for a PERMUTATION perm of 0 .. n-1 — what pcc_inverse_rows followed by
// pcc_gather_rows computes, in one launch (rows given in canonical order back into the tensor's row order)
__global__ void k_scatter_rows16(const uint4* __restrict__ src, const uint32_t* __restrict__ perm, int64_t n, int vec,
                                 uint4* __restrict__ dst) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t / vec;
  const int j = (int)(t - row * vec);
  if (row >= n) return;
  dst[(int64_t)perm[row] * vec + j] = src[row * vec + j];
}
int pcc_scatter_rows(pcc_ctx* ctx, const void* d_src, const uint32_t* d_perm, int64_t n, int row_bytes, void* d_dst) {
  PCC_REQUIRE(ctx && (n == 0 || (d_src && d_perm && d_dst)), PCC_E_ARG, "pcc_scatter_rows: null arg");
  PCC_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0 && (uintptr_t)d_src % 16 == 0 && (uintptr_t)d_dst % 16 == 0, PCC_E_ARG,
              "pcc_scatter_rows: rows of %d bytes (16-byte pieces of aligned tensors)", row_bytes);
  if (n <= 0) return PCC_OK;
  PccProfScope prof(ctx, "scatter_rows", n, row_bytes, 0, 0);
  const int vec = row_bytes / 16;
  hipLaunchKernelGGL(k_scatter_rows16, dim3(nblk(n * vec, 256)), dim3(256), 0, ctx->stream, (const uint4*)d_src, d_perm, n, vec,
                     (uint4*)d_dst);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_gather_rows_or_zero(pcc_ctx* ctx, const float* d_src, const int32_t* d_rows,
                                       int64_t m, int c, float* d_dst) {
  PCC_REQUIRE(ctx && (m == 0 || (d_src && d_rows && d_dst)), PCC_E_ARG, "pcc_gather_rows_or_zero: null arg");
  PCC_REQUIRE(c > 0, PCC_E_ARG, "pcc_gather_rows_or_zero: c=%d", c);
  if (m <= 0) return PCC_OK;
  hipLaunchKernelGGL(k_gather_rows_or_zero, dim3(nblk(m * c, 256)), dim3(256), 0, ctx->stream,
                     d_src, d_rows, m, c, d_dst);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_check_unique(pcc_ctx* ctx, const uint64_t* d_sorted_keys, int64_t n, int* h_dup) {
  PCC_REQUIRE(ctx && h_dup && (n == 0 || d_sorted_keys), PCC_E_ARG, "pcc_check_unique: null arg");
  *h_dup = 0;
  if (n <= 1) return PCC_OK;
  PCC_TRY(pcc_arena_reserve(ctx, 256));
  int32_t* flag = (int32_t*)pcc_arena_alloc(ctx, 4);
  if (!flag) return PCC_E_NOMEM;
  PCC_HIP(hipMemsetAsync(flag, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_check_unique, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_sorted_keys, n, flag);
  PCC_CHECK_LAUNCH();
  int32_t* h = (int32_t*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(h, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  PCC_HIP(hipStreamSynchronize(ctx->stream));
  *h_dup = h[0];
  return PCC_OK;
}

extern "C" int pcc_batch_offsets(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int n_batch,
                                 int64_t* h_offsets) {
  PCC_REQUIRE(ctx && h_offsets && (n == 0 || d_keys), PCC_E_ARG, "pcc_batch_offsets: null arg");
  PCC_REQUIRE(n_batch >= 0 && (size_t)(n_batch + 1) * 8 <= ctx->pinned_cap, PCC_E_ARG,
              "pcc_batch_offsets: n_batch=%d unsupported", n_batch);
  if (n <= 0) {
    for (int b = 0; b <= n_batch; ++b) h_offsets[b] = 0;
    return PCC_OK;
  }
  PCC_TRY(pcc_arena_reserve(ctx, (size_t)(n_batch + 1) * 8 + 256));
  int64_t* offs = (int64_t*)pcc_arena_alloc(ctx, (size_t)(n_batch + 1) * 8);
  if (!offs) return PCC_E_NOMEM;
  hipLaunchKernelGGL(k_batch_offsets, dim3(nblk(n_batch + 1, 64)), dim3(64), 0, ctx->stream, d_keys, n,
                     n_batch, offs);
  PCC_CHECK_LAUNCH();
  PCC_HIP(hipMemcpyAsync(ctx->pinned, offs, (size_t)(n_batch + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  PCC_HIP(hipStreamSynchronize(ctx->stream));
  for (int b = 0; b <= n_batch; ++b) h_offsets[b] = ((int64_t*)ctx->pinned)[b];
  return PCC_OK;
}
