// map.hip — coordinate hash, 3^3 rule book (kernel map), exact-lattice lookup.
//
// Replaces MinkowskiEngine's coordinate hash map + kernel-map generation for
// the stride-1 3^3 convolutions of g_a/h_a/h_s/g_s and
// SparseTensor.features_at_coordinates (codec_pipeline.py:401,
// codec_parallel.py:387).  Open addressing, 64-bit Morton keys, linear
// probing at load factor <= 0.5; the table lives in the ctx arena (L2 /
// Infinity-Cache resident for every size on the path: 1M voxels -> 24 MB).
// The rule book is out-stationary: nbr[k][n] = input row or -1, written
// coalesced over n for each of the 27 offsets.
#include "common.h"

#define HASH_EMPTY (~0ull)

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

__device__ __forceinline__ uint64_t hash64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__global__ void k_hash_fill(unsigned long long* __restrict__ tk, int64_t cap) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) tk[i] = HASH_EMPTY;
}

__global__ void k_hash_insert(const uint64_t* __restrict__ keys, int64_t n,
                              unsigned long long* __restrict__ tk, uint32_t* __restrict__ tv,
                              uint64_t mask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  uint64_t slot = hash64(key) & mask;
  // bounded: the table has >= 2n slots, so a free slot is met within cap steps
  for (uint64_t step = 0; step <= mask; ++step) {
    unsigned long long old = atomicCAS(&tk[slot], (unsigned long long)HASH_EMPTY,
                                       (unsigned long long)key);
    if (old == HASH_EMPTY || old == key) {
      tv[slot] = (uint32_t)i;
      return;
    }
    slot = (slot + 1) & mask;
  }
}

__device__ __forceinline__ int32_t hash_find(const unsigned long long* __restrict__ tk,
                                             const uint32_t* __restrict__ tv, uint64_t mask,
                                             uint64_t key) {
  uint64_t slot = hash64(key) & mask;
  for (uint64_t step = 0; step <= mask; ++step) {
    const unsigned long long k = tk[slot];
    if (k == key) return (int32_t)tv[slot];
    if (k == HASH_EMPTY) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}

// one thread per output row; 27 probes, writes coalesced per offset
__global__ __launch_bounds__(256) void k_build_map27(
    const uint64_t* __restrict__ keys, int64_t n, int stride,
    const unsigned long long* __restrict__ tk, const uint32_t* __restrict__ tv, uint64_t mask,
    int32_t* __restrict__ nbr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  int b, x, y, z;
  pcc_unmorton(key, &b, &x, &y, &z);
  uint64_t sx[3], sy[3], sz[3];
  bool okx[3], oky[3], okz[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int nx = x + (d - 1) * stride, ny = y + (d - 1) * stride, nz = z + (d - 1) * stride;
    okx[d] = (nx >= -32768) & (nx <= 32767);
    oky[d] = (ny >= -32768) & (ny <= 32767);
    okz[d] = (nz >= -32768) & (nz <= 32767);
    sx[d] = pcc_spread3((uint32_t)(nx + 32768)) << 2;
    sy[d] = pcc_spread3((uint32_t)(ny + 32768)) << 1;
    sz[d] = pcc_spread3((uint32_t)(nz + 32768));
  }
  const uint64_t bk = (uint64_t)(uint32_t)b << 48;
  // The 26 probes of a row are independent: their table reads are issued together, round by round (home slot,
  // +1, +2, ... until no lane of the wave has a probe pending): one memory latency per round instead of one per
  // probe step per offset — with 1728 probes per wave some probe of nearly every offset runs long, so walking
  // offset by offset cost the latent-sized levels (a handful of workgroups) 27 serial chains.  Later rounds read
  // all 27 slots unconditionally so that they stay batched; the table is sized for a load factor <= 0.25.
  uint64_t q[27];
  uint32_t slot[27];
  uint32_t pend = 0, hit = 0;  // bit k: still searching / found at slot[k]
  {
    unsigned long long got[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int dx = k / 9, dy = (k / 3) % 3, dz = k % 3;
      const bool ok = (k != 13) & okx[dx] & oky[dy] & okz[dz];
      q[k] = bk | sx[dx] | sy[dy] | sz[dz];
      slot[k] = (uint32_t)(hash64(q[k]) & mask);
      got[k] = ok ? tk[slot[k]] : HASH_EMPTY;
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      if (got[k] == q[k]) hit |= 1u << k;
      else if (got[k] != HASH_EMPTY) pend |= 1u << k;
    }
  }
  for (uint32_t rd = 1; rd <= (uint32_t)mask && __any(pend != 0); ++rd) {  // wave-uniform
    unsigned long long got[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) got[k] = tk[(slot[k] + rd) & (uint32_t)mask];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      if ((pend >> k) & 1u) {
        if (got[k] == q[k]) {
          hit |= 1u << k;
          pend &= ~(1u << k);
          slot[k] = (slot[k] + rd) & (uint32_t)mask;
        } else if (got[k] == HASH_EMPTY) {
          pend &= ~(1u << k);
        }
      }
    }
  }
  int32_t r[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) r[k] = (int32_t)tv[slot[k]];  // unconditional: batched; slot[k] is always in range
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    if (k == 13) r[k] = (int32_t)i;
    else if (!((hit >> k) & 1u)) r[k] = -1;
    nbr[(int64_t)k * n + i] = r[k];
  }
}

// The coarsest levels of a pyramid are a few thousand rows (the z level of the bench frame: 1594): hash fill + insert +
// probe rounds are three launches of pure latency for them.  Here every workgroup copies the whole (Morton-sorted) key
// array into LDS and a thread finds ONE neighbour of one row by binary search there (row = thread / 27... laid out
// offset-major: workgroup (bx, k) serves offset k of rows 256 bx ..) — one launch, a dozen dependent LDS reads per
// thread.  (With the 26 searches of a row in one thread, advancing together, the launch took 28 us at ~170 registers.)
#define MAP_SMALL_MAX 4096
__global__ __launch_bounds__(256) void k_build_map27_small(const uint64_t* __restrict__ keys, int n, int stride,
                                                           int32_t* __restrict__ nbr) {
  __shared__ uint64_t sk[MAP_SMALL_MAX];
  for (int e = threadIdx.x; e < n; e += 256) sk[e] = keys[e];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= n) return;
  int32_t r = -1;
  if (k == 13) {
    r = i;
  } else {
    int b, x, y, z;
    pcc_unmorton(sk[i], &b, &x, &y, &z);
    const int nx = x + (k / 9 - 1) * stride, ny = y + ((k / 3) % 3 - 1) * stride, nz = z + (k % 3 - 1) * stride;
    if (nx >= -32768 && nx <= 32767 && ny >= -32768 && ny <= 32767 && nz >= -32768 && nz <= 32767) {
      const uint64_t q = ((uint64_t)(uint32_t)b << 48) | (pcc_spread3((uint32_t)(nx + 32768)) << 2) |
                         (pcc_spread3((uint32_t)(ny + 32768)) << 1) | pcc_spread3((uint32_t)(nz + 32768));
      int lo = 0, hi = n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sk[mid] < q) lo = mid + 1;
        else hi = mid;
      }
      if (lo < n && sk[lo] == q) r = lo;
    }
  }
  nbr[(int64_t)k * n + i] = r;
}

__global__ void k_lookup(const uint64_t* __restrict__ qkeys, int64_t m,
                         const unsigned long long* __restrict__ tk, const uint32_t* __restrict__ tv,
                         uint64_t mask, int32_t* __restrict__ rows) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  rows[i] = hash_find(tk, tv, mask, qkeys[i]);
}

// ---- rule books derived from the parent level's rule book (no hashing) -------
// A child voxel (parent p, octant o) has its neighbour at offset d in the child
// of parent-neighbour p + D with octant o', where per axis t = o + d, D =
// floor(t/2), o' = t mod 2.  So nbr_child[k][c] is two table lookups.
__device__ __forceinline__ void child_step(int o, int k, int* kp, int* op) {
  const int ox = (o >> 2) & 1, oy = (o >> 1) & 1, oz = o & 1;
  const int tx = ox + (k / 9) - 1, ty = oy + ((k / 3) % 3) - 1, tz = oz + (k % 3) - 1;
  *kp = ((tx + 2) >> 1) * 9 + ((ty + 2) >> 1) * 3 + ((tz + 2) >> 1);
  *op = ((tx & 1) << 2) | ((ty & 1) << 1) | (tz & 1);
}

// generative children (all 8 exist): child row = 8 * parent row + octant.
// parent_rows / remap translate between the pruned parent level (rows j) and the
// candidate level its rule book was built on: prow = parent_rows[j], and a
// neighbour candidate row r maps back to the pruned row remap[r] (or -1).
__global__ __launch_bounds__(256) void k_derive_up(const int32_t* __restrict__ nbr_p, int64_t pitch_p,
                                                   const uint32_t* __restrict__ parent_rows,
                                                   const int32_t* __restrict__ remap, int64_t n_par,
                                                   int32_t* __restrict__ nbr) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nc = n_par * 8;
  if (c >= nc) return;
  const int64_t j = c >> 3;
  const int o = (int)(c & 7);
  const int64_t prow = parent_rows ? (int64_t)parent_rows[j] : j;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    int kp, op;
    child_step(o, k, &kp, &op);
    int32_t pr = nbr_p[(int64_t)kp * pitch_p + prow];
    if (pr >= 0 && remap) pr = remap[pr];
    nbr[(int64_t)k * nc + c] = pr < 0 ? -1 : ((pr << 3) | op);
  }
}

// Rule book of a PRUNED level straight from the rule book of the level its candidates were generated from:
// pruned row j is candidate c = keep[j] = child o of parent p; its neighbour at offset k is the candidate
// (nbr_p[kp][p] << 3 | o'), which maps to the pruned row remap[candidate] or -1.  The candidate level's own
// 27 x 8N rule book never exists.
__global__ __launch_bounds__(256) void k_subset_map_up(const int32_t* __restrict__ nbr_p, int64_t pitch_p,
                                                       const uint32_t* __restrict__ keep,
                                                       const int32_t* __restrict__ remap, int64_t n_keep,
                                                       int32_t* __restrict__ nbr) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_keep) return;
  const uint32_t c = keep[j];
  const int64_t p = c >> 3;
  const int o = (int)(c & 7u);
  int32_t pr[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) {  // the 27 parent-book reads are independent: issue them together
    int kp, op;
    child_step(o, k, &kp, &op);
    const int32_t q = nbr_p[(int64_t)kp * pitch_p + p];
    pr[k] = q < 0 ? -1 : ((q << 3) | op);
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) pr[k] = pr[k] < 0 ? -1 : remap[pr[k]];
#pragma unroll
  for (int k = 0; k < 27; ++k) nbr[(int64_t)k * n_keep + j] = pr[k];
}

// Rule-book columns of a latent's voxels among the 64 generated descendants of their stride-32 ancestors (h_s ends in a
// conv that is sampled at the latent's coordinates only).  The descendants are row = (8 z_row + octant at stride 16)
// * 8 + octant at stride 8, and the two stride-2 maps that produced z from y hold the ancestors: the row of voxel
// perm[j], then its 27 neighbours from the book of the 8 z_rows rows at stride 16 — the book of the 64 z_rows
// descendants never exists, nor the row list.
__global__ __launch_bounds__(256) void k_descendant_map(const int32_t* __restrict__ nbr_p, int64_t pitch_p,
                                                        const uint32_t* __restrict__ perm,
                                                        const uint64_t* __restrict__ ykeys,
                                                        const int32_t* __restrict__ parent_of8,
                                                        const int32_t* __restrict__ parent_of16, int64_t m,
                                                        int32_t* __restrict__ nbr) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const uint32_t r = perm[j];
  const uint64_t key = ykeys[r];
  const int64_t p = (int64_t)parent_of16[parent_of8[r]] * 8 + (int64_t)((key >> 12) & 7ull);
  const int o = (int)((key >> 9) & 7ull);
  int32_t pr[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    int kp, op;
    child_step(o, k, &kp, &op);
    const int32_t q = nbr_p[(int64_t)kp * pitch_p + p];
    pr[k] = q < 0 ? -1 : ((q << 3) | op);
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) nbr[(int64_t)k * m + j] = pr[k];
}

int pcc_descendant_map(pcc_ctx* ctx, const int32_t* d_nbr_parent, int64_t parent_pitch, const uint32_t* d_perm,
                       const uint64_t* d_ykeys, const int32_t* d_parent_of8, const int32_t* d_parent_of16, int64_t m,
                       int32_t* d_nbr) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_descendant_map: null ctx");
  if (m <= 0) return PCC_OK;
  PCC_REQUIRE(d_nbr_parent && d_perm && d_ykeys && d_parent_of8 && d_parent_of16 && d_nbr && parent_pitch >= 1 &&
                  parent_pitch < ((int64_t)1 << 27), PCC_E_ARG, "pcc_descendant_map: bad buffers");
  PccProfScope prof(ctx, "descendant_map", m, parent_pitch, 0, 27);
  hipLaunchKernelGGL(k_descendant_map, dim3(nblk(m, 256)), dim3(256), 0, ctx->stream, d_nbr_parent, parent_pitch, d_perm,
                     d_ykeys, d_parent_of8, d_parent_of16, m, d_nbr);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// children given by a stride-2 map (parent_of, nbr8): second lookup through nbr8
__global__ __launch_bounds__(256) void k_derive_down(const int32_t* __restrict__ nbr_p, int64_t n_par,
                                                     const int32_t* __restrict__ nbr8,
                                                     const int32_t* __restrict__ parent_of,
                                                     const uint64_t* __restrict__ keys, int64_t n, int cshift,
                                                     int32_t* __restrict__ nbr) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int64_t p = parent_of[c];
  const int o = (int)((keys[c] >> cshift) & 7ull);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    int kp, op;
    child_step(o, k, &kp, &op);
    const int32_t pr = nbr_p[(int64_t)kp * n_par + p];
    nbr[(int64_t)k * n + c] = pr < 0 ? -1 : nbr8[(int64_t)op * n_par + pr];
  }
}

__global__ void k_fill_neg1(int32_t* __restrict__ p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = -1;
}
__global__ void k_inverse_rows(const uint32_t* __restrict__ rows, int64_t m, int32_t* __restrict__ remap) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) remap[rows[j]] = (int32_t)j;
}

extern "C" int pcc_derive_map_up(pcc_ctx* ctx, const int32_t* d_nbr_parent, int64_t parent_pitch,
                                 const uint32_t* d_parent_rows, const int32_t* d_remap, int64_t n_parents,
                                 int32_t* d_nbr) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_derive_map_up: null ctx");
  if (n_parents <= 0) return PCC_OK;
  PCC_REQUIRE(d_nbr_parent && d_nbr && parent_pitch >= 1, PCC_E_ARG, "pcc_derive_map_up: bad buffers");
  PCC_REQUIRE((d_parent_rows == nullptr) == (d_remap == nullptr), PCC_E_ARG,
              "pcc_derive_map_up: parent_rows and remap go together");
  PCC_REQUIRE(n_parents < ((int64_t)1 << 27), PCC_E_ARG, "pcc_derive_map_up: n too large");
  PccProfScope prof(ctx, "derive_map_up", n_parents * 8, parent_pitch, d_remap ? 1 : 0, 27);
  hipLaunchKernelGGL(k_derive_up, dim3(nblk(n_parents * 8, 256)), dim3(256), 0, ctx->stream, d_nbr_parent,
                     parent_pitch, d_parent_rows, d_remap, n_parents, d_nbr);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// rule book of a subset of a level's rows: out[k][j] = nbr[k][rows[j]] (or -1 where rows[j] < 0)
__global__ __launch_bounds__(256) void k_gather_map_columns(const int32_t* __restrict__ nbr, int k_vol, int64_t pitch,
                                                            const int32_t* __restrict__ rows, int64_t m,
                                                            int32_t* __restrict__ out, int32_t* __restrict__ self) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const int32_t r = rows[j];
  for (int k = 0; k < k_vol; ++k) out[(int64_t)k * m + j] = r >= 0 ? nbr[(int64_t)k * pitch + r] : -1;
  if (self) self[j] = r >= 0 ? (int32_t)j : -1;
}

extern "C" int pcc_gather_map_columns(pcc_ctx* ctx, const int32_t* d_nbr, int k_vol, int64_t pitch,
                                      const int32_t* d_rows, int64_t m, int32_t* d_nbr_out, int32_t* d_self) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_gather_map_columns: null ctx");
  if (m <= 0) return PCC_OK;
  PCC_REQUIRE(d_nbr && d_rows && d_nbr_out && k_vol >= 1 && k_vol <= 27 && pitch >= 1, PCC_E_ARG,
              "pcc_gather_map_columns: bad buffers");
  PccProfScope prof(ctx, "gather_map_columns", m, k_vol, 0, 0);
  hipLaunchKernelGGL(k_gather_map_columns, dim3(nblk(m, 256)), dim3(256), 0, ctx->stream, d_nbr, k_vol, pitch, d_rows,
                     m, d_nbr_out, d_self);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_subset_map_up(pcc_ctx* ctx, const int32_t* d_nbr_parent, int64_t parent_pitch,
                                 const uint32_t* d_keep, const int32_t* d_remap, int64_t n_keep, int32_t* d_nbr) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_subset_map_up: null ctx");
  if (n_keep <= 0) return PCC_OK;
  PCC_REQUIRE(d_nbr_parent && d_keep && d_remap && d_nbr && parent_pitch >= 1, PCC_E_ARG,
              "pcc_subset_map_up: bad buffers");
  PccProfScope prof(ctx, "subset_map_up", n_keep, parent_pitch, 0, 27);
  hipLaunchKernelGGL(k_subset_map_up, dim3(nblk(n_keep, 256)), dim3(256), 0, ctx->stream, d_nbr_parent,
                     parent_pitch, d_keep, d_remap, n_keep, d_nbr);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_derive_map_down(pcc_ctx* ctx, const int32_t* d_nbr_parent, int64_t n_parent,
                                   const int32_t* d_nbr8, const int32_t* d_parent_of, const uint64_t* d_keys,
                                   int64_t n, int child_shift, int32_t* d_nbr) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_derive_map_down: null ctx");
  PCC_REQUIRE(child_shift >= 0 && child_shift <= 42 && child_shift % 3 == 0, PCC_E_ARG,
              "pcc_derive_map_down: child_shift=%d", child_shift);
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_nbr_parent && d_nbr8 && d_parent_of && d_keys && d_nbr && n_parent >= 1, PCC_E_ARG,
              "pcc_derive_map_down: bad buffers");
  PccProfScope prof(ctx, "derive_map_down", n, n_parent, 0, 27);
  hipLaunchKernelGGL(k_derive_down, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_nbr_parent, n_parent,
                     d_nbr8, d_parent_of, d_keys, n, child_shift, d_nbr);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_inverse_rows(pcc_ctx* ctx, const uint32_t* d_rows, int64_t m, int64_t n, int32_t* d_remap) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_inverse_rows: null ctx");
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_remap && (m == 0 || d_rows) && m <= n, PCC_E_ARG, "pcc_inverse_rows: bad buffers");
  if (m < n) {  // a full permutation (m == n; the caller's rows are distinct) writes every entry itself
    hipLaunchKernelGGL(k_fill_neg1, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_remap, n);
    PCC_CHECK_LAUNCH();
  }
  if (m > 0) {
    hipLaunchKernelGGL(k_inverse_rows, dim3(nblk(m, 256)), dim3(256), 0, ctx->stream, d_rows, m, d_remap);
    PCC_CHECK_LAUNCH();
  }
  return PCC_OK;
}

static int64_t hash_capacity(int64_t n) {
  int64_t cap = 1024;
  while (cap < 4 * n) cap <<= 1;  // load factor <= 0.25: short probe sequences (the rounds of k_build_map27)
  return cap;
}

static int build_table(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, unsigned long long** tk,
                       uint32_t** tv, uint64_t* mask) {
  const int64_t cap = hash_capacity(n);
  *tk = (unsigned long long*)pcc_arena_alloc(ctx, (size_t)cap * 8);
  *tv = (uint32_t*)pcc_arena_alloc(ctx, (size_t)cap * 4);
  if (!*tk || !*tv) return PCC_E_NOMEM;
  *mask = (uint64_t)cap - 1;
  hipLaunchKernelGGL(k_hash_fill, dim3(nblk(cap, 256)), dim3(256), 0, ctx->stream, *tk, cap);
  PCC_CHECK_LAUNCH();
  if (n > 0) {
    hipLaunchKernelGGL(k_hash_insert, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_keys, n,
                       *tk, *tv, *mask);
    PCC_CHECK_LAUNCH();
  }
  return PCC_OK;
}

// Latent-sized levels (a few thousand to a few ten thousand rows): one thread per (row, offset) instead of 27 batched
// probes per row — 27 times the threads for a launch that was a handful of workgroups waiting on 27-deep register
// arrays (34 us for 26k rows); grid.y = offset, so a workgroup's stores are one coalesced run of its offset's row.
__global__ __launch_bounds__(256) void k_build_map27_each(const uint64_t* __restrict__ keys, int n, int stride,
                                                          const unsigned long long* __restrict__ tk,
                                                          const uint32_t* __restrict__ tv, uint64_t mask,
                                                          int32_t* __restrict__ nbr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= n) return;
  int32_t r = -1;
  if (k == 13) {
    r = i;
  } else {
    int b, x, y, z;
    pcc_unmorton(keys[i], &b, &x, &y, &z);
    const int nx = x + (k / 9 - 1) * stride, ny = y + ((k / 3) % 3 - 1) * stride, nz = z + (k % 3 - 1) * stride;
    if (nx >= -32768 && nx <= 32767 && ny >= -32768 && ny <= 32767 && nz >= -32768 && nz <= 32767) {
      const uint64_t q = ((uint64_t)(uint32_t)b << 48) | (pcc_spread3((uint32_t)(nx + 32768)) << 2) |
                         (pcc_spread3((uint32_t)(ny + 32768)) << 1) | pcc_spread3((uint32_t)(nz + 32768));
      uint32_t slot = (uint32_t)(hash64(q) & mask);
      for (uint64_t step = 0; step <= mask; ++step) {
        const unsigned long long got = tk[slot];
        if (got == q) { r = (int32_t)tv[slot]; break; }
        if (got == HASH_EMPTY) break;
        slot = (slot + 1) & (uint32_t)mask;
      }
    }
  }
  nbr[(int64_t)k * n + i] = r;
}

extern "C" int pcc_build_map(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int stride,
                             int32_t* d_nbr) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_build_map: null ctx");
  PCC_REQUIRE(stride >= 1 && stride <= 16384 && (stride & (stride - 1)) == 0, PCC_E_ARG,
              "pcc_build_map: stride=%d must be a power of two", stride);
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_keys && d_nbr, PCC_E_ARG, "pcc_build_map: null buffers");
  PCC_REQUIRE(n < ((int64_t)1 << 30), PCC_E_ARG, "pcc_build_map: n too large");
  PccProfScope prof(ctx, "build_map", n, stride, 0, 27);
  if (n <= MAP_SMALL_MAX) {  // keys are Morton-sorted (every coordinate set of the path is): binary search in LDS
    hipLaunchKernelGGL(k_build_map27_small, dim3(nblk(n, 256), 27), dim3(256), 0, ctx->stream, d_keys, (int)n, stride,
                       d_nbr);
    PCC_CHECK_LAUNCH();
    return PCC_OK;
  }
  const int64_t cap = hash_capacity(n);
  PCC_TRY(pcc_arena_reserve(ctx, pcc_align((size_t)cap * 8) + pcc_align((size_t)cap * 4) + 512));
  unsigned long long* tk;
  uint32_t* tv;
  uint64_t mask;
  PCC_TRY(build_table(ctx, d_keys, n, &tk, &tv, &mask));
  if (n <= 65536)
    hipLaunchKernelGGL(k_build_map27_each, dim3(nblk(n, 256), 27), dim3(256), 0, ctx->stream, d_keys, (int)n, stride,
                       (const unsigned long long*)tk, (const uint32_t*)tv, mask, d_nbr);
  else
    hipLaunchKernelGGL(k_build_map27, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_keys, n,
                       stride, (const unsigned long long*)tk, (const uint32_t*)tv, mask, d_nbr);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_lookup(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, const uint64_t* d_qkeys,
                          int64_t m, int32_t* d_rows) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_lookup: null ctx");
  if (m <= 0) return PCC_OK;
  PCC_REQUIRE(d_qkeys && d_rows && (n == 0 || d_keys), PCC_E_ARG, "pcc_lookup: null buffers");
  PCC_REQUIRE(n < ((int64_t)1 << 30), PCC_E_ARG, "pcc_lookup: n too large");
  const int64_t cap = hash_capacity(n);
  PCC_TRY(pcc_arena_reserve(ctx, pcc_align((size_t)cap * 8) + pcc_align((size_t)cap * 4) + 512));
  unsigned long long* tk;
  uint32_t* tv;
  uint64_t mask;
  PccProfScope prof(ctx, "lookup", n, m, 0, 0);
  PCC_TRY(build_table(ctx, d_keys, n, &tk, &tv, &mask));
  hipLaunchKernelGGL(k_lookup, dim3(nblk(m, 256)), dim3(256), 0, ctx->stream, d_qkeys, m,
                     (const unsigned long long*)tk, (const uint32_t*)tv, mask, d_rows);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}
