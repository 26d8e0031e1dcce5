// topk.hip — per-frame top-k occupancy pruning.
//
// Replaces the MinkowskiPruning + per-batch torch.topk inside
// model.g_s(y_hat, k=ks) (codec_parallel.py:469): of the candidate voxels of
// frame b keep the k[b] with the largest occupancy logit.  Exact and
// order-independent: the threshold is found by an MSB-first radix select on
// the order-preserving integer image of the logit (4 histogram passes), ties at
// the threshold are resolved by row order (lower Morton key first); one prefix
// scan over the "above" and "equal" flags ranks both and places the survivors
// (stable compaction).  6 launches per call (was 20, then 9, then 7: every block advances the
// threshold state itself from the finished histograms instead of a one-block update
// launch per pass, and the flags / scan / placement are two launches over per-frame
// tiles instead of four): at the sizes of this path a call is launch-bound, not
// bandwidth-bound.
// No sort of the 2M candidates, no floating-point comparisons that could
// differ between encoder, decoder and the CPU oracle.
#include "common.h"
#include <string.h>
#include <algorithm>

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

struct TopkState {
  uint32_t prefix;     // threshold key bits fixed so far
  uint32_t k_rem;      // how many still to take among keys matching the prefix
  int32_t mode;        // 0 keep none, 1 keep all, 2 select
  uint32_t keep_base;  // rows kept in the frames before this one (= sum of min(k, count): known on the host)
};

// per-frame parameters of a small GOP travel as kernel arguments: no pinned staging, hence no stream synchronisation on
// their account, and no launch that would only copy them into HBM
#define TOPK_ARG_FRAMES 8
struct TopkArgs {
  int64_t offs[TOPK_ARG_FRAMES + 1];
  TopkState st[TOPK_ARG_FRAMES];
};
struct TopkIn {
  const int64_t* offs;      // by_arg == 0: in HBM (staged through pinned memory)
  const TopkState* state;
  int by_arg;
  TopkArgs a;
};
__device__ __forceinline__ int64_t tk_off(const TopkIn& p, int f) { return p.by_arg ? p.a.offs[f] : p.offs[f]; }
__device__ __forceinline__ TopkState tk_state(const TopkIn& p, int f) { return p.by_arg ? p.a.st[f] : p.state[f]; }
// the histograms of a small GOP live in two buffers of the context used in turn: the last launch of a call clears the
// buffer of the NEXT call (nobody reads that one any more: stream order), so no launch exists to clear them either
constexpr int kTopkHistWords = 4 * 256 * TOPK_ARG_FRAMES;
__device__ __forceinline__ void tk_clear_next(uint32_t* __restrict__ next_hist) {
  if (!next_hist) return;
  const int gsz = gridDim.x * gridDim.y * blockDim.x;
  for (int j = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; j < kTopkHistWords; j += gsz) next_hist[j] = 0u;
}

__device__ __forceinline__ uint32_t ordered_key(float v) {
  uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// The threshold after `passes` histogram passes, recomputed by every block for itself from the initial state and the
// finished histograms (hist layout [pass][frame][256]): for each pass the digit d with
//   (keys above d among the candidates) < k_rem <= (those + keys with digit d),
// walking from the top digit down and stopping at d = 0 — k_topk_update's serial walk as a 256-lane suffix sum.
// The first 256 threads of the block do the walk (blocks of 256 or more threads; every thread gets the result); a few
// hundred cycles per pass, against one launch per pass saved.
__device__ __forceinline__ TopkState topk_state_after(TopkState s, const uint32_t* __restrict__ hist, int n_frames,
                                                      int f, int passes, uint32_t* lds /*[8]*/) {
  if (s.mode != 2) return s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int p = 0; p < passes; ++p) {
    const uint32_t c = t < 256 ? hist[((size_t)p * n_frames + f) * 256 + (255 - t)] : 0u;  // thread t holds digit 255 - t
    uint32_t inc = c;  // inclusive prefix over descending digits = keys with digit >= 255 - t
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t u = __shfl_up(inc, d, 64);
      if (lane >= d) inc += u;
    }
    if (lane == 63 && wave < 4) lds[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave && w < 4; ++w) base += lds[w];
    inc += base;
    const uint32_t above = inc - c;  // keys with a larger digit
    const int digit = 255 - t;
    // the walk stops at the first digit (from the top) whose cumulative count reaches k_rem, but not below digit 1
    const bool hit = t < 256 && digit >= 1 && above < s.k_rem && s.k_rem <= inc;
    if (hit) { lds[4] = (uint32_t)digit; lds[5] = above; }
    if (t == 255) { lds[6] = above; }  // digit 0: everything above it
    __syncthreads();
    // exactly one digit >= 1 hits unless the candidates above digit 0 are fewer than k_rem
    const bool any = lds[6] >= s.k_rem;  // the cumulative count reaches k_rem at some digit >= 1
    const uint32_t dsel = any ? lds[4] : 0u, rem_sub = any ? lds[5] : lds[6];
    s.prefix |= dsel << (24 - 8 * p);
    s.k_rem -= rem_sub;
    __syncthreads();
  }
  return s;
}

// grid (gx, F): histogram of the current digit over keys of frame f matching the prefix.  Pass 0 reads the logits
// and leaves their order-preserving integer images in `keys` for the later passes.
// (Closing a pass in the same launch — the last block of a frame to take a ticket walks the histogram — was
// built and dropped: the device-scope fence each block needs before its ticket writes back its XCD's L2, 13 MB of
// freshly written keys in pass 0: 64-97 us per pass instead of 5-23.)
// Blocks of TKH_THREADS threads: the 256 counters of a block are merged into the frame's histogram with one global
// atomic each, and same-address atomics serialise — with 1024 blocks of 256 threads that merge was 9 (top byte: a few
// hot digits) and 18 us (second byte) of the 21 / 24 us of the first two passes over 3.26M keys; a quarter of the
// blocks, a quarter of the merges.
constexpr int TKH_THREADS = 1024;
__global__ __launch_bounds__(TKH_THREADS) void k_topk_hist(const float* __restrict__ logits, uint32_t* __restrict__ keys,
                                                   const TopkIn in, int pass,
                                                   uint32_t* __restrict__ hist, int n_frames) {
  __shared__ uint32_t lh[256];
  __shared__ uint32_t st_lds[8];
  const int f = blockIdx.y;
  const int64_t lo = tk_off(in, f), hi = tk_off(in, f + 1);
  const TopkState s0 = tk_state(in, f);
  if (s0.mode != 2) return;  // nothing to select in this frame: its keys are never read (block-uniform)
  const TopkState s = topk_state_after(s0, hist, n_frames, f, pass, st_lds);
  if (threadIdx.x < 256) lh[threadIdx.x] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const uint32_t himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
  // four independent loads in flight per thread: with one, a thread's dozen iterations each paid a full memory
  // latency in front of their LDS atomic (23-26 us for 3.26M keys, a 13-MB read)
  const int64_t stride = (int64_t)gridDim.x * TKH_THREADS;
  for (int64_t r0 = lo + (int64_t)blockIdx.x * TKH_THREADS + threadIdx.x; r0 < hi; r0 += 4 * stride) {
    uint32_t k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = r0 + u * stride;
      k[u] = 0u;
      if (r < hi) k[u] = pass == 0 ? ordered_key(logits[r]) : keys[r];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = r0 + u * stride;
      if (r < hi) {
        if (pass == 0) keys[r] = k[u];
        if (((k[u] ^ s.prefix) & himask) == 0u) atomicAdd(&lh[(k[u] >> shift) & 255u], 1u);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const uint32_t c = lh[threadIdx.x];
    if (c) atomicAdd(&hist[((size_t)pass * n_frames + f) * 256 + threadIdx.x], c);
  }
}

// keep = every key above the threshold + the first k_rem keys equal to it in row order.  One flag array of 2n
// words, [0,n) = "above", [n,2n) = "equal": ONE exclusive scan of it ranks both (the second half continues the
// first, and only differences within a frame are used).
__global__ __launch_bounds__(256) void k_topk_flags(const uint32_t* __restrict__ keys, const TopkIn in,
                                                    const uint32_t* __restrict__ hist, int n_frames, int64_t n,
                                                    uint32_t* __restrict__ fl) {
  __shared__ uint32_t st_lds[8];
  const int f = blockIdx.y;
  const TopkState s = topk_state_after(tk_state(in, f), hist, n_frames, f, 4, st_lds);
  const int64_t lo = tk_off(in, f), hi = tk_off(in, f + 1);
  for (int64_t r = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; r < hi; r += (int64_t)gridDim.x * 256) {
    uint32_t gt = 0u, eq = 0u;
    if (s.mode == 1) {
      gt = 1u;
    } else if (s.mode == 2) {
      const uint32_t k = keys[r];
      gt = k > s.prefix ? 1u : 0u;
      eq = k == s.prefix ? 1u : 0u;
    }
    fl[r] = gt;
    fl[n + r] = eq;
  }
}

__global__ __launch_bounds__(256) void k_topk_emit(const uint32_t* __restrict__ fl, const uint32_t* __restrict__ ex,
                                                   const TopkIn in,
                                                   const uint32_t* __restrict__ hist, int n_frames, int64_t n,
                                                   uint32_t* __restrict__ rows, int32_t* __restrict__ remap,
                                                   uint32_t* __restrict__ next_hist) {
  __shared__ uint32_t st_lds[8];
  tk_clear_next(next_hist);
  const int f = blockIdx.y;
  const TopkState s = topk_state_after(tk_state(in, f), hist, n_frames, f, 4, st_lds);
  const int64_t lo = tk_off(in, f), hi = tk_off(in, f + 1);
  if (lo >= hi) return;
  const uint32_t gt_base = ex[lo], eq_base = ex[n + lo];
  for (int64_t r = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; r < hi; r += (int64_t)gridDim.x * 256) {
    const uint32_t gt_before = ex[r] - gt_base, eq_before = ex[n + r] - eq_base;
    const bool keep = fl[r] || (fl[n + r] && eq_before < s.k_rem);
    const uint32_t at = s.keep_base + gt_before + min(eq_before, s.k_rem);
    if (keep) rows[at] = (uint32_t)r;
    if (remap) remap[r] = keep ? (int32_t)at : -1;
  }
}

// The same in two launches without flag / offset arrays, for frames of up to 4M candidates (2048 tiles of 2048):
// k_topk_tile_counts counts, per frame and tile, the keys above and equal to the threshold; k_topk_place recomputes
// them, adds up the counts of the frame's tiles in front of its own, scans its tile and writes the kept rows.
#define TK_ITEMS 8
#define TK_TILE (256 * TK_ITEMS)

__device__ __forceinline__ uint2 tk_block_excl2(uint2 v, uint2* total, uint32_t* lds /*[8]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint2 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t a = __shfl_up(inc.x, d, 64), b = __shfl_up(inc.y, d, 64);
    if (lane >= d) { inc.x += a; inc.y += b; }
  }
  if (lane == 63) { lds[2 * wave] = inc.x; lds[2 * wave + 1] = inc.y; }
  __syncthreads();
  uint2 base = make_uint2(0u, 0u), tot = make_uint2(0u, 0u);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t a = lds[2 * w], b = lds[2 * w + 1];
    if (w < wave) { base.x += a; base.y += b; }
    tot.x += a; tot.y += b;
  }
  __syncthreads();
  *total = tot;
  return make_uint2(base.x + inc.x - v.x, base.y + inc.y - v.y);
}

// grid (tiles, F); sums [F][tiles][2]
__global__ __launch_bounds__(256) void k_topk_tile_counts(const uint32_t* __restrict__ keys, const TopkIn in,
                                                          const uint32_t* __restrict__ hist, int n_frames,
                                                          uint32_t* __restrict__ sums) {
  __shared__ uint32_t st_lds[8];
  const int f = blockIdx.y;
  const int64_t lo = tk_off(in, f), hi = tk_off(in, f + 1);
  const int64_t base = lo + (int64_t)blockIdx.x * TK_TILE + (int64_t)threadIdx.x * TK_ITEMS;
  const TopkState s = topk_state_after(tk_state(in, f), hist, n_frames, f, 4, st_lds);
  uint2 c = make_uint2(0u, 0u);
  if (s.mode == 2) {
#pragma unroll
    for (int j = 0; j < TK_ITEMS; ++j)
      if (base + j < hi) {
        const uint32_t k = keys[base + j];
        c.x += k > s.prefix ? 1u : 0u;
        c.y += k == s.prefix ? 1u : 0u;
      }
  } else if (s.mode == 1) {
#pragma unroll
    for (int j = 0; j < TK_ITEMS; ++j) c.x += base + j < hi ? 1u : 0u;
  }
  uint2 tot;
  tk_block_excl2(c, &tot, st_lds);
  if (threadIdx.x == 0) {
    sums[((size_t)f * gridDim.x + blockIdx.x) * 2] = tot.x;
    sums[((size_t)f * gridDim.x + blockIdx.x) * 2 + 1] = tot.y;
  }
}

__global__ __launch_bounds__(256) void k_topk_place(const uint32_t* __restrict__ keys, const TopkIn in,
                                                    const uint32_t* __restrict__ hist, int n_frames,
                                                    const uint32_t* __restrict__ sums, uint32_t* __restrict__ rows,
                                                    int32_t* __restrict__ remap, uint32_t* __restrict__ next_hist) {
  __shared__ uint32_t st_lds[8];
  tk_clear_next(next_hist);
  const int f = blockIdx.y;
  const int64_t lo = tk_off(in, f), hi = tk_off(in, f + 1);
  if (lo + (int64_t)blockIdx.x * TK_TILE >= hi) return;   // block-uniform
  const int64_t base = lo + (int64_t)blockIdx.x * TK_TILE + (int64_t)threadIdx.x * TK_ITEMS;
  const TopkState s = topk_state_after(tk_state(in, f), hist, n_frames, f, 4, st_lds);
  if (s.mode == 0) {   // block-uniform
    if (remap) {
#pragma unroll
      for (int j = 0; j < TK_ITEMS; ++j)
        if (base + j < hi) remap[base + j] = -1;
    }
    return;
  }
  uint2 part = make_uint2(0u, 0u);
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) {
    part.x += sums[((size_t)f * gridDim.x + t) * 2];
    part.y += sums[((size_t)f * gridDim.x + t) * 2 + 1];
  }
  uint2 tile_off;
  tk_block_excl2(part, &tile_off, st_lds);
  uint32_t gt[TK_ITEMS], eq[TK_ITEMS];
  uint2 c = make_uint2(0u, 0u);
#pragma unroll
  for (int j = 0; j < TK_ITEMS; ++j) {
    gt[j] = 0u;
    eq[j] = 0u;
    if (base + j < hi) {
      if (s.mode == 1) gt[j] = 1u;
      else {
        const uint32_t k = keys[base + j];
        gt[j] = k > s.prefix ? 1u : 0u;
        eq[j] = k == s.prefix ? 1u : 0u;
      }
    }
    c.x += gt[j];
    c.y += eq[j];
  }
  uint2 tot;
  uint2 ex = tk_block_excl2(c, &tot, st_lds);
  ex.x += tile_off.x;
  ex.y += tile_off.y;
#pragma unroll
  for (int j = 0; j < TK_ITEMS; ++j) {
    const bool keep = gt[j] || (eq[j] && ex.y < s.k_rem);
    const uint32_t at = s.keep_base + ex.x + min(ex.y, s.k_rem);
    if (keep) rows[at] = (uint32_t)(base + j);
    if (remap && base + j < hi) remap[base + j] = keep ? (int32_t)at : -1;
    ex.x += gt[j];
    ex.y += eq[j];
  }
}

extern "C" int pcc_topk_prune(pcc_ctx* ctx, const float* d_logits, int64_t n, int n_batch,
                              const int64_t* h_offsets, const int64_t* h_k, uint32_t* d_keep_rows,
                              int64_t* h_n_keep) {
  return pcc_topk_prune_map(ctx, d_logits, n, n_batch, h_offsets, h_k, d_keep_rows, h_n_keep, nullptr);
}

extern "C" int pcc_topk_prune_map(pcc_ctx* ctx, const float* d_logits, int64_t n, int n_batch,
                                  const int64_t* h_offsets, const int64_t* h_k, uint32_t* d_keep_rows,
                                  int64_t* h_n_keep, int32_t* d_remap) {
  PCC_REQUIRE(ctx && h_offsets && h_k, PCC_E_ARG, "pcc_topk_prune: null arg");
  PCC_REQUIRE(n_batch >= 1 && n_batch <= 120, PCC_E_ARG, "pcc_topk_prune: n_batch=%d", n_batch);
  if (h_n_keep) *h_n_keep = 0;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_logits && d_keep_rows, PCC_E_ARG, "pcc_topk_prune: null buffers");
  PCC_REQUIRE(n < ((int64_t)1 << 31), PCC_E_ARG, "pcc_topk_prune: n too large");
  PCC_REQUIRE(h_offsets[0] == 0 && h_offsets[n_batch] == n, PCC_E_ARG,
              "pcc_topk_prune: offsets must span [0,n]");
  hipStream_t st = ctx->stream;
  const size_t nb4 = pcc_align((size_t)n * 4);
  const bool by_arg = n_batch <= TOPK_ARG_FRAMES;
  PCC_TRY(pcc_arena_reserve(ctx, 5 * nb4 + pcc_scan_scratch_bytes(2 * n) + 4 * 256 * 4 * (size_t)n_batch + 8192 +
                                     (size_t)n_batch * 64));
  uint32_t* keys = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 4);
  uint32_t* fl = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 8);    // "above" flags, then "equal" flags
  uint32_t* ex = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 8);    // their exclusive scan
  uint32_t* hist = nullptr;                                          // [pass][frame][256]
  uint32_t* next_hist = nullptr;
  int64_t* offs = nullptr;
  TopkState* state = nullptr;
  if (!keys || !fl || !ex) return PCC_E_NOMEM;
  if (by_arg) {
    if (!ctx->topk_hist[0]) {   // both buffers once per context, cleared here; from then on by the calls themselves
      uint32_t* two = nullptr;
      PCC_HIP(hipMalloc((void**)&two, (size_t)2 * kTopkHistWords * 4));
      if (hipMemsetAsync(two, 0, (size_t)2 * kTopkHistWords * 4, st) != hipSuccess) {
        (void)hipFree(two);
        pcc_set_error("pcc_topk_prune: clearing the histograms failed");
        return PCC_E_HIP;
      }
      ctx->topk_hist[0] = two;
      ctx->topk_hist[1] = two + kTopkHistWords;
      ctx->topk_clean[0] = ctx->topk_clean[1] = true;
      ctx->topk_next = 0;
    }
    const int cur = ctx->topk_next;
    hist = ctx->topk_hist[cur];
    next_hist = ctx->topk_hist[cur ^ 1];
    if (!ctx->topk_clean[cur]) {   // a call that failed half-way left it in an unknown state
      PCC_HIP(hipMemsetAsync(hist, 0, (size_t)kTopkHistWords * 4, st));
      ctx->topk_clean[cur] = true;
    }
  } else {
    hist = (uint32_t*)pcc_arena_alloc(ctx, (size_t)4 * 256 * 4 * n_batch);
    offs = (int64_t*)pcc_arena_alloc(ctx, (size_t)(n_batch + 1) * 8);
    state = (TopkState*)pcc_arena_alloc(ctx, sizeof(TopkState) * n_batch);
    if (!hist || !offs || !state) return PCC_E_NOMEM;
  }
  PccProfScope prof(ctx, "topk_prune", n, n_batch, 0, 0);

  // per-frame parameters: kernel arguments for small GOPs, else staged through the pinned buffer
  TopkIn in;
  memset(&in, 0, sizeof(in));
  in.offs = offs;
  in.state = state;
  in.by_arg = by_arg ? 1 : 0;
  char* hp = (char*)ctx->pinned;
  int64_t max_cnt = 0, kept = 0;
  if (by_arg) memcpy(in.a.offs, h_offsets, (size_t)(n_batch + 1) * 8);
  else memcpy(hp, h_offsets, (size_t)(n_batch + 1) * 8);
  TopkState* hs = by_arg ? in.a.st : (TopkState*)(hp + (size_t)(n_batch + 1) * 8);
  for (int f = 0; f < n_batch; ++f) {
    const int64_t cnt = h_offsets[f + 1] - h_offsets[f];
    PCC_REQUIRE(cnt >= 0 && h_k[f] >= 0, PCC_E_ARG, "pcc_topk_prune: negative count/k in frame %d", f);
    if (cnt > max_cnt) max_cnt = cnt;
    hs[f].prefix = 0;
    hs[f].keep_base = (uint32_t)kept;
    if (h_k[f] == 0 || cnt == 0) { hs[f].mode = 0; hs[f].k_rem = 0; }
    else if (h_k[f] >= cnt) { hs[f].mode = 1; hs[f].k_rem = 0; }
    else { hs[f].mode = 2; hs[f].k_rem = (uint32_t)h_k[f]; }
    kept += h_k[f] < cnt ? h_k[f] : cnt;
  }
  if (by_arg) {
    ctx->topk_clean[ctx->topk_next] = false;   // until the last launch of this call is queued (it clears the other one)
  } else {
    PCC_HIP(hipMemcpyAsync(offs, hp, (size_t)(n_batch + 1) * 8, hipMemcpyHostToDevice, st));
    PCC_HIP(hipMemcpyAsync(state, hs, sizeof(TopkState) * n_batch, hipMemcpyHostToDevice, st));
    PCC_HIP(hipMemsetAsync(hist, 0, (size_t)4 * 256 * 4 * n_batch, st));
  }
  // the pinned buffer is reused below only after the final synchronise
  auto call_done = [&]() {   // the call's last launch is queued: the other buffer is clear for the next call
    if (by_arg) {
      ctx->topk_clean[ctx->topk_next ^ 1] = true;
      ctx->topk_next ^= 1;
    }
  };

  unsigned gx = nblk(max_cnt, 256 * 8);
  if (gx < 1) gx = 1;
  if (gx > 1024) gx = 1024;
  const dim3 grid2(gx, (unsigned)n_batch);
  unsigned gxh = nblk(max_cnt, TKH_THREADS * 8);
  if (gxh < 1) gxh = 1;
  if (gxh > 256) gxh = 256;
  const dim3 gridh(gxh, (unsigned)n_batch);
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(k_topk_hist, gridh, dim3(TKH_THREADS), 0, st, d_logits, keys, in, pass, hist, n_batch);
    PCC_CHECK_LAUNCH();
  }
  const unsigned tiles = std::max(1u, nblk(max_cnt, TK_TILE));
  if (max_cnt <= (int64_t)TK_TILE * TK_TILE && (size_t)n_batch * tiles * 2 <= (size_t)n * 2) {
    // two launches: counts per tile, then scan + placement (no flag / offset arrays)
    uint32_t* sums = fl;   // [F][tiles][2], in the flag array of the four-launch form
    const dim3 gridt(tiles, (unsigned)n_batch);
    hipLaunchKernelGGL(k_topk_tile_counts, gridt, dim3(256), 0, st, (const uint32_t*)keys, in, (const uint32_t*)hist, n_batch,
                       sums);
    PCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_topk_place, gridt, dim3(256), 0, st, (const uint32_t*)keys, in, (const uint32_t*)hist, n_batch,
                       (const uint32_t*)sums, d_keep_rows, d_remap, next_hist);
    PCC_CHECK_LAUNCH();
    call_done();
    if (h_n_keep) *h_n_keep = kept;
    if (!by_arg) PCC_HIP(hipStreamSynchronize(st));
    return PCC_OK;
  }
  hipLaunchKernelGGL(k_topk_flags, grid2, dim3(256), 0, st, (const uint32_t*)keys, in, (const uint32_t*)hist, n_batch, n, fl);
  PCC_CHECK_LAUNCH();
  PCC_TRY(pcc_scan_exclusive_u32(ctx, fl, ex, 2 * n, nullptr));
  hipLaunchKernelGGL(k_topk_emit, grid2, dim3(256), 0, st, (const uint32_t*)fl, (const uint32_t*)ex, in,
                     (const uint32_t*)hist, n_batch, n, d_keep_rows, d_remap, next_hist);
  PCC_CHECK_LAUNCH();
  call_done();
  // the number of kept rows is sum_f min(k_f, cnt_f) by construction (exact top-k, ties broken by row): no read-back
  if (h_n_keep) *h_n_keep = kept;
  if (!by_arg) PCC_HIP(hipStreamSynchronize(st));  // staged parameters consumed; pinned buffer free again
  return PCC_OK;
}
