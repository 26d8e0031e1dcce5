// topk.hip — per-frame top-k occupancy pruning.
//
// Replaces the MinkowskiPruning + per-batch torch.topk inside
// model.g_s(y_hat, k=ks) (codec_parallel.py:469): of the candidate voxels of
// frame b keep the k[b] with the largest occupancy logit.  Exact and
// order-independent: the threshold is found by an MSB-first radix select on
// the order-preserving integer image of the logit (4 histogram passes), ties at
// the threshold are resolved by row order (lower Morton key first); one prefix
// scan over the "above" and "equal" flags ranks both and places the survivors
// (stable compaction).  14 launches per call (was 20): at the sizes of this path
// a call is launch-bound, not bandwidth-bound.
// No sort of the 2M candidates, no floating-point comparisons that could
// differ between encoder, decoder and the CPU oracle.
#include "common.h"
#include <string.h>

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

struct TopkState {
  uint32_t prefix;     // threshold key bits fixed so far
  uint32_t k_rem;      // how many still to take among keys matching the prefix
  int32_t mode;        // 0 keep none, 1 keep all, 2 select
  uint32_t keep_base;  // rows kept in the frames before this one (= sum of min(k, count): known on the host)
};

__device__ __forceinline__ uint32_t ordered_key(float v) {
  uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// grid (gx, F): histogram of the current digit over keys of frame f matching the prefix.  Pass 0 reads the logits
// and leaves their order-preserving integer images in `keys` for the later passes.
// (Closing a pass in the same launch — the last block of a frame to take a ticket walks the histogram — was
// built and dropped: the device-scope fence each block needs before its ticket writes back its XCD's L2, 13 MB of
// freshly written keys in pass 0: 64-97 us per pass instead of 5-23.)
__global__ __launch_bounds__(256) void k_topk_hist(const float* __restrict__ logits, uint32_t* __restrict__ keys,
                                                   const int64_t* __restrict__ offs,
                                                   const TopkState* __restrict__ state, int pass,
                                                   uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[256];
  const int f = blockIdx.y;
  const TopkState s = state[f];
  const int64_t lo = offs[f], hi = offs[f + 1];
  if (s.mode != 2) return;  // nothing to select in this frame: its keys are never read
  lh[threadIdx.x] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const uint32_t himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
  for (int64_t r = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; r < hi; r += (int64_t)gridDim.x * 256) {
    uint32_t k;
    if (pass == 0) {
      k = ordered_key(logits[r]);
      keys[r] = k;
    } else {
      k = keys[r];
    }
    if (((k ^ s.prefix) & himask) == 0u) atomicAdd(&lh[(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  const uint32_t c = lh[threadIdx.x];
  if (c) atomicAdd(&hist[f * 256 + threadIdx.x], c);
}

// one block per frame: walk the histogram from the top digit down, then clear it for the next pass
__global__ void k_topk_update(uint32_t* __restrict__ hist, TopkState* __restrict__ state, int pass) {
  const int f = blockIdx.x;
  if (threadIdx.x == 0) {
    TopkState s = state[f];
    if (s.mode == 2) {
      const int shift = 24 - 8 * pass;
      uint32_t rem = s.k_rem;
      int d = 255;
      for (; d > 0; --d) {
        const uint32_t c = hist[f * 256 + d];
        if (c >= rem) break;
        rem -= c;
      }
      s.prefix |= (uint32_t)d << shift;
      s.k_rem = rem;
      state[f] = s;
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < 256; d += blockDim.x) hist[f * 256 + d] = 0;
}

// keep = every key above the threshold + the first k_rem keys equal to it in row order.  One flag array of 2n
// words, [0,n) = "above", [n,2n) = "equal": ONE exclusive scan of it ranks both (the second half continues the
// first, and only differences within a frame are used).
__global__ __launch_bounds__(256) void k_topk_flags(const uint32_t* __restrict__ keys,
                                                    const int64_t* __restrict__ offs,
                                                    const TopkState* __restrict__ state, int64_t n,
                                                    uint32_t* __restrict__ fl) {
  const int f = blockIdx.y;
  const TopkState s = state[f];
  const int64_t lo = offs[f], hi = offs[f + 1];
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
                                                   const int64_t* __restrict__ offs,
                                                   const TopkState* __restrict__ state, int64_t n,
                                                   uint32_t* __restrict__ rows) {
  const int f = blockIdx.y;
  const TopkState s = state[f];
  const int64_t lo = offs[f], hi = offs[f + 1];
  if (lo >= hi) return;
  const uint32_t gt_base = ex[lo], eq_base = ex[n + lo];
  for (int64_t r = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; r < hi; r += (int64_t)gridDim.x * 256) {
    const uint32_t gt_before = ex[r] - gt_base, eq_before = ex[n + r] - eq_base;
    const bool keep = fl[r] || (fl[n + r] && eq_before < s.k_rem);
    if (keep) rows[s.keep_base + gt_before + min(eq_before, s.k_rem)] = (uint32_t)r;
  }
}

// per-frame parameters of a small GOP travel as kernel arguments: no pinned staging, hence no
// stream synchronisation on their account
#define TOPK_ARG_FRAMES 8
struct TopkArgs {
  int64_t offs[TOPK_ARG_FRAMES + 1];
  TopkState st[TOPK_ARG_FRAMES];
};
// also clears the histograms
__global__ __launch_bounds__(256) void k_topk_params(TopkArgs a, int n_batch, int64_t* __restrict__ offs,
                                                     TopkState* __restrict__ state, uint32_t* __restrict__ hist) {
  const int t = threadIdx.x;
  if (t <= n_batch) offs[t] = a.offs[t];
  if (t < n_batch) state[t] = a.st[t];
  for (int j = t; j < 256 * n_batch; j += 256) hist[j] = 0u;
}

extern "C" int pcc_topk_prune(pcc_ctx* ctx, const float* d_logits, int64_t n, int n_batch,
                              const int64_t* h_offsets, const int64_t* h_k, uint32_t* d_keep_rows,
                              int64_t* h_n_keep) {
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
  PCC_TRY(pcc_arena_reserve(ctx, 5 * nb4 + pcc_scan_scratch_bytes(2 * n) + 256 * 4 * (size_t)n_batch + 8192 +
                                     (size_t)n_batch * 64));
  uint32_t* keys = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 4);
  uint32_t* fl = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 8);    // "above" flags, then "equal" flags
  uint32_t* ex = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 8);    // their exclusive scan
  uint32_t* hist = (uint32_t*)pcc_arena_alloc(ctx, (size_t)256 * 4 * n_batch);
  int64_t* offs = (int64_t*)pcc_arena_alloc(ctx, (size_t)(n_batch + 1) * 8);
  TopkState* state = (TopkState*)pcc_arena_alloc(ctx, sizeof(TopkState) * n_batch);
  if (!keys || !fl || !ex || !hist || !offs || !state) return PCC_E_NOMEM;
  PccProfScope prof(ctx, "topk_prune", n, n_batch, 0, 0);

  // per-frame parameters: kernel arguments for small GOPs, else staged through the pinned buffer
  const bool by_arg = n_batch <= TOPK_ARG_FRAMES;
  TopkArgs args;
  char* hp = (char*)ctx->pinned;
  int64_t max_cnt = 0, kept = 0;
  if (by_arg) memcpy(args.offs, h_offsets, (size_t)(n_batch + 1) * 8);
  else memcpy(hp, h_offsets, (size_t)(n_batch + 1) * 8);
  TopkState* hs = by_arg ? args.st : (TopkState*)(hp + (size_t)(n_batch + 1) * 8);
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
    hipLaunchKernelGGL(k_topk_params, dim3(1), dim3(256), 0, st, args, n_batch, offs, state, hist);
    PCC_CHECK_LAUNCH();
  } else {
    PCC_HIP(hipMemcpyAsync(offs, hp, (size_t)(n_batch + 1) * 8, hipMemcpyHostToDevice, st));
    PCC_HIP(hipMemcpyAsync(state, hs, sizeof(TopkState) * n_batch, hipMemcpyHostToDevice, st));
    PCC_HIP(hipMemsetAsync(hist, 0, (size_t)256 * 4 * n_batch, st));
  }
  // the pinned buffer is reused below only after the final synchronise

  unsigned gx = nblk(max_cnt, 256 * 8);
  if (gx < 1) gx = 1;
  if (gx > 1024) gx = 1024;
  const dim3 grid2(gx, (unsigned)n_batch);
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(k_topk_hist, grid2, dim3(256), 0, st, d_logits, keys, (const int64_t*)offs,
                       (const TopkState*)state, pass, hist);
    PCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_topk_update, dim3((unsigned)n_batch), dim3(64), 0, st, hist, state, pass);
    PCC_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_topk_flags, grid2, dim3(256), 0, st, (const uint32_t*)keys, (const int64_t*)offs,
                     (const TopkState*)state, n, fl);
  PCC_CHECK_LAUNCH();
  PCC_TRY(pcc_scan_exclusive_u32(ctx, fl, ex, 2 * n, nullptr));
  hipLaunchKernelGGL(k_topk_emit, grid2, dim3(256), 0, st, (const uint32_t*)fl, (const uint32_t*)ex,
                     (const int64_t*)offs, (const TopkState*)state, n, d_keep_rows);
  PCC_CHECK_LAUNCH();
  // the number of kept rows is sum_f min(k_f, cnt_f) by construction (exact top-k, ties broken by row): no read-back
  if (h_n_keep) *h_n_keep = kept;
  if (!by_arg) PCC_HIP(hipStreamSynchronize(st));  // staged parameters consumed; pinned buffer free again
  return PCC_OK;
}
