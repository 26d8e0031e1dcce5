// common.h — shared host/device helpers for libpcc_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/pcc.h"

#define PCC_WAVE 64

void pcc_set_error(const char* fmt, ...);

#define PCC_HIP(call)                                                        \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      pcc_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call,             \
                    hipGetErrorString(e_));                                  \
      return PCC_E_HIP;                                                      \
    }                                                                        \
  } while (0)

#define PCC_CHECK_LAUNCH() PCC_HIP(hipGetLastError())

#define PCC_REQUIRE(cond, code, ...)                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      pcc_set_error(__VA_ARGS__);                                            \
      return (code);                                                         \
    }                                                                        \
  } while (0)

#define PCC_TRY(expr)                                                        \
  do {                                                                       \
    int r_ = (expr);                                                         \
    if (r_ != PCC_OK) return r_;                                             \
  } while (0)

struct pcc_ctx {
  int device;
  hipStream_t stream;
  // bump arena for per-call scratch
  char* arena;
  size_t arena_cap;
  size_t arena_off;
  // small pinned host staging for count read-backs
  void* pinned;
  size_t pinned_cap;
  hipEvent_t ev0, ev1;
  bool ev_valid;
  // per-launch profiler (bench.py roofline figure): event pairs around the
  // kernels of an API call, read back after a synchronise
  bool prof_on;
  char prof_only[32];  // "" = every entry point, else only ops whose name starts with this
  int64_t prof_only_d0;  // and, when >= 0, whose first dimension (rows) is this
  int prof_n, prof_cap;
  struct pcc_prof_rec* prof;
  // weights in MFMA operand order, by device pointer (pcc_conv_prepare, conv.hip)
  struct PccWeightCache* wcache;
  // pinned staging grown on demand (octree2.hip: blobs and decoded points on their way over PCIe)
  void* stage;
  size_t stage_cap;
  // topk.hip: the histograms of GOPs of up to 8 frames, two buffers used in turn (one hipMalloc; each call's last
  // launch clears the other buffer for the next call)
  uint32_t* topk_hist[2];
  bool topk_clean[2];
  int topk_next;
};

struct pcc_prof_rec {
  hipEvent_t e0, e1;
  const char* op;
  int64_t dims[4];
};

// RAII event pair around the launches of one C-ABI call (no-op unless enabled)
struct PccProfScope {
  pcc_ctx* c;
  int slot;
  PccProfScope(pcc_ctx* ctx, const char* op, int64_t d0, int64_t d1, int64_t d2, int64_t d3);
  ~PccProfScope();
};

#include "rans_gate.h"

// True when pcc_sparse_conv_head_up has a kernel to run (always, except under PCC_FORCE_SCALAR=1); the whole-GOP
// decoder otherwise materialises the child rule books.
bool pcc_conv_up_fused();
// conv.hip: frees the operand-ordered weight copies of a context (pcc_destroy)
void pcc_wcache_free(pcc_ctx* ctx);
// conv.hip: pcc_sparse_conv_head_up on input rows whose 32 channels are stored in the order kConv16Perm below
// sort.hip: pcc_morton_keys that also flags batch indexes >= n_batch
int pcc_morton_keys_batch(pcc_ctx* ctx, const int32_t* d_coords, int64_t n, int n_batch, uint64_t* d_keys, int32_t* d_flag);
// sort.hip: pcc_sort_pairs on given key bytes only (no look at the keys, no host round trip)
int pcc_sort_pairs_bytes(pcc_ctx* ctx, uint64_t* d_keys, uint32_t* d_perm, int64_t n, unsigned byte_mask);
// sort.hip: canonical order + rows in that order of a small coordinate set given by its Morton keys
int64_t pcc_sort_small_max();
int pcc_sort_keys_canonical(pcc_ctx* ctx, const uint64_t* d_mkeys, int64_t n, uint32_t* d_perm, int32_t* d_sorted_coords);
// sort.hip: dst[perm[i]] = src[i] for a permutation perm (pcc_inverse_rows + pcc_gather_rows in one launch)
int pcc_scatter_rows(pcc_ctx* ctx, const void* d_src, const uint32_t* d_perm, int64_t n, int row_bytes, void* d_dst);
// map.hip: rule-book columns of a latent's voxels among the 64 generated descendants of their stride-32 ancestors
int pcc_descendant_map(pcc_ctx* ctx, const int32_t* d_nbr_parent, int64_t parent_pitch, const uint32_t* d_perm,
                       const uint64_t* d_ykeys, const int32_t* d_parent_of8, const int32_t* d_parent_of16, int64_t m,
                       int32_t* d_nbr);
// octree.hip: the single-workgroup octree kernel without any read-back (codec.hip's geometry slot)
int pcc_octree_small_max();
int pcc_octree_small_async(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, int depth, uint8_t* d_occ,
                           int64_t cap_s, uint32_t* d_counts);
// parts of a blob of version 3 for n leaves: min(8, n / 4096), at least 2 (the rule oracle/pcc_oracle.c states)
static inline int pcc_octree_parts_for(int64_t n) {
  const int64_t k = n / 4096;
  return (int)(k > 8 ? 8 : (k < 2 ? 2 : k));
}
// octree.hip: the same for blob version 3 — K parts of the frame's leaves under the frame's root, one workgroup each
int pcc_octree_parts_async(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, int depth, int K, uint8_t* d_occ,
                           int64_t cap_s, uint32_t* d_counts, int counts_stride);
int pcc_sparse_conv_head_up_perm_rgb(pcc_ctx* ctx, const float* d_in, int64_t n_parents, const int32_t* d_nbr_parent,
                                     int64_t parent_pitch, const float* d_w, const float* d_bias, int relu,
                                     const float* d_head_w, const float* d_head_b, float* d_head_out,
                                     const float* d_rgb_w, const float* d_rgb_b, float* d_rgb_out);
int pcc_sparse_conv_head_up_perm(pcc_ctx* ctx, const float* d_in, int64_t n_parents, const int32_t* d_nbr_parent,
                                 int64_t parent_pitch, const float* d_w, const float* d_bias, int relu, float* d_out,
                                 const float* d_head_w, const float* d_head_b, float* d_head_out);
// storage position j of a permuted row holds channel pcc_conv16_perm(j) = 4 (j & 7) + (j >> 3): the eight MFMA B
// operands of lane (slot, q) of k_gconv16 are then the 32 contiguous bytes at 8 q
static inline int pcc_conv16_perm(int j) { return 4 * (j & 7) + (j >> 3); }

// Reset the arena at the start of an API call.
static inline void pcc_arena_reset(pcc_ctx* c) { c->arena_off = 0; }
// Make sure the arena can hold `bytes` in total for this call; may
// synchronise + reallocate (only ever called before the first arena_alloc of
// an API call, so no live scratch is lost).
int pcc_arena_reserve(pcc_ctx* c, size_t bytes);
// Bump-allocate (256-B aligned).  Returns nullptr if the reservation was too
// small (a library bug, reported as PCC_E_NOMEM by callers).
void* pcc_arena_alloc(pcc_ctx* c, size_t bytes);

static inline size_t pcc_align(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- Morton helpers (host + device) ---------------------------------------
// spread the low 16 bits of v so that bit i lands at bit 3i
__host__ __device__ static inline uint64_t pcc_spread3(uint32_t v) {
  uint64_t x = v & 0xFFFFu;
  x = (x | (x << 16)) & 0x0000FF0000FFull;
  x = (x | (x << 8)) & 0x00F00F00F00Full;
  x = (x | (x << 4)) & 0x0C30C30C30C3ull;
  x = (x | (x << 2)) & 0x249249249249ull;
  return x;
}
__host__ __device__ static inline uint32_t pcc_compact3(uint64_t x) {
  x &= 0x249249249249ull;
  x = (x | (x >> 2)) & 0x0C30C30C30C3ull;
  x = (x | (x >> 4)) & 0x00F00F00F00Full;
  x = (x | (x >> 8)) & 0x0000FF0000FFull;
  x = (x | (x >> 16)) & 0xFFFFull;
  return (uint32_t)x;
}
// key = b<<48 | x bits at 3i+2, y at 3i+1, z at 3i  (biased by 32768)
__host__ __device__ static inline uint64_t pcc_morton(int b, int x, int y,
                                                      int z) {
  return ((uint64_t)(uint32_t)b << 48) |
         (pcc_spread3((uint32_t)(x + 32768)) << 2) |
         (pcc_spread3((uint32_t)(y + 32768)) << 1) |
         pcc_spread3((uint32_t)(z + 32768));
}
__host__ __device__ static inline void pcc_unmorton(uint64_t key, int* b,
                                                    int* x, int* y, int* z) {
  *b = (int)(key >> 48);
  *x = (int)pcc_compact3(key >> 2) - 32768;
  *y = (int)pcc_compact3(key >> 1) - 32768;
  *z = (int)pcc_compact3(key) - 32768;
}

static inline int pcc_ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// internal cross-file entry points
int pcc_scan_exclusive_u32(pcc_ctx* ctx, const uint32_t* d_in, uint32_t* d_out,
                           int64_t n, uint32_t* d_total /*nullable, device*/);
size_t pcc_scan_scratch_bytes(int64_t n);
