// voxelize.hip — the capture pre-step that feeds the codec.
//
// Replaces, on the GPU, what sender/capturer/capturer.py:88-126 does on the host with numpy +
// Open3D for every camera frame (ZED XYZRGBA float32 [M,4], colour packed in the 4th float):
// drop non-finite / far points (norm <= depth_clip), Open3D voxel_down_sample(voxel_size)
// ([RECALL] voxel index = floor((p - (min_bound - voxel_size/2)) / voxel_size) in double, position
// and colour averaged per voxel in double, input order), then round(mean / voxel_size) to the
// integer voxel.  De-duplication and the max_points cap reuse the codec's sort / scan / top-k
// entry points (see capture.py).  All arithmetic is single correctly-rounded IEEE operations so
// that the numpy oracle reproduces it bit for bit.
#include "common.h"
#include <string.h>

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

__device__ __forceinline__ uint32_t f2ord(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// valid[i] = finite && norm <= clip ; ordered-int minimum of the valid coordinates ; count
__global__ __launch_bounds__(256) void k_vox_valid(const float4* __restrict__ pts, int64_t m, float clip,
                                                   uint8_t* __restrict__ valid, uint32_t* __restrict__ mn_ord,
                                                   unsigned long long* __restrict__ count) {
  __shared__ uint32_t s_mn[3];
  __shared__ uint32_t s_cnt;
  if (threadIdx.x < 3) s_mn[threadIdx.x] = 0xFFFFFFFFu;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  uint32_t c = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = pts[i];
    const bool fin = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
    // np.linalg.norm on float32 rows: sqrt((x*x + y*y) + z*z), every step rounded to float32
    const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(p.x, p.x), __fmul_rn(p.y, p.y)), __fmul_rn(p.z, p.z)));
    const bool ok = fin && d <= clip;
    valid[i] = ok ? 1 : 0;
    if (ok) {
      mn[0] = min(mn[0], f2ord(p.x)); mn[1] = min(mn[1], f2ord(p.y)); mn[2] = min(mn[2], f2ord(p.z));
      ++c;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) atomicMin(&s_mn[a], mn[a]);
  atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&mn_ord[threadIdx.x], s_mn[threadIdx.x]);
  if (threadIdx.x == 0) atomicAdd(count, (unsigned long long)s_cnt);
}

// key = ix<<42 | iy<<21 | iz of the Open3D voxel index (all ones for invalid points, so they sort last)
__global__ void k_vox_keys(const float4* __restrict__ pts, const uint8_t* __restrict__ valid, int64_t m,
                           double bx, double by, double bz, double vs, uint64_t* __restrict__ keys,
                           int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  uint64_t k = ~0ull;
  if (valid[i]) {
    const float4 p = pts[i];
    const double ix = floor(__ddiv_rn(__dsub_rn((double)p.x, bx), vs));
    const double iy = floor(__ddiv_rn(__dsub_rn((double)p.y, by), vs));
    const double iz = floor(__ddiv_rn(__dsub_rn((double)p.z, bz), vs));
    if (ix < 0 || iy < 0 || iz < 0 || ix >= 2097152.0 || iy >= 2097152.0 || iz >= 2097152.0) {
      atomicOr(flag, 1);
    } else {
      k = ((uint64_t)ix << 42) | ((uint64_t)iy << 21) | (uint64_t)iz;
    }
  }
  keys[i] = k;
}

__global__ void k_vox_flags(const uint64_t* __restrict__ skeys, int64_t n, uint32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1u : 0u;
}

// one thread per voxel: mean of its points (double, input order), integer voxel = rint(mean / vs)
__global__ void k_vox_mean(const float4* __restrict__ pts, const uint64_t* __restrict__ skeys,
                           const uint32_t* __restrict__ perm, const uint32_t* __restrict__ flags,
                           const uint32_t* __restrict__ excl, int64_t n, double vs,
                           int32_t* __restrict__ out_pts /*[v,4] (0,x,y,z)*/, double* __restrict__ out_col /*[v,3]*/) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flags[i]) return;
  const uint64_t key = skeys[i];
  double sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0;
  int64_t cnt = 0;
  for (int64_t j = i; j < n && skeys[j] == key; ++j) {
    const float4 p = pts[perm[j]];
    const uint32_t rgba = __float_as_uint(p.w);
    sx = __dadd_rn(sx, (double)p.x); sy = __dadd_rn(sy, (double)p.y); sz = __dadd_rn(sz, (double)p.z);
    sr = __dadd_rn(sr, __ddiv_rn((double)(rgba & 0xFFu), 255.0));
    sg = __dadd_rn(sg, __ddiv_rn((double)((rgba >> 8) & 0xFFu), 255.0));
    sb = __dadd_rn(sb, __ddiv_rn((double)((rgba >> 16) & 0xFFu), 255.0));
    ++cnt;
  }
  const double c = (double)cnt;
  const int64_t v = excl[i];
  out_pts[4 * v + 0] = 0;
  out_pts[4 * v + 1] = (int32_t)rint(__ddiv_rn(__ddiv_rn(sx, c), vs));
  out_pts[4 * v + 2] = (int32_t)rint(__ddiv_rn(__ddiv_rn(sy, c), vs));
  out_pts[4 * v + 3] = (int32_t)rint(__ddiv_rn(__ddiv_rn(sz, c), vs));
  out_col[3 * v + 0] = __ddiv_rn(sr, c);
  out_col[3 * v + 1] = __ddiv_rn(sg, c);
  out_col[3 * v + 2] = __ddiv_rn(sb, c);
}

// first-of-run flags over rows already sorted by coordinate (duplicates adjacent)
__global__ void k_row_first_flags(const int4* __restrict__ sorted, int64_t n, uint32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool first = true;
  if (i > 0) {
    const int4 a = sorted[i - 1], b = sorted[i];
    first = (a.x != b.x) | (a.y != b.y) | (a.z != b.z) | (a.w != b.w);
  }
  flags[i] = first ? 1u : 0u;
}

__global__ void k_flagged_rows(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ excl, int64_t n,
                               uint32_t* __restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) rows[excl[i]] = (uint32_t)i;
}

extern "C" int pcc_vox_valid(pcc_ctx* ctx, const float* d_xyzrgba, int64_t m, float depth_clip, uint8_t* d_valid,
                             float* h_min_bound, int64_t* h_n_valid) {
  PCC_REQUIRE(ctx && h_min_bound && h_n_valid, PCC_E_ARG, "pcc_vox_valid: null arg");
  *h_n_valid = 0;
  if (m <= 0) return PCC_OK;
  PCC_REQUIRE(d_xyzrgba && d_valid && ((uintptr_t)d_xyzrgba % 16 == 0), PCC_E_ARG, "pcc_vox_valid: bad buffers");
  PCC_TRY(pcc_arena_reserve(ctx, 512));
  uint32_t* mn = (uint32_t*)pcc_arena_alloc(ctx, 16);
  unsigned long long* cnt = (unsigned long long*)pcc_arena_alloc(ctx, 8);
  if (!mn || !cnt) return PCC_E_NOMEM;
  PCC_HIP(hipMemsetAsync(mn, 0xFF, 16, ctx->stream));
  PCC_HIP(hipMemsetAsync(cnt, 0, 8, ctx->stream));
  unsigned g = nblk(m, 256);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_vox_valid, dim3(g), dim3(256), 0, ctx->stream, (const float4*)d_xyzrgba, m, depth_clip,
                     d_valid, mn, cnt);
  PCC_CHECK_LAUNCH();
  char* h = (char*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(h, mn, 16, hipMemcpyDeviceToHost, ctx->stream));
  PCC_HIP(hipMemcpyAsync(h + 16, cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
  PCC_HIP(hipStreamSynchronize(ctx->stream));
  *h_n_valid = (int64_t)*(unsigned long long*)(h + 16);
  for (int a = 0; a < 3; ++a) {
    uint32_t o = ((uint32_t*)h)[a];
    o = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;   // inverse of f2ord
    float f;
    memcpy(&f, &o, 4);
    h_min_bound[a] = f;
  }
  return PCC_OK;
}

extern "C" int pcc_vox_keys(pcc_ctx* ctx, const float* d_xyzrgba, const uint8_t* d_valid, int64_t m,
                            const double* h_voxel_min_bound, double voxel_size, uint64_t* d_keys, int32_t* d_flag) {
  PCC_REQUIRE(ctx && h_voxel_min_bound && voxel_size > 0, PCC_E_ARG, "pcc_vox_keys: bad arg");
  if (m <= 0) return PCC_OK;
  PCC_REQUIRE(d_xyzrgba && d_valid && d_keys && d_flag, PCC_E_ARG, "pcc_vox_keys: null buffers");
  hipLaunchKernelGGL(k_vox_keys, dim3(nblk(m, 256)), dim3(256), 0, ctx->stream, (const float4*)d_xyzrgba, d_valid,
                     m, h_voxel_min_bound[0], h_voxel_min_bound[1], h_voxel_min_bound[2], voxel_size, d_keys, d_flag);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_vox_mean(pcc_ctx* ctx, const float* d_xyzrgba, const uint64_t* d_sorted_keys,
                            const uint32_t* d_perm, int64_t n_valid, double voxel_size, int32_t* d_out_coords,
                            double* d_out_colors, int64_t cap, int64_t* h_n_voxels) {
  PCC_REQUIRE(ctx && h_n_voxels && voxel_size > 0, PCC_E_ARG, "pcc_vox_mean: bad arg");
  *h_n_voxels = 0;
  if (n_valid <= 0) return PCC_OK;
  PCC_REQUIRE(d_xyzrgba && d_sorted_keys && d_perm && d_out_coords && d_out_colors, PCC_E_ARG,
              "pcc_vox_mean: null buffers");
  PCC_REQUIRE(n_valid < ((int64_t)1 << 31), PCC_E_ARG, "pcc_vox_mean: n too large");
  hipStream_t st = ctx->stream;
  PCC_TRY(pcc_arena_reserve(ctx, 2 * pcc_align((size_t)n_valid * 4) + pcc_scan_scratch_bytes(n_valid) + 512));
  uint32_t* flags = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n_valid * 4);
  uint32_t* excl = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n_valid * 4);
  uint32_t* total = (uint32_t*)pcc_arena_alloc(ctx, 4);
  if (!flags || !excl || !total) return PCC_E_NOMEM;
  hipLaunchKernelGGL(k_vox_flags, dim3(nblk(n_valid, 256)), dim3(256), 0, st, d_sorted_keys, n_valid, flags);
  PCC_CHECK_LAUNCH();
  PCC_TRY(pcc_scan_exclusive_u32(ctx, flags, excl, n_valid, total));
  uint32_t* h = (uint32_t*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(h, total, 4, hipMemcpyDeviceToHost, st));
  PCC_HIP(hipStreamSynchronize(st));
  const int64_t v = (int64_t)h[0];
  PCC_REQUIRE(v <= cap, PCC_E_ARG, "pcc_vox_mean: output capacity %lld < %lld voxels", (long long)cap, (long long)v);
  hipLaunchKernelGGL(k_vox_mean, dim3(nblk(n_valid, 256)), dim3(256), 0, st, (const float4*)d_xyzrgba,
                     d_sorted_keys, d_perm, (const uint32_t*)flags, (const uint32_t*)excl, n_valid, voxel_size,
                     d_out_coords, d_out_colors);
  PCC_CHECK_LAUNCH();
  *h_n_voxels = v;
  return PCC_OK;
}

// rows of d_sorted_coords ([n,4], duplicates adjacent) that start a run: indices into the sorted order
extern "C" int pcc_unique_rows(pcc_ctx* ctx, const int32_t* d_sorted_coords, int64_t n, uint32_t* d_rows,
                               int64_t* h_n_unique) {
  PCC_REQUIRE(ctx && h_n_unique, PCC_E_ARG, "pcc_unique_rows: null arg");
  *h_n_unique = 0;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_sorted_coords && d_rows && n < ((int64_t)1 << 31), PCC_E_ARG, "pcc_unique_rows: bad buffers");
  hipStream_t st = ctx->stream;
  PCC_TRY(pcc_arena_reserve(ctx, 2 * pcc_align((size_t)n * 4) + pcc_scan_scratch_bytes(n) + 512));
  uint32_t* flags = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 4);
  uint32_t* excl = (uint32_t*)pcc_arena_alloc(ctx, (size_t)n * 4);
  uint32_t* total = (uint32_t*)pcc_arena_alloc(ctx, 4);
  if (!flags || !excl || !total) return PCC_E_NOMEM;
  hipLaunchKernelGGL(k_row_first_flags, dim3(nblk(n, 256)), dim3(256), 0, st, (const int4*)d_sorted_coords, n, flags);
  PCC_CHECK_LAUNCH();
  PCC_TRY(pcc_scan_exclusive_u32(ctx, flags, excl, n, total));
  hipLaunchKernelGGL(k_flagged_rows, dim3(nblk(n, 256)), dim3(256), 0, st, (const uint32_t*)flags,
                     (const uint32_t*)excl, n, d_rows);
  PCC_CHECK_LAUNCH();
  uint32_t* h = (uint32_t*)ctx->pinned;
  PCC_HIP(hipMemcpyAsync(h, total, 4, hipMemcpyDeviceToHost, st));
  PCC_HIP(hipStreamSynchronize(st));
  *h_n_unique = (int64_t)h[0];
  return PCC_OK;
}
