// What four waves per SIMD get out of the matrix pipe when every wave alternates chains of v_mfma_f32_16x16x4_f32 with
// vector / scalar / LDS work the way a k_gconv16 offset step does (conv16.h): per "step" M items of 16 MFMAs (two
// interleaved chains of 8), V dependent vector instructions, S scalar instructions, optionally an accumulator tile
// round trip through LDS per item.  Prints matrix-pipe busy = MFMAs x 32 cycles / elapsed cycles per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mix.hip -o tools/micro/mix && tools/micro/mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, int S, bool LDS, int G = 0, int B = 0, int OOB = -1>
__global__ void k(int iters, int items, unsigned long long* out, float* sink, const float4* __restrict__ gbuf = nullptr) {
  __shared__ __attribute__((aligned(16))) float tile[16][64 * 4 + 4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
  float x = threadIdx.x * 0.001f, y = 1.0f + x;
  int v = threadIdx.x;
  int sacc = iters;
  float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gq[G > 0 ? G : 1];
  for (int u = 0; u < (G > 0 ? G : 1); ++u) gq[u] = gsum;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    for (int g = 0; g < items; ++g) {
      if (LDS) {
        const float4 a = *reinterpret_cast<const float4*>(&tile[wave][lane * 4]);
        lo[0] += a.x; lo[1] += a.y; lo[2] += a.z; lo[3] += a.w;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        lo = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, hi, 0, 0, 0);
      }
      if (LDS) *reinterpret_cast<float4*>(&tile[wave][lane * 4]) = make_float4(lo[0], lo[1], lo[2], lo[3]);
    }
    if (G > 0) {   // G 16-byte loads per lane from a 64-KB buffer, consumed one step later (as the gathers and weights are)
#pragma unroll
      for (int u = 0; u < G; ++u) { gsum.x += gq[u].x; }
      if constexpr (OOB < 0) {
#pragma unroll
        for (int u = 0; u < G; ++u) gq[u] = gbuf[((i * 7 + u * 64 + lane) & 4095)];
      } else {   // raw buffer loads; the first OOB of them beyond num_records (return zero, fetch nothing)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gbuf, 0, 65536, 0x00027000);
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const unsigned off = u < OOB ? 0x7FFF0000u : (unsigned)(((i * 7 + u * 64 + lane) & 4095) * 16);
          const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
          gq[u] = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < B; ++u) {   // B data-dependent (never taken) scalar branches
      if (__builtin_amdgcn_readfirstlane(sacc) == 0x12345 + u) asm volatile("s_nop 1");
    }
#pragma unroll
    for (int u = 0; u < V; ++u) asm volatile("v_mad_u32_u24 %0, %0, 3, %0" : "+v"(v));
#pragma unroll
    for (int u = 0; u < S; ++u) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(sacc));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = lo[0] + hi[0] + (float)v + (float)sacc + gsum.x;
}

template <int V, int S, bool LDS, int G = 0, int B = 0, int OOB = -1>
static void run(int waves_per_simd, int items, unsigned long long* out, float* sink, const float4* gbuf = nullptr) {
  const int iters = 4000;
  const int threads = 256 * waves_per_simd;   // one workgroup on one CU: waves_per_simd waves on each of its 4 SIMDs
  unsigned long long h[16];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<V, S, LDS, G, B, OOB>), dim3(1), dim3(threads), 0, 0, iters, items, out, sink, gbuf);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, 8 * 4 * waves_per_simd, hipMemcpyDeviceToHost);
  }
  unsigned long long mx = 0;
  for (int w = 0; w < 4 * waves_per_simd; ++w) mx = h[w] > mx ? h[w] : mx;
  const double mfma_cycles = (double)iters * items * 16 * 32 * waves_per_simd;   // per SIMD
  printf("waves/SIMD %d  items %d  V %3d  S %3d  LDS %d  G %2d  B %2d  OOB %2d : %8.0f cycles per step and wave, pipe busy %5.1f %%\n", waves_per_simd,
         items, V, S, (int)LDS, G, B, OOB, (double)mx / iters, 100.0 * mfma_cycles / (double)mx);
}

int main() {
  unsigned long long* out;
  float* sink;
  (void)hipMalloc(&out, 8 * 64);
  (void)hipMalloc(&sink, 4 * 4096);
  float4* gbuf;
  (void)hipMalloc(&gbuf, 65536);
  (void)hipMemset(gbuf, 0, 65536);
  for (int w : {4}) {   // the k_gconv16 step: 13 loads, ten branches
    run<50, 40, true, 13, 0>(w, 3, out, sink, gbuf);
    run<50, 40, true, 13, 10>(w, 3, out, sink, gbuf);
    run<50, 40, true, 0, 10>(w, 3, out, sink, gbuf);
    run<50, 40, true, 26, 10>(w, 3, out, sink, gbuf);
    run<50, 40, true, 13, 10, 0>(w, 3, out, sink, gbuf);
    run<50, 40, true, 13, 10, 3>(w, 3, out, sink, gbuf);
    run<50, 40, true, 13, 10, 13>(w, 3, out, sink, gbuf);
  }
  // a 128-row window per wave (5 items, 21 loads, 18 branches per step) at two and three waves per SIMD
  run<60, 45, true, 21, 18>(2, 5, out, sink, gbuf);
  run<60, 45, true, 21, 18>(3, 5, out, sink, gbuf);
  run<60, 45, true, 21, 18, 0>(2, 5, out, sink, gbuf);
  run<50, 40, true, 13, 10, 0>(4, 3, out, sink, gbuf);
  for (int w : {4}) {
    run<0, 0, false>(w, 3, out, sink);
    run<50, 0, false>(w, 3, out, sink);
    run<50, 40, false>(w, 3, out, sink);
    run<50, 40, true>(w, 3, out, sink);
    run<100, 40, true>(w, 3, out, sink);
  }
  return 0;
}
