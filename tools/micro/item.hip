// The pieces of one item of k_gconv_up's remainder, one wave alone on its SIMD: where do the cycles beyond 16 x 32 go?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/item.hip -o tools/micro/item && tools/micro/item
// Loop body = 16 v_mfma_f32_16x16x4_f32 in two alternating chains plus, by template bit:
//   1  the accumulator tile read from LDS (two ds_read_b128, requested one item ahead behind the first MFMA pair)
//   2  the tile written back at the end of the item (two ds_write_b128 behind its last MFMA)
//   4  the tile written back one item late, behind the first MFMA pair of the next item (what the kernel does)
//   8  two 16-B buffer loads per item, consumed FOUR items later (a ring: latency hidden, issue cost only)
//   16 the address VALU of the kernel's item (v_xor for the tile, v_or for the gather), each where the kernel has it
//   32 the same VALU in one group
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int M>
__global__ __launch_bounds__(64, 2) void k(int iters, const float* __restrict__ src, int src_bytes, unsigned long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) float acc[2 * 129 * 16];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2 * 129 * 16; i += 64) acc[i] = 0.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00027000);
  float w[16];
  for (int j = 0; j < 16; ++j) w[j] = 0.001f * (lane + j);
  u32x4 g[4][2];
  for (int r = 0; r < 4; ++r) { g[r][0] = u32x4{1, 2, 3, 4}; g[r][1] = u32x4{5, 6, 7, 8}; }
  const unsigned q16 = (unsigned)(lane >> 4) * 16u, qoff = (unsigned)(lane >> 4) * 32u;
  unsigned rows[4], recs[4];
  for (int r = 0; r < 4; ++r) { rows[r] = (unsigned)(((lane & 15) + 16 * r) * 64); recs[r] = (unsigned)((lane & 15) + 16 * r) * 128u; }
  f32x4 lo[2], hi[2];
  lo[0] = lo[1] = hi[0] = hi[1] = f32x4{0, 0, 0, 0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; i += 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // item r of a "step": tiles alternate between two register sets, as in the kernel
      f32x4& L = lo[r & 1]; f32x4& H = hi[r & 1]; f32x4& PL = lo[(r + 1) & 1]; f32x4& PH = hi[(r + 1) & 1];
      float x[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[j] = __uint_as_float(g[r][0][j]); x[4 + j] = __uint_as_float(g[r][1][j]); }
      unsigned a_prev = rows[(r + 3) & 3], a_next = rows[(r + 1) & 3], goff = recs[r];
      if (M & 32) { a_prev ^= q16; a_next ^= q16; goff |= qoff; }
      __builtin_amdgcn_sched_barrier(0);
      L = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], x[0], L, 0, 0, 0);
      H = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8], x[0], H, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (M & 4) {
        if (M & 16) a_prev ^= q16;
        float* b = reinterpret_cast<float*>(reinterpret_cast<char*>(acc) + a_prev);
        *reinterpret_cast<float4*>(b) = make_float4(PL[0], PL[1], PL[2], PL[3]);
        *reinterpret_cast<float4*>(b + 129 * 16) = make_float4(PH[0], PH[1], PH[2], PH[3]);
      }
      if (M & 1) {
        if (M & 16) a_next ^= q16;
        const float* b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(acc) + a_next);
        const float4 a4 = *reinterpret_cast<const float4*>(b), b4 = *reinterpret_cast<const float4*>(b + 129 * 16);
        PL = f32x4{a4.x, a4.y, a4.z, a4.w};
        PH = f32x4{b4.x, b4.y, b4.z, b4.w};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 1; s < 8; ++s) {
        L = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], x[s], L, 0, 0, 0);
        H = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8 + s], x[s], H, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (M & 2) {
        float* b = reinterpret_cast<float*>(reinterpret_cast<char*>(acc) + rows[r]);
        *reinterpret_cast<float4*>(b) = make_float4(L[0], L[1], L[2], L[3]);
        *reinterpret_cast<float4*>(b + 129 * 16) = make_float4(H[0], H[1], H[2], H[3]);
      }
      if (M & 8) {
        if (M & 16) goff |= qoff;
        g[r][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff & 0x3FFFu, 0, 0);
        g[r][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (goff & 0x3FFFu) + 16u, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[0] = t1 - t0;
  float s = lo[0][0] + hi[0][1] + lo[1][2] + hi[1][3];
  for (int r = 0; r < 4; ++r) s += __uint_as_float(g[r][0][0]) + __uint_as_float(g[r][1][1]);
  sink[lane] = s;
}

int main() {
  unsigned long long* out;
  float *sink, *src;
  (void)hipMalloc(&out, 16);
  (void)hipMalloc(&sink, 4096);
  (void)hipMalloc(&src, 1 << 20);
  (void)hipMemset(src, 0, 1 << 20);
  const int iters = 4000;
#define CASE(M, NAME) do { unsigned long long h = 0; for (int rep = 0; rep < 2; ++rep) { \
      hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, iters, src, 1 << 20, out, sink); \
      (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); } \
    printf("%-60s %7.1f cycles / item\n", NAME, (double)h / iters); } while (0)
  CASE(0, "16 MFMAs");
  CASE(1, "+ tile read one item ahead");
  CASE(2, "+ tile written at the end of the item");
  CASE(4, "+ tile written one item late");
  CASE(1 + 2, "+ read ahead, write at the end");
  CASE(1 + 4, "+ read ahead, write late");
  CASE(8, "+ two buffer loads, consumed 4 items later");
  CASE(1 + 4 + 8, "+ read, late write, loads");
  CASE(1 + 4 + 8 + 16, "+ read, late write, loads, address VALU in place");
  CASE(1 + 4 + 8 + 32, "+ read, late write, loads, address VALU grouped");
  return 0;
}
