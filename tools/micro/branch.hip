// What does a lone wave pay for a scalar compare-and-branch between two runs of v_mfma_f32_16x16x4_f32?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/branch.hip -o tools/micro/branch && tools/micro/branch
// Loop body: 16 MFMAs, then a branch on a loop-invariant SGPR around a block of 16 more MFMAs (K bytes of code), then
// the loop back.  Modes: the block never executes (the forward branch is taken), always executes (not taken).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M>
__global__ __launch_bounds__(64, 2) void k(int iters, int flag, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x;
  float w[16], x[8];
  for (int j = 0; j < 16; ++j) w[j] = 0.001f * (lane + j);
  for (int j = 0; j < 8; ++j) x[j] = 0.5f + j;
  f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < M; ++r) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        lo = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], x[s], lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8 + s], x[s], hi, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (flag > r) {   // scalar, loop-invariant, opaque
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          lo = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8 + s], x[s], lo, 0, 0, 0);
          hi = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], x[s], hi, 0, 0, 0);
        }
        asm volatile("s_nop 0" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[0] = t1 - t0;
  sink[lane] = lo[0] + hi[1];
}

int main() {
  unsigned long long* out;
  float* sink;
  (void)hipMalloc(&out, 16);
  (void)hipMalloc(&sink, 4096);
  const int iters = 4000;
#define CASE(M, FLAG, NAME) do { unsigned long long h = 0; for (int rep = 0; rep < 2; ++rep) { \
      hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, iters, FLAG, out, sink); \
      (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); } \
    printf("%-64s %7.1f cycles / iteration\n", NAME, (double)h / iters); } while (0)
  CASE(1, 0, "16 MFMAs + branch taken around 16 more");
  CASE(1, 1, "16 MFMAs + 16 more (branch not taken)");
  CASE(4, 0, "4 x (16 MFMAs + branch taken)            [ideal 2048]");
  CASE(4, 2, "4 x 16 MFMAs, 2 blocks run, 2 skipped     [ideal 3072]");
  CASE(4, 4, "4 x 16 MFMAs, 4 blocks run                [ideal 4096]");
  return 0;
}
