// Issue rate of v_mfma_f32_16x16x4_f32 from ONE wave as a function of the number of independent accumulator chains.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/chains.hip -o tools/micro/chains && tools/micro/chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NC>
__global__ __launch_bounds__(64) void k(int iters, unsigned long long* out, float* sink) {
  f32x4 acc[NC];
  for (int c = 0; c < NC; ++c) acc[c] = f32x4{0, 0, 0, 0};
  const float x = threadIdx.x * 0.001f, y = 1.0f + x;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[c], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) out[0] = t1 - t0;
  float s = 0;
  for (int c = 0; c < NC; ++c) s += acc[c][0];
  sink[threadIdx.x] = s;
}

template <int NC>
static void run(unsigned long long* out, float* sink) {
  const int iters = 20000;
  unsigned long long h = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<NC>, dim3(1), dim3(64), 0, 0, iters, out, sink);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  }
  printf("%d chains: %.2f cycles per v_mfma_f32_16x16x4_f32\n", NC, (double)h / iters / (4 * NC));
}

int main() {
  unsigned long long* out;
  float* sink;
  (void)hipMalloc(&out, 16);
  (void)hipMalloc(&sink, 4096);
  run<1>(out, sink);
  run<2>(out, sink);
  run<3>(out, sink);
  run<4>(out, sink);
  run<8>(out, sink);
  return 0;
}
