// Does v_mfma_f32_16x16x4_f32 of one wave overlap with plain VALU work of ANOTHER wave on the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coexec.hip -o /tmp/coexec && /tmp/coexec
// One workgroup of 5 waves: waves 0 and 4 land on the same SIMD (round-robin placement), waves 1..3 exit.
// mode bit 0: wave 0 runs the matrix chain; bit 1: wave 4 runs the VALU chain; bit 2: the matrix chain is bf16
// (v_mfma_f32_32x32x16_bf16) instead of f32.  Prints shader cycles (s_memtime) of each wave's loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(320) void k(int mode, int iters, unsigned long long* out, float* sink) {
  int wave = threadIdx.x >> 6;
  if (mode & 8) wave = wave == 0 ? 4 : wave == 4 ? 0 : wave;  // bit 3: roles swapped (the VALU chain in the OLDER wave)
  unsigned long long t0, t1;
  if (wave == 0 && (mode & 1)) {
    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    f32x16 c = {0}, d = {0};
    const float x = threadIdx.x * 0.001f, y = 1.0f + x;
    bf16x8 p, q;
    for (int i = 0; i < 8; ++i) { p[i] = (__bf16)x; q[i] = (__bf16)y; }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
      if (mode & 4) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, c, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q, p, d, 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, b, 0, 0, 0);
        }
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = a[0] + b[1] + c[0] + d[3];
  } else if (wave == 4 && (mode & 2)) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
    const float m = 1.0001f, n = 0.5f;
    if (mode & 16) __builtin_amdgcn_s_setprio(3);  // bit 4: the VALU wave at raised priority
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], m, n);   // 32 independent-ish v_fma_f32 per iteration
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[1] = t1 - t0;
    float s = 0;
    for (int j = 0; j < 8; ++j) s += v[j];
    sink[threadIdx.x] = s;
  }
}

int main() {
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, 16);
  hipMalloc(&sink, 4096);
  const int iters = 20000;
  const char* names[] = {"", "f32 matrix chain alone", "VALU chain alone", "f32 matrix + VALU on one SIMD", "", "bf16 matrix chain alone", "",
                         "bf16 matrix + VALU on one SIMD"};
  for (int mode : {1, 2, 3, 5, 7, 3 + 8, 3 + 16, 3 + 8 + 16, 7 + 8, 7 + 16}) {
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(out, 0, 16);
      hipLaunchKernelGGL(k, dim3(1), dim3(320), 0, 0, mode, iters, out, sink);
      hipDeviceSynchronize();
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    }
    const char* nm = mode < 8 ? names[mode] : (mode & 8) && (mode & 16) ? "both, VALU wave older + prio 3" : (mode & 8) ? "both, VALU wave older" : "both, VALU wave prio 3";
    printf("%-34s matrix wave %9llu cycles (%.1f / iteration), VALU wave %9llu cycles (%.1f / iteration)\n", mode < 8 ? names[mode] : (mode & 4 ? (mode & 8 ? "bf16 + VALU, VALU wave older" : "bf16 + VALU, VALU prio 3") : nm), h[0],
           (double)h[0] / iters, h[1], (double)h[1] / iters);
  }
  return 0;
}
