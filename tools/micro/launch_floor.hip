// What a dependent launch costs on one stream: N tiny kernels (one workgroup, or 256 workgroups writing 1 MB) queued
// back to back, plain launches against the same chain replayed as a HIP graph.  tools/micro/launch_floor [N]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_tiny(unsigned* p, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] + v; }
__global__ void k_wide(unsigned* p, unsigned v) { p[blockIdx.x * blockDim.x + threadIdx.x] += v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 200;
  unsigned* d;
  CK(hipMalloc(&d, 1 << 20));
  CK(hipMemset(d, 0, 1 << 20));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int wide = 0; wide < 2; ++wide) {
    auto chain = [&]() {
      for (int i = 0; i < n; ++i) {
        if (wide) hipLaunchKernelGGL(k_wide, dim3(1024), dim3(256), 0, st, d, 1u);
        else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d, 1u);
      }
    };
    chain();
    CK(hipStreamSynchronize(st));
    float ms = 0;
    double t0 = now();
    CK(hipEventRecord(e0, st));
    chain();
    CK(hipEventRecord(e1, st));
    double t1 = now();
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: %d plain launches: %.2f us each on the GPU, %.2f us each to queue\n", wide ? "1024 workgroups" : "one wave", n,
           1e3 * ms / n, 1e6 * (t1 - t0) / n);
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    t0 = now();
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    t1 = now();
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: the same as one graph: %.2f us per node on the GPU, %.2f us per node to queue\n", wide ? "1024 workgroups" : "one wave",
           1e3 * ms / n, 1e6 * (t1 - t0) / n);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
