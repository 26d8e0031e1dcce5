// What a host-side wait costs: one tiny kernel + hipStreamSynchronize, and one tiny kernel + an event another thread
// would wait on (hipEventSynchronize), N times; with the device's default scheduling flags, after
// hipSetDeviceFlags(hipDeviceScheduleSpin) (argument "spin"), or under ROC_ACTIVE_WAIT_TIMEOUT (environment).
// tools/micro/sync_latency [spin|yield|block] [N]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_tiny(unsigned* p) { if (threadIdx.x == 0) p[0] += 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "default";
  const int n = argc > 2 ? atoi(argv[2]) : 2000;
  if (!strcmp(mode, "spin")) CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
  if (!strcmp(mode, "yield")) CK(hipSetDeviceFlags(hipDeviceScheduleYield));
  if (!strcmp(mode, "block")) CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
  unsigned* d;
  CK(hipMalloc(&d, 256));
  CK(hipMemset(d, 0, 256));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d); CK(hipStreamSynchronize(st)); }
  double t0 = now();
  for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d); CK(hipStreamSynchronize(st)); }
  double t1 = now();
  for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d); CK(hipEventRecord(ev, st)); CK(hipEventSynchronize(ev)); }
  double t2 = now();
  // a wait that starts long before the work ends (a 200-us kernel chain): what the waiter adds behind the GPU's end
  printf("%s: launch + hipStreamSynchronize %.2f us, launch + event record + hipEventSynchronize %.2f us (n = %d)\n", mode,
         1e6 * (t1 - t0) / n, 1e6 * (t2 - t1) / n, n);
  return 0;
}
