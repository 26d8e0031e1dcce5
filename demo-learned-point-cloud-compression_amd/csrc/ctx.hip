// ctx.hip — context, error string, scratch arena, timers.
#include "common.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void pcc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int pcc_abi_version(void) { return PCC_ABI_VERSION; }
extern "C" const char* pcc_last_error(void) { return g_err; }

extern "C" pcc_ctx* pcc_create(int device, void* stream) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    pcc_set_error("pcc_create: no HIP device visible (%s)",
                  e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0 || device >= ndev) {
    pcc_set_error("pcc_create: device %d out of range [0,%d)", device, ndev);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    pcc_set_error("pcc_create: hipSetDevice(%d) failed", device);
    return nullptr;
  }
  pcc_ctx* c = (pcc_ctx*)calloc(1, sizeof(pcc_ctx));
  if (!c) return nullptr;
  c->device = device;
  c->stream = (hipStream_t)stream;
  c->pinned_cap = 4096;
  if (hipHostMalloc(&c->pinned, c->pinned_cap, hipHostMallocDefault) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    pcc_set_error("pcc_create: resource allocation failed");
    free(c);
    return nullptr;
  }
  return c;
}

extern "C" void pcc_destroy(pcc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->arena) (void)hipFree(c->arena);
  if (c->pinned) (void)hipHostFree(c->pinned);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  free(c);
}

extern "C" int pcc_set_stream(pcc_ctx* c, void* stream) {
  PCC_REQUIRE(c, PCC_E_ARG, "null ctx");
  c->stream = (hipStream_t)stream;
  return PCC_OK;
}

extern "C" int pcc_sync(pcc_ctx* c) {
  PCC_REQUIRE(c, PCC_E_ARG, "null ctx");
  PCC_HIP(hipStreamSynchronize(c->stream));
  return PCC_OK;
}

extern "C" int pcc_timer_start(pcc_ctx* c) {
  PCC_REQUIRE(c, PCC_E_ARG, "null ctx");
  PCC_HIP(hipEventRecord(c->ev0, c->stream));
  c->ev_valid = false;
  return PCC_OK;
}
extern "C" int pcc_timer_stop(pcc_ctx* c) {
  PCC_REQUIRE(c, PCC_E_ARG, "null ctx");
  PCC_HIP(hipEventRecord(c->ev1, c->stream));
  c->ev_valid = true;
  return PCC_OK;
}
extern "C" int pcc_timer_elapsed_ms(pcc_ctx* c, float* h_ms) {
  PCC_REQUIRE(c && h_ms, PCC_E_ARG, "null arg");
  PCC_REQUIRE(c->ev_valid, PCC_E_ARG, "timer not stopped");
  PCC_HIP(hipEventSynchronize(c->ev1));
  PCC_HIP(hipEventElapsedTime(h_ms, c->ev0, c->ev1));
  return PCC_OK;
}

int pcc_arena_reserve(pcc_ctx* c, size_t bytes) {
  pcc_arena_reset(c);
  bytes = pcc_align(bytes) + 4096;
  if (bytes <= c->arena_cap) return PCC_OK;
  // grow geometrically; the old arena may still be read by queued kernels
  size_t want = c->arena_cap ? c->arena_cap : ((size_t)1 << 22);
  while (want < bytes) want *= 2;
  PCC_HIP(hipStreamSynchronize(c->stream));
  if (c->arena) PCC_HIP(hipFree(c->arena));
  c->arena = nullptr;
  c->arena_cap = 0;
  PCC_HIP(hipMalloc((void**)&c->arena, want));
  c->arena_cap = want;
  return PCC_OK;
}

void* pcc_arena_alloc(pcc_ctx* c, size_t bytes) {
  size_t off = pcc_align(c->arena_off);
  if (off + bytes > c->arena_cap) {
    pcc_set_error("scratch arena overflow: need %zu at %zu of %zu", bytes, off,
                  c->arena_cap);
    return nullptr;
  }
  c->arena_off = off + bytes;
  return c->arena + off;
}
