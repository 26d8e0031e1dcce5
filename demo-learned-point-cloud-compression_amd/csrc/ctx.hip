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
  pcc_wcache_free(c);
  if (c->arena) (void)hipFree(c->arena);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->stage) (void)hipHostFree(c->stage);
  if (c->topk_hist[0]) (void)hipFree(c->topk_hist[0]);   // one allocation holds both
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  for (int i = 0; i < c->prof_cap; ++i)
    if (c->prof[i].e0) { (void)hipEventDestroy(c->prof[i].e0); (void)hipEventDestroy(c->prof[i].e1); }
  free(c->prof);
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

// ---------------------------------------------------------------- profiler
PccProfScope::PccProfScope(pcc_ctx* ctx, const char* op, int64_t d0, int64_t d1, int64_t d2, int64_t d3)
    : c(ctx), slot(-1) {
  if (!c || !c->prof_on) return;
  if (c->prof_only[0] && strncmp(op, c->prof_only, strlen(c->prof_only)) != 0) return;
  if (c->prof_only[0] && c->prof_only_d0 >= 0 && d0 != c->prof_only_d0) return;
  if (c->prof_n == c->prof_cap) {
    const int ncap = c->prof_cap ? c->prof_cap * 2 : 256;
    pcc_prof_rec* np = (pcc_prof_rec*)realloc(c->prof, sizeof(pcc_prof_rec) * (size_t)ncap);
    if (!np) return;
    for (int i = c->prof_cap; i < ncap; ++i) {
      np[i].e0 = nullptr;
      np[i].e1 = nullptr;
    }
    c->prof = np;
    c->prof_cap = ncap;
  }
  pcc_prof_rec& r = c->prof[c->prof_n];
  if (!r.e0 && (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess)) return;
  r.op = op;
  r.dims[0] = d0; r.dims[1] = d1; r.dims[2] = d2; r.dims[3] = d3;
  if (hipEventRecord(r.e0, c->stream) != hipSuccess) return;
  slot = c->prof_n++;
}

PccProfScope::~PccProfScope() {
  if (slot >= 0) (void)hipEventRecord(c->prof[slot].e1, c->stream);
}

extern "C" int pcc_prof_enable(pcc_ctx* c, int on) {
  PCC_REQUIRE(c, PCC_E_ARG, "null ctx");
  c->prof_n = 0;
  // `on` > 1 pre-creates that many event pairs so that the timed region only records
  if (on > 1 && on > c->prof_cap) {
    pcc_prof_rec* np = (pcc_prof_rec*)realloc(c->prof, sizeof(pcc_prof_rec) * (size_t)on);
    PCC_REQUIRE(np, PCC_E_NOMEM, "pcc_prof_enable: out of memory");
    for (int i = c->prof_cap; i < on; ++i) {
      np[i].e0 = nullptr;
      np[i].e1 = nullptr;
    }
    c->prof = np;
    c->prof_cap = on;
  }
  if (on > 1)
    for (int i = 0; i < c->prof_cap; ++i)
      if (!c->prof[i].e0) {
        PCC_HIP(hipEventCreate(&c->prof[i].e0));
        PCC_HIP(hipEventCreate(&c->prof[i].e1));
      }
  c->prof_on = on != 0;
  return PCC_OK;
}

extern "C" int pcc_prof_only(pcc_ctx* c, const char* h_op_prefix, int64_t d0) {
  PCC_REQUIRE(c, PCC_E_ARG, "null ctx");
  snprintf(c->prof_only, sizeof(c->prof_only), "%s", h_op_prefix ? h_op_prefix : "");
  c->prof_only_d0 = d0;
  return PCC_OK;
}

extern "C" int pcc_prof_count(pcc_ctx* c) { return c ? c->prof_n : 0; }

extern "C" int pcc_prof_get(pcc_ctx* c, int i, char* h_op, int cap, float* h_ms, int64_t* h_dims) {
  PCC_REQUIRE(c && h_op && h_ms && h_dims && cap > 0, PCC_E_ARG, "pcc_prof_get: null arg");
  PCC_REQUIRE(i >= 0 && i < c->prof_n, PCC_E_ARG, "pcc_prof_get: index %d of %d", i, c->prof_n);
  pcc_prof_rec& r = c->prof[i];
  PCC_HIP(hipEventSynchronize(r.e1));
  PCC_HIP(hipEventElapsedTime(h_ms, r.e0, r.e1));
  snprintf(h_op, (size_t)cap, "%s", r.op);
  for (int k = 0; k < 4; ++k) h_dims[k] = r.dims[k];
  return PCC_OK;
}

// number of entries >= 0 in an int32 array (active pairs of a rule book)
__global__ void k_count_nonneg(const int32_t* __restrict__ p, int64_t n, unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    c += p[i] >= 0 ? 1ull : 0ull;
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

extern "C" int pcc_count_nonneg(pcc_ctx* c, const int32_t* d_p, int64_t n, int64_t* h_count) {
  PCC_REQUIRE(c && h_count, PCC_E_ARG, "pcc_count_nonneg: null arg");
  *h_count = 0;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_p, PCC_E_ARG, "pcc_count_nonneg: null buffer");
  PCC_TRY(pcc_arena_reserve(c, 256));
  unsigned long long* out = (unsigned long long*)pcc_arena_alloc(c, 8);
  if (!out) return PCC_E_NOMEM;
  PCC_HIP(hipMemsetAsync(out, 0, 8, c->stream));
  unsigned g = (unsigned)((n + 255) / 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_count_nonneg, dim3(g), dim3(256), 0, c->stream, d_p, n, out);
  PCC_CHECK_LAUNCH();
  PCC_HIP(hipMemcpyAsync(c->pinned, out, 8, hipMemcpyDeviceToHost, c->stream));
  PCC_HIP(hipStreamSynchronize(c->stream));
  *h_count = (int64_t)*(unsigned long long*)c->pinned;
  return PCC_OK;
}
