// entropy.hip — device side of the entropy models: symbol / index formation
// for the factorized bottleneck (z) and the Gaussian conditional (y), and the
// decoder's offset de-quantisation.
//
// Replaces the tensor arithmetic around the CompressAI calls in
//   factorized_model_step_batched  codec_pipeline.py:294-317 / codec_parallel.py:291-318
//   gaussian_model_step_batched    codec_pipeline.py:397-437 / codec_parallel.py:382-419
// Every operation is a single IEEE float32 op (no contraction, no
// transcendental), so encoder, decoder and the CPU oracle derive identical
// integers; the serial rANS itself runs on the host (rans_host.cpp).
// Layouts: features [n,c] row-major; symbols / indexes channel-major [.., c, n]
// (the order CompressAI flattens [B,C,N] in).
#include "common.h"

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

// thread t -> (ch = t / n, i = t % n): coalesced channel-major writes
__global__ __launch_bounds__(256) void k_factorized_quant(const float* __restrict__ z, int64_t n, int c,
                                                          const float* __restrict__ med,
                                                          int32_t* __restrict__ sym,
                                                          float* __restrict__ zhat) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const int ch = (int)(t / n);
  const int64_t i = t - (int64_t)ch * n;
  const float m = med[ch];
  const float r = rintf(__fsub_rn(z[i * c + ch], m));
  sym[t] = (int32_t)r;
  zhat[i * c + ch] = __fadd_rn(r, m);
}

__global__ __launch_bounds__(256) void k_factorized_dequant(const int32_t* __restrict__ sym, int64_t n,
                                                            int c, const float* __restrict__ med,
                                                            float* __restrict__ zhat) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const int ch = (int)(t / n);
  const int64_t i = t - (int64_t)ch * n;
  zhat[i * c + ch] = __fadd_rn((float)sym[t], med[ch]);
}

__device__ __forceinline__ int32_t scale_index(float sc, const float* __restrict__ table, int n_tab) {
  // GaussianConditional.build_indexes: lower-bound at table[0], then
  // idx = (n_tab-1) - #{t in table[:-1] : scale <= t}
  const float s = fmaxf(sc, table[0]);
  int32_t idx = n_tab - 1;
  for (int j = 0; j < n_tab - 1; ++j) idx -= (s <= table[j]) ? 1 : 0;
  return idx;
}

__global__ __launch_bounds__(256) void k_gaussian_quant(
    const float* __restrict__ y, const float* __restrict__ params, int64_t n, int c,
    const float* __restrict__ scale, int nq, const float* __restrict__ table, int n_tab,
    int32_t* __restrict__ sym, int32_t* __restrict__ idx) {
  __shared__ float tab[64];
  if (threadIdx.x < n_tab) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const int ch = (int)(t / n);
  const int64_t i = t - (int64_t)ch * n;
  const float yv = y[i * c + ch];
  const float sc = params[i * 2 * c + ch];
  const float mu = params[i * 2 * c + c + ch];
  for (int q = 0; q < nq; ++q) {
    const float s = scale[q * c + ch];
    const float v = __fsub_rn(__fmul_rn(yv, s), __fmul_rn(mu, s));
    sym[(int64_t)q * n * c + t] = (int32_t)rintf(v);
    idx[(int64_t)q * n * c + t] = scale_index(__fmul_rn(sc, s), tab, n_tab);
  }
}

// compact form: int16 symbols / uint8 indexes (3 B instead of 8 B per symbol over PCIe);
// *flag is OR-ed with 1 if a symbol does not fit int16 (caller then uses the int32 form)
__global__ __launch_bounds__(256) void k_gaussian_quant16(
    const float* __restrict__ y, const float* __restrict__ params, int64_t n, int c,
    const float* __restrict__ scale, int nq, const float* __restrict__ table, int n_tab,
    int16_t* __restrict__ sym, uint8_t* __restrict__ idx, int32_t* __restrict__ flag) {
  __shared__ float tab[64];
  if (threadIdx.x < n_tab) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const int ch = (int)(t / n);
  const int64_t i = t - (int64_t)ch * n;
  const float yv = y[i * c + ch];
  const float sc = params[i * 2 * c + ch];
  const float mu = params[i * 2 * c + c + ch];
  bool over = false;
  for (int q = 0; q < nq; ++q) {
    const float s = scale[q * c + ch];
    const float v = rintf(__fsub_rn(__fmul_rn(yv, s), __fmul_rn(mu, s)));
    over |= !(v >= -32768.0f && v <= 32767.0f);
    sym[(int64_t)q * n * c + t] = (int16_t)(int32_t)v;
    idx[(int64_t)q * n * c + t] = (uint8_t)scale_index(__fmul_rn(sc, s), tab, n_tab);
  }
  if (over) atomicOr(flag, 1);
}

// int32 symbols / uint8 indexes: what the GPU coder reads (rans_gpu.hip); no overflow case
__global__ __launch_bounds__(256) void k_gaussian_quant_dev(
    const float* __restrict__ y, const float* __restrict__ params, int64_t n, int c,
    const float* __restrict__ scale, int nq, const float* __restrict__ table, int n_tab,
    int32_t* __restrict__ sym, uint8_t* __restrict__ idx) {
  __shared__ float tab[64];
  if (threadIdx.x < n_tab) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const int ch = (int)(t / n);
  const int64_t i = t - (int64_t)ch * n;
  const float yv = y[i * c + ch];
  const float sc = params[i * 2 * c + ch];
  const float mu = params[i * 2 * c + c + ch];
  for (int q = 0; q < nq; ++q) {
    const float s = scale[q * c + ch];
    const float v = __fsub_rn(__fmul_rn(yv, s), __fmul_rn(mu, s));
    sym[(int64_t)q * n * c + t] = (int32_t)rintf(v);
    idx[(int64_t)q * n * c + t] = (uint8_t)scale_index(__fmul_rn(sc, s), tab, n_tab);
  }
}

__global__ __launch_bounds__(256) void k_gaussian_indexes8(const float* __restrict__ params, int64_t n,
                                                           int c, const float* __restrict__ scale,
                                                           const float* __restrict__ table, int n_tab,
                                                           uint8_t* __restrict__ idx) {
  __shared__ float tab[64];
  if (threadIdx.x < n_tab) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const int ch = (int)(t / n);
  const int64_t i = t - (int64_t)ch * n;
  idx[t] = (uint8_t)scale_index(__fmul_rn(params[i * 2 * c + ch], scale[ch]), tab, n_tab);
}

__global__ __launch_bounds__(256) void k_gaussian_indexes(const float* __restrict__ params, int64_t n,
                                                          int c, const float* __restrict__ scale,
                                                          const float* __restrict__ table, int n_tab,
                                                          int32_t* __restrict__ idx) {
  __shared__ float tab[64];
  if (threadIdx.x < n_tab) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const int ch = (int)(t / n);
  const int64_t i = t - (int64_t)ch * n;
  idx[t] = scale_index(__fmul_rn(params[i * 2 * c + ch], scale[ch]), tab, n_tab);
}

__global__ __launch_bounds__(256) void k_gaussian_dequant(const int32_t* __restrict__ sym,
                                                          const float* __restrict__ params, int64_t n,
                                                          int c, const float* __restrict__ scale,
                                                          float bound, float off_a, float off_b,
                                                          float* __restrict__ yhat) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const int ch = (int)(t / n);
  const int64_t i = t - (int64_t)ch * n;
  const float s = scale[ch];
  const float rescale = __fdiv_rn(1.0f, s);
  const float sigma = fmaxf(__fmul_rn(params[i * 2 * c + ch], s), bound);
  const float mu = params[i * 2 * c + c + ch];
  const int32_t q = sym[t];
  const float q_abs = fabsf((float)q);
  const float sign = (q > 0) ? 1.0f : ((q < 0) ? -1.0f : 0.0f);
  // get_offsets(sigma, scale) := off_a / (off_b + sigma); applied negated, zero for the zero bin
  float q_off = -__fdiv_rn(off_a, __fadd_rn(off_b, sigma));
  if (q_abs < 0.0001f) q_off = 0.0f;
  const float v = __fmul_rn(sign, __fadd_rn(q_abs, q_off));
  yhat[i * c + ch] = __fadd_rn(__fmul_rn(v, rescale), mu);
}

// ---- element-wise forms behind the CompressAI-shaped methods (any tensor shape, flat) --------
__global__ __launch_bounds__(256) void k_build_indexes(const float* __restrict__ scales, int64_t n,
                                                       const float* __restrict__ table, int n_tab,
                                                       int32_t* __restrict__ idx) {
  __shared__ float tab[64];
  if (threadIdx.x < n_tab) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) idx[t] = scale_index(scales[t], tab, n_tab);
}

__global__ __launch_bounds__(256) void k_quantize_symbols(const float* __restrict__ x,
                                                          const float* __restrict__ means /*nullable*/,
                                                          int64_t n, int32_t* __restrict__ sym) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float v = means ? __fsub_rn(x[t], means[t]) : x[t];
  sym[t] = (int32_t)rintf(v);
}

extern "C" int pcc_build_indexes(pcc_ctx* ctx, const float* d_scales, int64_t n, const float* d_table,
                                 int n_tab, int32_t* d_idx) {
  PCC_REQUIRE(ctx && n_tab >= 2 && n_tab <= 64, PCC_E_ARG, "pcc_build_indexes: bad arg");
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_scales && d_table && d_idx, PCC_E_ARG, "pcc_build_indexes: null buffers");
  hipLaunchKernelGGL(k_build_indexes, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_scales, n, d_table,
                     n_tab, d_idx);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_quantize_symbols(pcc_ctx* ctx, const float* d_x, const float* d_means, int64_t n,
                                    int32_t* d_sym) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_quantize_symbols: null ctx");
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_x && d_sym, PCC_E_ARG, "pcc_quantize_symbols: null buffers");
  hipLaunchKernelGGL(k_quantize_symbols, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_x, d_means, n, d_sym);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_factorized_quant(pcc_ctx* ctx, const float* d_z, int64_t n, int c,
                                    const float* d_med, int32_t* d_sym, float* d_zhat) {
  PCC_REQUIRE(ctx && c >= 1, PCC_E_ARG, "pcc_factorized_quant: bad arg");
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_z && d_med && d_sym && d_zhat, PCC_E_ARG, "pcc_factorized_quant: null buffers");
  hipLaunchKernelGGL(k_factorized_quant, dim3(nblk(n * c, 256)), dim3(256), 0, ctx->stream, d_z, n, c,
                     d_med, d_sym, d_zhat);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_factorized_dequant(pcc_ctx* ctx, const int32_t* d_sym, int64_t n, int c,
                                      const float* d_med, float* d_zhat) {
  PCC_REQUIRE(ctx && c >= 1, PCC_E_ARG, "pcc_factorized_dequant: bad arg");
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_sym && d_med && d_zhat, PCC_E_ARG, "pcc_factorized_dequant: null buffers");
  hipLaunchKernelGGL(k_factorized_dequant, dim3(nblk(n * c, 256)), dim3(256), 0, ctx->stream, d_sym, n,
                     c, d_med, d_zhat);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_gaussian_quant(pcc_ctx* ctx, const float* d_y, const float* d_params, int64_t n,
                                  int c, const float* d_scale, int q, const float* d_table, int n_tab,
                                  int32_t* d_sym, int32_t* d_idx) {
  PCC_REQUIRE(ctx && c >= 1 && q >= 1 && n_tab >= 2 && n_tab <= 64, PCC_E_ARG,
              "pcc_gaussian_quant: bad arg (c=%d q=%d n_tab=%d)", c, q, n_tab);
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_y && d_params && d_scale && d_table && d_sym && d_idx, PCC_E_ARG,
              "pcc_gaussian_quant: null buffers");
  hipLaunchKernelGGL(k_gaussian_quant, dim3(nblk(n * c, 256)), dim3(256), 0, ctx->stream, d_y, d_params,
                     n, c, d_scale, q, d_table, n_tab, d_sym, d_idx);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_gaussian_quant16(pcc_ctx* ctx, const float* d_y, const float* d_params, int64_t n,
                                    int c, const float* d_scale, int q, const float* d_table, int n_tab,
                                    int16_t* d_sym, uint8_t* d_idx, int32_t* d_flag) {
  PCC_REQUIRE(ctx && c >= 1 && q >= 1 && n_tab >= 2 && n_tab <= 64, PCC_E_ARG,
              "pcc_gaussian_quant16: bad arg (c=%d q=%d n_tab=%d)", c, q, n_tab);
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_y && d_params && d_scale && d_table && d_sym && d_idx && d_flag, PCC_E_ARG,
              "pcc_gaussian_quant16: null buffers");
  hipLaunchKernelGGL(k_gaussian_quant16, dim3(nblk(n * c, 256)), dim3(256), 0, ctx->stream, d_y, d_params,
                     n, c, d_scale, q, d_table, n_tab, d_sym, d_idx, d_flag);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_gaussian_quant_dev(pcc_ctx* ctx, const float* d_y, const float* d_params, int64_t n, int c,
                                      const float* d_scale, int q, const float* d_table, int n_tab, int32_t* d_sym,
                                      uint8_t* d_idx) {
  PCC_REQUIRE(ctx && c >= 1 && q >= 1 && n_tab >= 2 && n_tab <= 64, PCC_E_ARG,
              "pcc_gaussian_quant_dev: bad arg (c=%d q=%d n_tab=%d)", c, q, n_tab);
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_y && d_params && d_scale && d_table && d_sym && d_idx, PCC_E_ARG, "pcc_gaussian_quant_dev: null buffers");
  hipLaunchKernelGGL(k_gaussian_quant_dev, dim3(nblk(n * c, 256)), dim3(256), 0, ctx->stream, d_y, d_params, n, c,
                     d_scale, q, d_table, n_tab, d_sym, d_idx);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_gaussian_indexes8(pcc_ctx* ctx, const float* d_params, int64_t n, int c,
                                     const float* d_scale, const float* d_table, int n_tab,
                                     uint8_t* d_idx) {
  PCC_REQUIRE(ctx && c >= 1 && n_tab >= 2 && n_tab <= 64, PCC_E_ARG, "pcc_gaussian_indexes8: bad arg");
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_params && d_scale && d_table && d_idx, PCC_E_ARG, "pcc_gaussian_indexes8: null buffers");
  hipLaunchKernelGGL(k_gaussian_indexes8, dim3(nblk(n * c, 256)), dim3(256), 0, ctx->stream, d_params, n,
                     c, d_scale, d_table, n_tab, d_idx);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_gaussian_indexes(pcc_ctx* ctx, const float* d_params, int64_t n, int c,
                                    const float* d_scale, const float* d_table, int n_tab,
                                    int32_t* d_idx) {
  PCC_REQUIRE(ctx && c >= 1 && n_tab >= 2 && n_tab <= 64, PCC_E_ARG, "pcc_gaussian_indexes: bad arg");
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_params && d_scale && d_table && d_idx, PCC_E_ARG, "pcc_gaussian_indexes: null buffers");
  hipLaunchKernelGGL(k_gaussian_indexes, dim3(nblk(n * c, 256)), dim3(256), 0, ctx->stream, d_params, n,
                     c, d_scale, d_table, n_tab, d_idx);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_gaussian_dequant(pcc_ctx* ctx, const int32_t* d_sym, const float* d_params,
                                    int64_t n, int c, const float* d_scale, float bound, float off_a,
                                    float off_b, float* d_yhat) {
  PCC_REQUIRE(ctx && c >= 1, PCC_E_ARG, "pcc_gaussian_dequant: bad arg");
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_sym && d_params && d_scale && d_yhat, PCC_E_ARG, "pcc_gaussian_dequant: null buffers");
  hipLaunchKernelGGL(k_gaussian_dequant, dim3(nblk(n * c, 256)), dim3(256), 0, ctx->stream, d_sym,
                     d_params, n, c, d_scale, bound, off_a, off_b, d_yhat);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}
