// conv.hip — sparse network layers: gather-convolution (3^3 stride 1 and 2^3
// stride 2 share their kernels), generative transposed convolution, 1x1 linear.
//
// Replaces MinkowskiEngine's convolution / generative transposed convolution /
// linear forward kernels executed inside model.g_a, model.g_s, h_a, h_s
// (codec_pipeline.py:273,287,354; codec_parallel.py:302-303,376,469).
//
// Arithmetic contract (pcc.h): out = bias, then for k ascending over PRESENT
// neighbours, ci ascending: out = fmaf(x, w, out).  v_mfma_f32_32x32x2_f32 and
// v_mfma_f32_16x16x4_f32 are bit-for-bit that chain (2 / 4 ci per instruction,
// k-ordered), so the MFMA kernels and the scalar-fmaf kernel give identical
// bits, and both equal the C oracle.  No atomics: a wave owns its output rows,
// so results do not depend on scheduling.
//
// Kernels of the gather-convolution:
//   k_gconv16 (conv16.h)  every 32 -> 32 and 32 -> 64 layer: the rows of a 64-row window that HAVE an offset are
//                         packed into 16-slot items (ballot + mbcnt), accumulators in LDS
//   k_gconv_first         the 4 -> 32 input layer (16-B feature rows, HBM-bound): dense 32-row tiles, accumulators
//                         in registers, offsets absent for the whole tile skipped
//   k_gconv_scalar        any shape, one thread per output element: the cross-check (PCC_FORCE_SCALAR=1) and the
//                         fallback for shapes the model does not have
#include "common.h"

#include <map>

typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

#define GC_WAVES 4
#ifndef PCC_FIRST_DEPTH
#define PCC_FIRST_DEPTH 8
#endif

#include "conv16.h"
#include "convup.h"
#include "convgen.h"

// The 4 -> 32 input layer (1M rows, 16-B feature rows): HBM-bound (252 MB: 108 MB of rule book, 128 MB of output).
// One wave per 32-row tile, the product computed transposed (D[co][row] = W^T X^T) so that a lane ends up with 16
// channels of one row and stores 16 B at a time.  All 27 neighbour indices of the tile are fetched at once;
// PCC_FIRST_DEPTH row gathers are in flight ahead of the offset being contracted.  Both half-waves fetch the same 32
// rows (same lines, one request each) and a lane keeps the two input channels it feeds to the B operand (K index =
// lane >> 5) — no LDS transpose and no wave barrier in the step.  The 13.8 KB of weights sit in LDS (one load per
// workgroup).  Offsets nobody in the tile has are skipped (11.7 of 27 are present on the bench frame).
// 105 us (round 1: LDS tile, dword stores, weights from global, one gather ahead) -> 77 us.  Tried and slower: 64-row
// windows (registers), index loads split between the half-waves (14 full-wave loads + v_permlane32_swap: +15 us).
// NT: 32-column tiles of the output (cout = any multiple of 16 up to 32 NT: the model default's 32 is NT = 1; columns beyond
// cout are computed on zero weights and not stored)
// FULL: cout == 32 NT, known at compile time (no column predicate anywhere: the shipped shape)
template <int NT, bool FULL>
__global__ __launch_bounds__(GC_WAVES * 64) void k_gconv_first(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int k_vol, int64_t pitch,
    int64_t n_out, const float* __restrict__ w, const float* __restrict__ bias, int relu,
    float* __restrict__ out, int cout_arg) {
  constexpr int CIN = 4;
  const int cout = FULL ? 32 * NT : cout_arg;
  constexpr int COUT = 32 * NT;   // padded width of the LDS copy
  __shared__ float w_lds[27 * CIN * COUT];   // 13.8 KB per tile: one load per workgroup instead of two dword loads per offset and wave
  for (int t = threadIdx.x; t < k_vol * CIN * COUT; t += GC_WAVES * 64) {
    const int col = t % COUT, rowi = t / COUT;
    w_lds[t] = col < cout ? w[rowi * cout + col] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int64_t n_wg = (n_out + 32 * GC_WAVES - 1) / (32 * GC_WAVES);
  // resident workgroups (four per CU: pcc_sparse_conv), tiles taken with the grid's stride: the 13.8 KB of weights are
  // read once per workgroup instead of once per 128 rows (108 MB of L2 reads for 1M rows — as much as the rule book)
  // and nobody waits at the barrier above again.  One box, 1M rows: one workgroup per tile 87.1 us, 768 / 1024 / 2048
  // resident workgroups 79.2 / 72.7 / 76.1; a contiguous run of tiles per workgroup instead of the stride 82.4 (1024).
  for (int64_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x) {
  const int64_t row0 = (wg * GC_WAVES + wave) * 32;
  if (row0 >= n_out) break;  // wave-uniform

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      acc[t][r] = col < cout ? bias[col] : 0.f;
    }

  const bool row_ok = (row0 + i) < n_out;
  {
    int32_t nbs[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) nbs[k] = (row_ok && k < k_vol) ? nbr[(int64_t)k * pitch + row0 + i] : -1;
    // both half-waves fetch the same 32 rows (same lines: one request) and each keeps the two input channels it feeds
    // to the MFMA B operand (K index = lane >> 5): no LDS transpose, no wave barriers in the step
    auto rows_of = [&](int32_t nb) -> float4 {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nb >= 0) v = *reinterpret_cast<const float4*>(in + (int64_t)nb * CIN);
      return v;
    };
    constexpr int D = PCC_FIRST_DEPTH;  // gathers in flight
    float4 gq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gq[d] = rows_of(nbs[d]);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const float4 gc = gq[k % D];
      if (k + D < 27) gq[k % D] = rows_of(nbs[k + D]);  // -1 past k_vol: no load
      if (k < k_vol && __ballot(nbs[k] >= 0) != 0ull) {  // uniform: somebody in this tile has offset k
        const float x0 = h ? gc.y : gc.x, x1 = h ? gc.w : gc.z;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float wc0 = w_lds[(k * CIN + 0 + h) * COUT + 32 * t + i], wc1 = w_lds[(k * CIN + 2 + h) * COUT + 32 * t + i];
          // transposed product D[co][row] = W^T x X^T: a lane ends up with 16 channels of ONE row (16-B stores below)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc0, x0, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc1, x1, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: D layout col = lane & 31 = row slot, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = channel:
  // lane (i, h) holds channels 8 j + 4 h + (0..3), j = 0..3, of row row0 + i: four 16-B stores
  const int64_t g = row0 + i;
  if (g < n_out) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 v = make_float4(acc[t][4 * j], acc[t][4 * j + 1], acc[t][4 * j + 2], acc[t][4 * j + 3]);
        if (relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        const int col = 32 * t + 8 * j + 4 * h;
        if (col < cout) *reinterpret_cast<float4*>(out + g * cout + col) = v;
      }
  }
  }
}

// scalar-fmaf reference path on the GPU (any cin/cout), same bits as the MFMA path
__global__ __launch_bounds__(256) void k_gconv_scalar(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int k_vol, int64_t pitch,
    int64_t n_out, const float* __restrict__ w, const float* __restrict__ bias, int cin, int cout,
    int relu, float* __restrict__ out, int sib) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t / cout;
  const int co = (int)(t - row * cout);
  if (row >= n_out) return;
  float acc = bias[co];
  // sib: siblings first (pcc.h) — the neighbours inside the row's own block of 8 rows, then the others
  for (int pass = sib ? 0 : 1; pass < 2; ++pass)
  for (int k = 0; k < k_vol; ++k) {
    const int32_t nb = nbr[(int64_t)k * pitch + row];
    if (nb < 0) continue;
    if (sib && (((int64_t)nb >> 3) == (row >> 3)) != (pass == 0)) continue;
    const float* x = in + (int64_t)nb * cin;
    const float* wk = w + ((int64_t)k * cin) * cout + co;
    for (int ci = 0; ci < cin; ++ci) acc = fmaf(x[ci], wk[(int64_t)ci * cout], acc);
  }
  if (relu) acc = fmaxf(acc, 0.0f);
  out[t] = acc;
}

// generative transposed convolution, kernel 2 stride 2: out[8p+o] = W[o]^T in[p] + b
// rows (nullable): parent p reads input row rows[p] — the up stage right after a pruning, on the kept rows in place
// CIN: input channels (a multiple of 16); NT: 32-column tiles of the output, cout = any multiple of 16 up to 32 NT (columns
// beyond cout run on zero weights and are not stored).  <32, 1> and <32, 2> are the model default's shapes.
// FULL: cout == 32 NT, known at compile time (no column predicate: the weights of an octant are then plain loads the
// compiler requests together; under a predicate each waited for the one before — 145 -> 238 us on the bench's largest)
template <int CIN, int NT, bool FULL>
__global__ __launch_bounds__(GC_WAVES * 64) void k_convT_mfma(
    const float* __restrict__ in, int64_t n_in, const float* __restrict__ w,
    const float* __restrict__ bias, int relu, float* __restrict__ out,
    const uint32_t* __restrict__ rows, int cout_arg) {
  constexpr int PITCH = CIN + 1;
  const int cout = FULL ? 32 * NT : cout_arg;
  __shared__ float a_lds[GC_WAVES][32 * PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * GC_WAVES + wave) * 32;
  if (row0 >= n_in) return;
  const int i = lane & 31, h = lane >> 5;
  float* a = a_lds[wave];
#pragma unroll
  for (int it = 0; it < CIN / 8; ++it) {
    const int idx = it * 64 + lane, r = idx / (CIN / 4), c4 = idx % (CIN / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < n_in) {
      const int64_t src = rows ? (int64_t)rows[row0 + r] : row0 + r;
      v = *reinterpret_cast<const float4*>(in + src * CIN + c4 * 4);
    }
    float* d = a + r * PITCH + c4 * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float av[CIN / 2];
#pragma unroll
  for (int s = 0; s < CIN / 2; ++s) av[s] = a[i * PITCH + 2 * s + h];

  for (int o = 0; o < 8; ++o) {
    const float* wo = w + (int64_t)o * CIN * cout;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int colb = t * 32 + i;   // the last column's value stands in beyond cout (never stored)
      const float b = bias[FULL || colb < cout ? colb : cout - 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = b;
    }
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int colw = t * 32 + i;
        const float bv = wo[(2 * s + h) * cout + (FULL || colw < cout ? colw : cout - 1)];   // unconditional load
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int64_t p = row0 + row;
        if (p < n_in && (FULL || t * 32 + i < cout)) {
          float v = acc[t][r];
          if (relu) v = fmaxf(v, 0.0f);
          out[(p * 8 + o) * cout + t * 32 + i] = v;
        }
      }
    }
  }
}

template <int CIN>
static void launch_convT(hipStream_t st, const float* d_in, int64_t n_in, const float* d_w, const float* d_bias, int relu,
                         float* d_out, const uint32_t* d_rows, int cout) {
  const dim3 grid(nblk(n_in, 32 * GC_WAVES)), block(GC_WAVES * 64);
#define PCC_CONVT(NT_)                                                                                                        \
  if (cout == 32 * NT_)                                                                                                       \
    hipLaunchKernelGGL((k_convT_mfma<CIN, NT_, true>), grid, block, 0, st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout); \
  else                                                                                                                        \
    hipLaunchKernelGGL((k_convT_mfma<CIN, NT_, false>), grid, block, 0, st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout)
  switch ((cout + 31) / 32) {
    case 1: PCC_CONVT(1); break;
    case 2: PCC_CONVT(2); break;
    case 3: PCC_CONVT(3); break;
    default: PCC_CONVT(4); break;
  }
#undef PCC_CONVT
}
// widths the matrix-core kernel takes: multiples of 16 up to 128
static bool convT_widths(int cin, int cout) { return cin % 16 == 0 && cout % 16 == 0 && cin >= 16 && cin <= 128 && cout >= 16 && cout <= 128; }
static void launch_convT_any(hipStream_t st, const float* d_in, int64_t n_in, const float* d_w, const float* d_bias, int relu,
                             float* d_out, const uint32_t* d_rows, int cin, int cout) {
  switch (cin / 16) {
    case 1: launch_convT<16>(st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout); break;
    case 2: launch_convT<32>(st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout); break;
    case 3: launch_convT<48>(st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout); break;
    case 4: launch_convT<64>(st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout); break;
    case 5: launch_convT<80>(st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout); break;
    case 6: launch_convT<96>(st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout); break;
    case 7: launch_convT<112>(st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout); break;
    default: launch_convT<128>(st, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, cout); break;
  }
}

// The model default's up stage (32 -> 32) on the transposed 16-slot product of conv16.h: a wave takes 16 parents, holds
// their rows as MFMA B operands (two 16-B loads per lane, once), and per octant runs two chains of eight
// v_mfma_f32_16x16x4_f32 on the operand-ordered weights (four coalesced 16-B loads per lane and octant, requested an octant
// ahead), turns the tile through 2 KB of LDS and stores it as whole 128-B rows, 16 B per lane, with raw buffer stores (a
// row of a parent past the end lies beyond the buffer and is dropped: no branch around a store, so the compiler's vmcnt
// counts stay exact — under `if (row exists) store` every octant waited for the previous one's stores to be acknowledged).
// k_convT_mfma<32, 1> wrote the same rows with 4-B stores, sixteen store instructions per octant behind a chain of sixteen
// dependent 32x32x2 MFMAs whose B operands were 4-B loads; chains and stores took turns inside a wave (83 us without the
// stores, 84 us without the chains, 158 us together for the 408k -> 3.26M stage isolated; a plain fill of the same 417 MB
// takes 63 us).  This kernel: 135 us there, 33 us (was 51) for 106k -> 846k.  A variant with the weights in LDS (one copy
// per 256-thread workgroup, 8 tiles per wave, no global load inside a tile but the next tile's rows) was built: the
// same 134 us at the large size, slower at the small ones (few workgroups); its parts alone — 116 us without the
// chains, 100 us without the stores — say the limit is neither the weight loads nor the order of loads and stores.
__global__ __launch_bounds__(64) void k_convT16(
    const float* __restrict__ in, int64_t n_in, const float* __restrict__ wsw, const float* __restrict__ bias, int relu,
    float* __restrict__ out, const uint32_t* __restrict__ rows) {
  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(uint32_t)(n_in * 1024), 0x00027000);
  __shared__ __attribute__((aligned(16))) float tile[2][2][16 * 16];   // [buffer][plane][row][16]: the layout of conv16.h's accumulators
  const int lane = threadIdx.x, n = lane & 15, q = lane >> 4, grow = lane >> 3, chunk = lane & 7;
  const int64_t p0 = (int64_t)blockIdx.x * 16;
  if (p0 >= n_in) return;
  const int64_t p = p0 + n;
  float xv[8];
  {
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
    if (p < n_in) {
      const int64_t src = rows ? (int64_t)rows[p] : p;
      const float4* xp = reinterpret_cast<const float4*>(in + src * 32 + q * 8);
      g0 = xp[0];
      g1 = xp[1];
    }
    unsigned m[2][4] = {{__float_as_uint(g0.x), __float_as_uint(g0.y), __float_as_uint(g0.z), __float_as_uint(g0.w)},
                        {__float_as_uint(g1.x), __float_as_uint(g1.y), __float_as_uint(g1.z), __float_as_uint(g1.w)}};
#pragma unroll
    for (int b = 0; b < 2; ++b) {   // conv16.h, natural rows: 4 x 4 transposes among the q-lanes of a slot
      u32x2 t = __builtin_amdgcn_permlane32_swap(m[b][0], m[b][2], false, false);
      m[b][0] = t[0]; m[b][2] = t[1];
      t = __builtin_amdgcn_permlane32_swap(m[b][1], m[b][3], false, false);
      m[b][1] = t[0]; m[b][3] = t[1];
      t = __builtin_amdgcn_permlane16_swap(m[b][0], m[b][1], false, false);
      m[b][0] = t[0]; m[b][1] = t[1];
      t = __builtin_amdgcn_permlane16_swap(m[b][2], m[b][3], false, false);
      m[b][2] = t[0]; m[b][3] = t[1];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) xv[2 * t4 + b] = __uint_as_float(m[b][t4]);
    }
  }
  const float* bp = bias + 4 * q;
  const f32x4 bl = {bp[0], bp[1], bp[2], bp[3]}, bh = {bp[16], bp[17], bp[18], bp[19]};
  auto load_w = [&](float4 (&W)[4], int o) {
    const float4* wp = reinterpret_cast<const float4*>(wsw + (int64_t)(o < 8 ? o : 7) * 1024 + lane * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) W[j] = wp[j];
  };
  float4 Wa[4], Wb[4];
  load_w(Wa, 0);
  auto octant = [&](int o, const float4 (&W)[4], float4 (&Wn)[4]) {
    load_w(Wn, o + 1);
    const float wl[8] = {W[0].x, W[0].y, W[0].z, W[0].w, W[1].x, W[1].y, W[1].z, W[1].w};
    const float wh[8] = {W[2].x, W[2].y, W[2].z, W[2].w, W[3].x, W[3].y, W[3].z, W[3].w};
    f32x4 lo = bl, hi = bh;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      lo = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xv[s], lo, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xv[s], hi, 0, 0, 0);
    }
    float4 a = make_float4(lo[0], lo[1], lo[2], lo[3]), b = make_float4(hi[0], hi[1], hi[2], hi[3]);
    if (relu) {
      a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
      b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f);
    }
    float (&T)[2][16 * 16] = tile[o & 1];
    // lane (n, q) holds channels 4q .. of row n in plane 0 and 16 + 4q .. in plane 1
    *reinterpret_cast<float4*>(&T[0][n * 16 + 4 * q]) = a;
    *reinterpret_cast<float4*>(&T[1][n * 16 + 4 * q]) = b;
    PCC16_SYNC();
#pragma unroll
    for (int it = 0; it < 2; ++it) {   // lane (grow, chunk): piece `chunk` (4 channels) of rows grow and grow + 8
      const int r = grow + 8 * it;
      const float4 v = *reinterpret_cast<const float4*>(&T[chunk >> 2][r * 16 + 4 * (chunk & 3)]);
      const uint32_t off = (uint32_t)(((p0 + r) * 8 + o) * 128 + 16 * chunk);   // beyond the buffer for a parent past the end
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      __builtin_amdgcn_raw_buffer_store_b128(bits, out_rs, off, 0, 0);
    }
  };
#pragma unroll
  for (int o = 0; o < 8; o += 2) {
    octant(o, Wa, Wb);
    octant(o + 1, Wb, Wa);
  }
}

// k_convT16 as persistent waves: a wave keeps the weights of all 8 octants in registers (128) for its whole life and
// walks tiles of 16 parents, the next tile's rows (and the row index of the one after, when the rows are gathered)
// requested before the current tile's first store.  Nothing the chains wait for is ever younger than a store: in
// k_convT16 the weights of octant o + 2 were requested behind the stores of octant o, and vmcnt retires in order — every
// wave waited for its stores to be acknowledged once per octant.  Weight traffic from L2: 32 KB per wave instead of per
// 16 parents (835 MB -> 64 MB for the 408k-parent stage).
#ifndef PCC_CT16P_WPS
#define PCC_CT16P_WPS 2   // waves per SIMD
#endif
template <bool ROWS, bool RELU>
__global__ __launch_bounds__(64, PCC_CT16P_WPS) void k_convT16p(
    const float* __restrict__ in, int64_t n_in, const float* __restrict__ wsw, const float* __restrict__ bias,
    float* __restrict__ out, const uint32_t* __restrict__ rows, int64_t n_tiles) {
  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(uint32_t)(n_in * 1024), 0x00027000);
  __shared__ __attribute__((aligned(16))) float tile[2][2][16 * 16];
  const int lane = threadIdx.x, n = lane & 15, q = lane >> 4, grow = lane >> 3, chunk = lane & 7;
  float4 W[8][4];
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const float4* wp = reinterpret_cast<const float4*>(wsw + (int64_t)o * 1024 + lane * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) W[o][j] = wp[j];
  }
  const float* bp = bias + 4 * q;
  // the weights have arrived HERE: left to the loop, their wait (vmcnt retires in order) would stand in front of every
  // tile's chains and cover the previous tile's stores
#pragma unroll
  for (int o = 0; o < 8; ++o)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(W[o][j].x), "+v"(W[o][j].y), "+v"(W[o][j].z), "+v"(W[o][j].w));
  const int64_t last = n_in - 1;
  auto parent_of = [&](int64_t t) -> int64_t {   // clamped: a slot past the end reloads the last row, its stores are dropped
    const int64_t p = t * 16 + n;
    return p < n_in ? p : last;
  };
  auto row_index = [&](int64_t t) -> int64_t {   // a tile past the wave's last: the last tile again (loaded, never used)
    const int64_t p = parent_of(t < n_tiles ? t : n_tiles - 1);
    if constexpr (ROWS) return (int64_t)rows[p];
    return p;
  };
  auto load_rows = [&](int64_t src, float4& g0, float4& g1) {
    const float4* xp = reinterpret_cast<const float4*>(in + src * 32 + q * 8);
    g0 = xp[0];
    g1 = xp[1];
  };
  const int64_t stride = gridDim.x;
  int64_t t = blockIdx.x;
  if (t >= n_tiles) return;
  float4 g0, g1;
  load_rows(row_index(t), g0, g1);
  int64_t src_next = row_index(t + stride);
  // nothing pending on entry (the loop header joins this state with the back edge's: a load still in flight here would be
  // waited for at the top of every tile, behind the previous tile's stores)
  asm volatile("" : "+v"(g0.x), "+v"(g0.y), "+v"(g0.z), "+v"(g0.w), "+v"(g1.x), "+v"(g1.y), "+v"(g1.z), "+v"(g1.w), "+v"(src_next));
  float bias8[8] = {bp[0], bp[1], bp[2], bp[3], bp[16], bp[17], bp[18], bp[19]};
  asm volatile("" : "+v"(bias8[0]), "+v"(bias8[1]), "+v"(bias8[2]), "+v"(bias8[3]), "+v"(bias8[4]), "+v"(bias8[5]), "+v"(bias8[6]), "+v"(bias8[7]));
  const f32x4 bl = {bias8[0], bias8[1], bias8[2], bias8[3]}, bh = {bias8[4], bias8[5], bias8[6], bias8[7]};
  for (; t < n_tiles; t += stride) {
    float xv[8];
    {
      unsigned m[2][4] = {{__float_as_uint(g0.x), __float_as_uint(g0.y), __float_as_uint(g0.z), __float_as_uint(g0.w)},
                          {__float_as_uint(g1.x), __float_as_uint(g1.y), __float_as_uint(g1.z), __float_as_uint(g1.w)}};
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        u32x2 w2 = __builtin_amdgcn_permlane32_swap(m[b][0], m[b][2], false, false);
        m[b][0] = w2[0]; m[b][2] = w2[1];
        w2 = __builtin_amdgcn_permlane32_swap(m[b][1], m[b][3], false, false);
        m[b][1] = w2[0]; m[b][3] = w2[1];
        w2 = __builtin_amdgcn_permlane16_swap(m[b][0], m[b][1], false, false);
        m[b][0] = w2[0]; m[b][1] = w2[1];
        w2 = __builtin_amdgcn_permlane16_swap(m[b][2], m[b][3], false, false);
        m[b][2] = w2[0]; m[b][3] = w2[1];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) xv[2 * t4 + b] = __uint_as_float(m[b][t4]);
      }
    }
    // next tile's rows and the index of the one after: in flight before this tile's first store
    load_rows(src_next, g0, g1);
    src_next = row_index(t + 2 * stride);
    const int64_t p0 = t * 16;
    // the rows of octant o leave (LDS read + stores) behind the chains of octant o + 1: the turn through LDS has a whole
    // chain to complete in
    auto store_octant = [&](int o) {
      float (&T)[2][16 * 16] = tile[o & 1];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r = grow + 8 * it;
        const float4 v = *reinterpret_cast<const float4*>(&T[chunk >> 2][r * 16 + 4 * (chunk & 3)]);
        const uint32_t off = (uint32_t)(((p0 + r) * 8 + o) * 128 + 16 * chunk);   // beyond the buffer for a parent past the end
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(bits, out_rs, off, 0, 0);
      }
    };
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const float wl[8] = {W[o][0].x, W[o][0].y, W[o][0].z, W[o][0].w, W[o][1].x, W[o][1].y, W[o][1].z, W[o][1].w};
      const float wh[8] = {W[o][2].x, W[o][2].y, W[o][2].z, W[o][2].w, W[o][3].x, W[o][3].y, W[o][3].z, W[o][3].w};
      f32x4 lo = bl, hi = bh;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        lo = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xv[s], lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xv[s], hi, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (o > 0) store_octant(o - 1);
      float4 a = make_float4(lo[0], lo[1], lo[2], lo[3]), b = make_float4(hi[0], hi[1], hi[2], hi[3]);
      if constexpr (RELU) {
        a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
        b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f);
      }
      float (&T)[2][16 * 16] = tile[o & 1];
      *reinterpret_cast<float4*>(&T[0][n * 16 + 4 * q]) = a;
      *reinterpret_cast<float4*>(&T[1][n * 16 + 4 * q]) = b;
      PCC16_SYNC();
      __builtin_amdgcn_sched_barrier(0);
    }
    store_octant(7);
    __builtin_amdgcn_sched_barrier(0);
  }
}

static int convT16_waves() {   // two waves per SIMD on every CU of the device
  static int n = 0;
  if (n == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    n = cus * 4 * PCC_CT16P_WPS;
  }
  return n;
}

static void launch_convT16(hipStream_t st, const float* d_in, int64_t n_in, const float* wsw, const float* d_bias, int relu,
                           float* d_out, const uint32_t* d_rows) {
  const int64_t n_tiles = (n_in + 15) / 16;
  static const bool one_shot = getenv("PCC_CONVT_ONESHOT") != nullptr;   // A/B switch: the one-tile-per-wave form
  // from 4 tiles per resident wave on (a wave with 4 tiles beside one with 3 leaves a quarter of the machine idle at the
  // end: 106k parents = 3.2 tiles per wave take 36.6 us this way, 33.1 us one tile per wave; 408k: 97 against 135)
  if (one_shot || n_tiles < 4 * (int64_t)convT16_waves()) {
    hipLaunchKernelGGL(k_convT16, dim3((unsigned)n_tiles), dim3(64), 0, st, d_in, n_in, wsw, d_bias, relu, d_out, d_rows);
  } else {
    const dim3 grid((unsigned)convT16_waves());
    if (d_rows) {
      if (relu) hipLaunchKernelGGL((k_convT16p<true, true>), grid, dim3(64), 0, st, d_in, n_in, wsw, d_bias, d_out, d_rows, n_tiles);
      else hipLaunchKernelGGL((k_convT16p<true, false>), grid, dim3(64), 0, st, d_in, n_in, wsw, d_bias, d_out, d_rows, n_tiles);
    } else {
      if (relu) hipLaunchKernelGGL((k_convT16p<false, true>), grid, dim3(64), 0, st, d_in, n_in, wsw, d_bias, d_out, d_rows, n_tiles);
      else hipLaunchKernelGGL((k_convT16p<false, false>), grid, dim3(64), 0, st, d_in, n_in, wsw, d_bias, d_out, d_rows, n_tiles);
    }
  }
}

__global__ __launch_bounds__(256) void k_convT_scalar(
    const float* __restrict__ in, int64_t n_in, const float* __restrict__ w,
    const float* __restrict__ bias, int cin, int cout, int relu, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t orow = t / cout;  // 8p+o
  const int co = (int)(t - orow * cout);
  if (orow >= n_in * 8) return;
  const int64_t p = orow >> 3;
  const int o = (int)(orow & 7);
  float acc = bias[co];
  const float* x = in + p * cin;
  const float* wo = w + ((int64_t)o * cin) * cout + co;
  for (int ci = 0; ci < cin; ++ci) acc = fmaf(x[ci], wo[(int64_t)ci * cout], acc);
  if (relu) acc = fmaxf(acc, 0.0f);
  out[t] = acc;
}

// 1x1: one thread per (row, co)
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ in, int64_t n,
                                                const float* __restrict__ w,
                                                const float* __restrict__ bias, int cin, int cout,
                                                int relu, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t / cout;
  const int co = (int)(t - row * cout);
  if (row >= n) return;
  float acc = bias[co];
  const float* x = in + row * cin;
  for (int ci = 0; ci < cin; ++ci) acc = fmaf(x[ci], w[(int64_t)ci * cout + co], acc);
  if (relu) acc = fmaxf(acc, 0.0f);
  out[t] = acc;
}

// 1x1 with coalesced row reads: a block stages 256 rows (cin*4 B each) through LDS with 16-B
// lane loads, then thread t runs the fmaf chains of row t (pitch cin+1: conflict-free).  The
// thread-per-row form above fetches every 128-B line up to 8 times (PMC: 2.2 GB for 0.42 GB of rows).
// rows (nullable): output row j reads input row rows[j] — the colour head applied to the voxels that survive the last
// pruning, without first gathering their 128-B feature rows into a tensor of their own.
template <int CIN>
__global__ __launch_bounds__(256) void k_linear_rows(const float* __restrict__ in, int64_t n,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias, int cout, int relu,
                                                     float* __restrict__ out,
                                                     const uint32_t* __restrict__ rows = nullptr) {
  constexpr int PITCH = CIN + 1;
  constexpr int VEC = CIN / 4;  // float4 per row
  __shared__ float tile[256 * PITCH];
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  for (int v = threadIdx.x; v < 256 * VEC; v += 256) {
    const int r = v / VEC, c4 = v - r * VEC;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < n) {
      const int64_t src = rows ? (int64_t)rows[row0 + r] : row0 + r;
      x = *reinterpret_cast<const float4*>(in + src * CIN + c4 * 4);
    }
    float* d = tile + r * PITCH + c4 * 4;
    d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
  }
  __syncthreads();
  const int64_t row = row0 + threadIdx.x;
  if (row >= n) return;
  const float* x = tile + threadIdx.x * PITCH;
  for (int co = 0; co < cout; ++co) {
    float acc = bias[co];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc = fmaf(x[ci], w[ci * cout + co], acc);
    if (relu) acc = fmaxf(acc, 0.0f);
    out[row * cout + co] = acc;
  }
}

// PCC_FORCE_SCALAR=1 in the environment routes every layer through the scalar
// kernels (used by the parity tests to check MFMA == scalar on the device).
static bool force_scalar() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PCC_FORCE_SCALAR");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

// True when pcc_sparse_conv_head_up has a kernel to run: always, except under the scalar cross-check switch (the whole-
// GOP decoder then materialises the child rule books: pcc_derive_map_up + pcc_sparse_conv_head)
bool pcc_conv_up_fused() { return !force_scalar(); }

// ---- weights in MFMA operand order (conv16.h).  A layer's [k][32][cout] weights registered with pcc_conv_prepare are
// swizzled once and found again by their device pointer; weights that were not registered are swizzled into the
// call's scratch arena in front of the launch (one small kernel, ~2 us).
struct PccWeightCache {
  struct Entry {
    float* wsw;
    int k_vol, cout;
    int cin = 32;   // 32 with the operand order of conv16.h; any other width (or `gen`): the chunked order of convgen.h
    bool gen = false;
  };
  std::map<const float*, Entry> m;
};

void pcc_wcache_free(pcc_ctx* ctx) {
  if (!ctx || !ctx->wcache) return;
  for (auto& kv : ctx->wcache->m) (void)hipFree(kv.second.wsw);
  delete ctx->wcache;
  ctx->wcache = nullptr;
}

static int swizzle_launch(hipStream_t st, const float* d_w, int k_vol, int cout, float* wsw) {
  hipLaunchKernelGGL(k_conv16_swizzle, dim3(nblk((int64_t)k_vol * cout * 32, 256)), dim3(256), 0, st, d_w, k_vol, cout, wsw);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// ---- any widths that are multiples of 16 (convgen.h)
static int gen_chunk(int cin) { return cin % 32 == 0 ? 32 : 16; }
static size_t gen_wsw_floats(int k_vol, int cin, int cout) { return (size_t)k_vol * cin * ((cout + 31) / 32) * 32; }
static bool convgen_widths(int k_vol, int cin, int cout) {
  return (k_vol == 27 || k_vol == 8) && cin % 16 == 0 && cout % 16 == 0 && cin >= 16 && cin <= 128 && cout >= 16 && cout <= 256;
}
static int gen_swizzle_launch(hipStream_t st, const float* d_w, int k_vol, int cin, int cout, float* wsw) {
  hipLaunchKernelGGL(k_convgen_swizzle, dim3(nblk((int64_t)gen_wsw_floats(k_vol, cin, cout), 256)), dim3(256), 0, st, d_w, k_vol,
                     cin, cout, gen_chunk(cin), wsw);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_conv_prepare(pcc_ctx* ctx, const float* d_w, int k_vol, int cin, int cout) {
  PCC_REQUIRE(ctx && d_w, PCC_E_ARG, "pcc_conv_prepare: null argument");
  const bool tuned = (k_vol == 27 || k_vol == 8) && cin == 32 && (cout == 32 || cout == 64);
  PCC_REQUIRE(tuned || convgen_widths(k_vol, cin, cout), PCC_E_ARG,
              "pcc_conv_prepare: k_vol=%d cin=%d cout=%d has no pre-arranged form", k_vol, cin, cout);
  if (!ctx->wcache) ctx->wcache = new (std::nothrow) PccWeightCache();
  PCC_REQUIRE(ctx->wcache, PCC_E_NOMEM, "pcc_conv_prepare: out of memory");
  auto it = ctx->wcache->m.find(d_w);
  float* wsw = nullptr;
  if (it != ctx->wcache->m.end() && it->second.k_vol == k_vol && it->second.cout == cout && it->second.cin == cin &&
      it->second.gen == !tuned) {
    wsw = it->second.wsw;  // registered before: refresh (the tensor may have new contents)
  } else {
    if (it != ctx->wcache->m.end()) {
      PCC_HIP(hipStreamSynchronize(ctx->stream));
      (void)hipFree(it->second.wsw);
      ctx->wcache->m.erase(it);
    }
    PCC_HIP(hipMalloc((void**)&wsw, tuned ? (size_t)k_vol * cout * 32 * 4 : gen_wsw_floats(k_vol, cin, cout) * 4));
    PccWeightCache::Entry e;
    e.wsw = wsw; e.k_vol = k_vol; e.cout = cout; e.cin = cin; e.gen = !tuned;
    ctx->wcache->m[d_w] = e;
  }
  return tuned ? swizzle_launch(ctx->stream, d_w, k_vol, cout, wsw) : gen_swizzle_launch(ctx->stream, d_w, k_vol, cin, cout, wsw);
}

extern "C" int pcc_conv_forget(pcc_ctx* ctx, const float* d_w) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_conv_forget: null ctx");
  if (!ctx->wcache) return PCC_OK;
  auto it = ctx->wcache->m.find(d_w);
  if (it == ctx->wcache->m.end()) return PCC_OK;
  PCC_HIP(hipStreamSynchronize(ctx->stream));
  (void)hipFree(it->second.wsw);
  ctx->wcache->m.erase(it);
  return PCC_OK;
}

// operand-ordered weights of a launch: the registered copy, or one made now in the arena
static int weights_for(pcc_ctx* ctx, const float* d_w, int k_vol, int cout, const float** wsw) {
  if (ctx->wcache) {
    auto it = ctx->wcache->m.find(d_w);
    if (it != ctx->wcache->m.end() && it->second.k_vol == k_vol && it->second.cout == cout && !it->second.gen) {
      *wsw = it->second.wsw;
      return PCC_OK;
    }
  }
  const size_t bytes = (size_t)k_vol * cout * 32 * 4;
  PCC_TRY(pcc_arena_reserve(ctx, bytes + 512));
  float* tmp = (float*)pcc_arena_alloc(ctx, bytes);
  if (!tmp) return PCC_E_NOMEM;
  PCC_TRY(swizzle_launch(ctx->stream, d_w, k_vol, cout, tmp));
  *wsw = tmp;
  return PCC_OK;
}

// the same for the chunked operand order of convgen.h
static int gen_weights_for(pcc_ctx* ctx, const float* d_w, int k_vol, int cin, int cout, const float** wsw) {
  if (ctx->wcache) {
    auto it = ctx->wcache->m.find(d_w);
    if (it != ctx->wcache->m.end() && it->second.k_vol == k_vol && it->second.cout == cout && it->second.cin == cin &&
        it->second.gen) {
      *wsw = it->second.wsw;
      return PCC_OK;
    }
  }
  const size_t bytes = gen_wsw_floats(k_vol, cin, cout) * 4;
  PCC_TRY(pcc_arena_reserve(ctx, bytes + 512));
  float* tmp = (float*)pcc_arena_alloc(ctx, bytes);
  if (!tmp) return PCC_E_NOMEM;
  PCC_TRY(gen_swizzle_launch(ctx->stream, d_w, k_vol, cin, cout, tmp));
  *wsw = tmp;
  return PCC_OK;
}

// one k_gconv16 launch.  Grid rounded up to a multiple of 8 workgroups: the kernel maps workgroup -> window per XCD.
// Launches of fewer than kSmallLaunchRows rows (well under one round of 64-row windows on the chip's 4096 wave
// slots) take 32-row windows: such a launch lasts as long as one window, and a 32-row window is the shorter one
// (106k rows: 39 us on 64-row windows, 42 on 32-row ones; 26k rows: the other way round; 408k rows, 6.2 windows per SIMD:
// 157 us on 64-row windows, 198 on 32-row ones — that launch is not waiting for a last partial round).
constexpr int64_t kSmallLaunchRows = 100000;
// PCC_CONV_WIDE_ROWS=1 in the environment (read once): every launch takes the 64-bit row arithmetic that tensors of
// 2^25 rows and more need — how the tests reach that form without a 4-GB tensor.
static bool force_wide_rows() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PCC_CONV_WIDE_ROWS");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}
// convrows16.hip
void pcc_rows16_launch(hipStream_t st, int cout, int k_vol, unsigned n_windows, const float* d_in, const int32_t* d_nbr,
                       int64_t pitch, int64_t n_out, const float* wsw, const float* d_bias, int relu, float* d_out,
                       uint32_t in_bytes);
// Launches of at most this many 16-row windows (half as many for 32 -> 64, whose waves carry twice the matrix work) take
// k_gconv_rows16 (convrows16.h); above it the chip is filled and k_gconv16's compaction wins (tools/bench_small_conv.py, one
// box, k_gconv16 -> k_gconv_rows16: 1.6k / 6.6k rows 3^3 32 -> 32 24 -> 14 us, 32 -> 64 24 -> 22; 26k rows 2^3 15.0 -> 11.9,
// 3^3 32 -> 32 25 -> 25 alone and 33.3 -> 26.4 in the step, 3^3 32 -> 64 35.5 -> 41.8 alone and 45.6 -> 47.5 in the step:
// stays on k_gconv16).  PCC_CONV_ROWS16_MAX in the environment (read once) moves the bound: 0 keeps every launch on
// k_gconv16 (the cross-check of the two kernels).
constexpr int64_t kRows16MaxWaves = 2048;
static int64_t rows16_max_waves() {
  static int64_t v = -1;
  if (v < 0) {
    const char* e = getenv("PCC_CONV_ROWS16_MAX");
    v = e && e[0] ? (int64_t)atoll(e) : kRows16MaxWaves;
    if (v < 0) v = 0;
  }
  return v;
}
template <bool HEAD, bool UP, bool PERM, int COUT, bool WIDE>
static void launch16w(hipStream_t st, const float* d_in, const int32_t* d_nbr, int k_vol, int64_t pitch, int64_t n_out,
                      const float* wsw, const float* d_bias, int relu, float* d_out, const float* hw, const float* hb,
                      float* ho, const float* cw, const float* cb, float* co, uint32_t in_bytes) {
  if constexpr (!HEAD && !UP && !PERM && !WIDE) {
    // latent-sized launches on an explicit rule book: 16-row windows without compaction (convrows16.h)
    if ((int64_t)nblk(n_out, 16) * (COUT / 32) <= rows16_max_waves() && (k_vol == 27 || k_vol == 8)) {
      pcc_rows16_launch(st, COUT, k_vol, nblk(n_out, 16), d_in, d_nbr, pitch, n_out, wsw, d_bias, relu, d_out, in_bytes);
      return;
    }
  }
  if (n_out < kSmallLaunchRows)
    hipLaunchKernelGGL((k_gconv16<HEAD, UP, PERM, COUT, 32, WIDE>), dim3((nblk(n_out, 32) + 7) / 8 * 8, COUT / 32), dim3(64), 0,
                       st, d_in, d_nbr, k_vol, pitch, n_out, wsw, d_bias, relu, d_out, hw, hb, ho, cw, cb, co, in_bytes);
  else
    hipLaunchKernelGGL((k_gconv16<HEAD, UP, PERM, COUT, 64, WIDE>), dim3((nblk(n_out, 64) + 7) / 8 * 8, COUT / 32), dim3(64), 0,
                       st, d_in, d_nbr, k_vol, pitch, n_out, wsw, d_bias, relu, d_out, hw, hb, ho, cw, cb, co, in_bytes);
}
// n_in: rows of d_in (the byte offset of a row must fit 32 bits for the narrow form; UP: its parent book's pitch 24)
template <bool HEAD, bool UP, bool PERM, int COUT>
static void launch16(hipStream_t st, const float* d_in, int64_t n_in, const int32_t* d_nbr, int k_vol, int64_t pitch,
                     int64_t n_out, const float* wsw, const float* d_bias, int relu, float* d_out, const float* hw,
                     const float* hb, float* ho, const float* cw = nullptr, const float* cb = nullptr, float* co = nullptr) {
  const bool wide = n_in >= ((int64_t)1 << 25) || (UP && pitch >= ((int64_t)1 << 24)) || force_wide_rows();
  if (wide)
    launch16w<HEAD, UP, PERM, COUT, true>(st, d_in, d_nbr, k_vol, pitch, n_out, wsw, d_bias, relu, d_out, hw, hb, ho, cw, cb, co, 0u);
  else   // n_in * 128 <= 2^32 - 128: the pad offset of conv16.h lies beyond the buffer
    launch16w<HEAD, UP, PERM, COUT, false>(st, d_in, d_nbr, k_vol, pitch, n_out, wsw, d_bias, relu, d_out, hw, hb, ho, cw, cb, co,
                                           (uint32_t)(n_in * 128));
}

// PCC_CONV_UP_LEGACY=1 in the environment (read once): the g_s layers stay on k_gconv16's UP form (two passes over the
// offsets) instead of k_gconv_up — the cross-check of the two kernels and the form tensors of 2^25 rows and more take
static bool force_up_legacy() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PCC_CONV_UP_LEGACY");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}
// one g_s layer on the 8 n_parents children of a level: k_gconv_up (convup.h) when the byte offset of an input row fits
// 32 bits and the parent book's pitch 24, else k_gconv16's UP form
template <bool PERM>
static void launch_up(hipStream_t st, const float* d_in, int64_t n_parents, const int32_t* d_nbr_parent, int64_t pitch,
                      const float* wsw, const float* d_bias, int relu, float* d_out, const float* hw, const float* hb,
                      float* ho, const float* cw, const float* cb, float* co) {
  const int64_t n_out = 8 * n_parents;
  if (n_out < ((int64_t)1 << 25) && pitch < ((int64_t)1 << 24) && !force_wide_rows() && !force_up_legacy()) {
    const dim3 up_grid((nblk(n_parents, 16) + 7) / 8 * 8);
    hipLaunchKernelGGL((k_gconv_up<PERM>), up_grid, dim3(64), 0, st, d_in, d_nbr_parent, pitch,
                       n_parents, wsw, d_bias, relu, d_out, hw, hb, ho, cw, cb, co, (uint32_t)(n_out * 128));
  } else {
    launch16<true, true, PERM, 32>(st, d_in, n_out, d_nbr_parent, 27, pitch, n_out, wsw, d_bias, relu, d_out, hw, hb, ho, cw, cb, co);
  }
}

// PCC_CONVT_LEGACY=1 in the environment (read once): the 32 -> 32 up stages stay on k_convT_mfma<32, 1> (cross-check)
static bool convT_legacy() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PCC_CONVT_LEGACY");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

static bool conv16_shape(const float* d_in, const float* d_out, int k_vol, int cin, int cout) {
  return !force_scalar() && (uintptr_t)d_in % 16 == 0 && (uintptr_t)d_out % 16 == 0 && cin == 32 &&
         (cout == 32 || cout == 64) && (k_vol == 27 || k_vol == 8);
}

static int sparse_conv_impl(pcc_ctx* ctx, const float* d_in, int64_t n_in, const int32_t* d_nbr,
                            int k_vol, int64_t nbr_pitch, int64_t n_out, const float* d_w,
                            const float* d_bias, int cin, int cout, int relu, float* d_out, bool sib) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_sparse_conv: null ctx");
  PCC_REQUIRE(k_vol == 27 || k_vol == 8 || k_vol == 1, PCC_E_ARG, "pcc_sparse_conv: k_vol=%d", k_vol);
  PCC_REQUIRE(cin >= 1 && cin <= 128 && cout >= 1 && cout <= 256, PCC_E_ARG,
              "pcc_sparse_conv: cin=%d cout=%d unsupported", cin, cout);
  if (n_out <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_nbr && d_w && d_bias && d_out && nbr_pitch >= n_out && n_in > 0, PCC_E_ARG,
              "pcc_sparse_conv: bad buffers (pitch %lld, n_out %lld, n_in %lld)", (long long)nbr_pitch,
              (long long)n_out, (long long)n_in);
  hipStream_t st = ctx->stream;
  const float* nof = nullptr;
  float* nofo = nullptr;
  if (!sib && conv16_shape(d_in, d_out, k_vol, cin, cout)) {
    const float* wsw;
    PCC_TRY(weights_for(ctx, d_w, k_vol, cout, &wsw));
    PccProfScope prof(ctx, "sparse_conv", n_out, cin, cout, k_vol);
    if (cout == 32)
      launch16<false, false, false, 32>(st, d_in, n_in, d_nbr, k_vol, nbr_pitch, n_out, wsw, d_bias, relu, d_out, nof, nof, nofo);
    else
      launch16<false, false, false, 64>(st, d_in, n_in, d_nbr, k_vol, nbr_pitch, n_out, wsw, d_bias, relu, d_out, nof, nof, nofo);
  } else if (!force_scalar() && (uintptr_t)d_in % 16 == 0 && (uintptr_t)d_out % 16 == 0 && convgen_widths(k_vol, cin, cout)) {
    // widths other than the model default's: the chunked kernel (convgen.h); with sib, also 32 -> 32 on an explicit book
    const float* wsw;
    PCC_TRY(gen_weights_for(ctx, d_w, k_vol, cin, cout, &wsw));
    PccProfScope prof(ctx, "sparse_conv", n_out, cin, cout, k_vol);
    const dim3 grid((nblk(n_out, 64) + 7) / 8 * 8, (unsigned)((cout + 31) / 32));
    if (gen_chunk(cin) == 32)
      hipLaunchKernelGGL((k_gconv_gen<32>), grid, dim3(64), 0, st, d_in, d_nbr, k_vol, nbr_pitch, n_out, wsw, d_bias, cin, cout,
                         relu, sib ? 1 : 0, d_out);
    else
      hipLaunchKernelGGL((k_gconv_gen<16>), grid, dim3(64), 0, st, d_in, d_nbr, k_vol, nbr_pitch, n_out, wsw, d_bias, cin, cout,
                         relu, sib ? 1 : 0, d_out);
  } else if (!sib && !force_scalar() && (uintptr_t)d_in % 16 == 0 && (uintptr_t)d_out % 16 == 0 && cin == 4 &&
             cout % 16 == 0 && cout <= 128) {
    PccProfScope prof(ctx, "sparse_conv", n_out, cin, cout, k_vol);
    const unsigned full = nblk(n_out, 32 * GC_WAVES), resident = (unsigned)convT16_waves() / 2;   // four workgroups per CU
    const dim3 grid(resident < full ? resident : full), block(GC_WAVES * 64);
#define PCC_FIRST(NT_)                                                                                                          \
  if (cout == 32 * NT_)                                                                                                         \
    hipLaunchKernelGGL((k_gconv_first<NT_, true>), grid, block, 0, st, d_in, d_nbr, k_vol, nbr_pitch, n_out, d_w, d_bias, relu, \
                       d_out, cout);                                                                                            \
  else                                                                                                                          \
    hipLaunchKernelGGL((k_gconv_first<NT_, false>), grid, block, 0, st, d_in, d_nbr, k_vol, nbr_pitch, n_out, d_w, d_bias, relu, \
                       d_out, cout)
    switch ((cout + 31) / 32) {
      case 1: PCC_FIRST(1); break;
      case 2: PCC_FIRST(2); break;
      case 3: PCC_FIRST(3); break;
      default: PCC_FIRST(4); break;
    }
#undef PCC_FIRST
  } else {
    PccProfScope prof(ctx, "sparse_conv", n_out, cin, cout, k_vol);
    hipLaunchKernelGGL(k_gconv_scalar, dim3(nblk(n_out * cout, 256)), dim3(256), 0, st, d_in, d_nbr,
                       k_vol, nbr_pitch, n_out, d_w, d_bias, cin, cout, relu, d_out, sib ? 1 : 0);
  }
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_sparse_conv(pcc_ctx* ctx, const float* d_in, int64_t n_in, const int32_t* d_nbr,
                               int k_vol, int64_t nbr_pitch, int64_t n_out, const float* d_w,
                               const float* d_bias, int cin, int cout, int relu, float* d_out) {
  return sparse_conv_impl(ctx, d_in, n_in, d_nbr, k_vol, nbr_pitch, n_out, d_w, d_bias, cin, cout, relu, d_out, false);
}

extern "C" int pcc_sparse_conv_head(pcc_ctx* ctx, const float* d_in, int64_t n_in, const int32_t* d_nbr,
                                    int k_vol, int64_t nbr_pitch, int64_t n_out, const float* d_w,
                                    const float* d_bias, int cin, int cout, int relu, float* d_out,
                                    const float* d_head_w, const float* d_head_b, float* d_head_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_sparse_conv_head: null ctx");
  PCC_REQUIRE(d_head_w && d_head_b && d_head_out, PCC_E_ARG, "pcc_sparse_conv_head: null head buffers");
  // "siblings first" is DEFINED on row indexes: a neighbour belongs to the first pass when (input row >> 3) ==
  // (output row >> 3) — "same parent" on a generative level (rows in aligned blocks of the 8 children of a parent), and
  // simply "same aligned block of 8 rows" on any other set, for which the order is as deterministic and is what
  // oracle/pcc_oracle.c computes too (include/pcc.h) — also when input and output are different sets (the comparison is
  // of index values; tests/test_gpu_fullsize.py gathers from a 16.8M-row input into 250k output rows that way).
  if (n_out > 0 && cout == 32 && d_in && d_nbr && d_w && d_bias && d_out && nbr_pitch >= n_out && n_in > 0 &&
      conv16_shape(d_in, d_out, k_vol, cin, cout)) {
    const float* wsw;
    PCC_TRY(weights_for(ctx, d_w, k_vol, cout, &wsw));
    PccProfScope prof(ctx, "sparse_conv", n_out, cin, cout, k_vol);
    launch16<true, false, false, 32>(ctx->stream, d_in, n_in, d_nbr, k_vol, nbr_pitch, n_out, wsw, d_bias, relu, d_out, d_head_w,
                                     d_head_b, d_head_out);
    PCC_CHECK_LAUNCH();
    return PCC_OK;
  }
  // generic shapes: the two layers one after the other (same bits; the conv on the scalar kernel, siblings first)
  PCC_TRY(sparse_conv_impl(ctx, d_in, n_in, d_nbr, k_vol, nbr_pitch, n_out, d_w, d_bias, cin, cout, relu, d_out, true));
  return pcc_linear(ctx, d_out, n_out, d_head_w, d_head_b, cout, 1, 0, d_head_out);
}

static int head_up_impl(pcc_ctx* ctx, const float* d_in, int64_t n_parents, const int32_t* d_nbr_parent,
                        int64_t parent_pitch, const float* d_w, const float* d_bias, int relu, float* d_out,
                        const float* d_head_w, const float* d_head_b, float* d_head_out, bool in_perm) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_sparse_conv_head_up: null ctx");
  if (n_parents <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_nbr_parent && d_w && d_bias && d_out && d_head_w && d_head_b && d_head_out &&
                  parent_pitch >= n_parents && n_parents < ((int64_t)1 << 27),
              PCC_E_ARG, "pcc_sparse_conv_head_up: bad buffers (pitch %lld, parents %lld)", (long long)parent_pitch,
              (long long)n_parents);
  PCC_REQUIRE((uintptr_t)d_in % 16 == 0 && (uintptr_t)d_out % 16 == 0, PCC_E_ARG,
              "pcc_sparse_conv_head_up: feature rows must be 16-byte aligned");
  PCC_REQUIRE(pcc_conv_up_fused(), PCC_E_ARG,
              "pcc_sparse_conv_head_up: only the MFMA kernel has this form (PCC_FORCE_SCALAR selects the explicit rule "
              "book: pcc_derive_map_up + pcc_sparse_conv_head)");
  const int64_t n_out = 8 * n_parents;
  const float* wsw;
  PCC_TRY(weights_for(ctx, d_w, 27, 32, &wsw));
  PccProfScope prof(ctx, "sparse_conv", n_out, 32, 32, 27);
  if (in_perm)
    launch_up<true>(ctx->stream, d_in, n_parents, d_nbr_parent, parent_pitch, wsw, d_bias, relu, d_out, d_head_w, d_head_b,
                    d_head_out, nullptr, nullptr, nullptr);
  else
    launch_up<false>(ctx->stream, d_in, n_parents, d_nbr_parent, parent_pitch, wsw, d_bias, relu, d_out, d_head_w, d_head_b,
                     d_head_out, nullptr, nullptr, nullptr);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_sparse_conv_head_up(pcc_ctx* ctx, const float* d_in, int64_t n_parents, const int32_t* d_nbr_parent,
                                       int64_t parent_pitch, const float* d_w, const float* d_bias, int relu,
                                       float* d_out, const float* d_head_w, const float* d_head_b,
                                       float* d_head_out) {
  return head_up_impl(ctx, d_in, n_parents, d_nbr_parent, parent_pitch, d_w, d_bias, relu, d_out, d_head_w, d_head_b,
                      d_head_out, false);
}

// internal (common.h): the same on input rows stored in the channel order of PCC_CONV16_PERM (conv16.h) — what the
// whole-GOP decoder's generative up stages write (codec.hip)
int pcc_sparse_conv_head_up_perm(pcc_ctx* ctx, const float* d_in, int64_t n_parents, const int32_t* d_nbr_parent,
                                 int64_t parent_pitch, const float* d_w, const float* d_bias, int relu, float* d_out,
                                 const float* d_head_w, const float* d_head_b, float* d_head_out) {
  return head_up_impl(ctx, d_in, n_parents, d_nbr_parent, parent_pitch, d_w, d_bias, relu, d_out, d_head_w, d_head_b,
                      d_head_out, true);
}

// internal (common.h): the last stage of g_s — the same layer with the 32 -> 3 colour head evaluated on every candidate
// row as well (bit-identical to pcc_linear on the stored rows) and the rows themselves NOT stored: after the pruning
// only their occupancy logit and their colour are ever read, so 128 B per candidate row stay out of HBM.
int pcc_sparse_conv_head_up_perm_rgb(pcc_ctx* ctx, const float* d_in, int64_t n_parents, const int32_t* d_nbr_parent,
                                     int64_t parent_pitch, const float* d_w, const float* d_bias, int relu,
                                     const float* d_head_w, const float* d_head_b, float* d_head_out,
                                     const float* d_rgb_w, const float* d_rgb_b, float* d_rgb_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_sparse_conv_head_up_perm_rgb: null ctx");
  if (n_parents <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_nbr_parent && d_w && d_bias && d_head_w && d_head_b && d_head_out && d_rgb_w && d_rgb_b &&
                  d_rgb_out && parent_pitch >= n_parents && n_parents < ((int64_t)1 << 27) && (uintptr_t)d_in % 16 == 0,
              PCC_E_ARG, "pcc_sparse_conv_head_up_perm_rgb: bad buffers (pitch %lld, parents %lld)",
              (long long)parent_pitch, (long long)n_parents);
  PCC_REQUIRE(pcc_conv_up_fused(), PCC_E_ARG, "pcc_sparse_conv_head_up_perm_rgb: only the MFMA kernel has this form");
  const int64_t n_out = 8 * n_parents;
  const float* wsw;
  PCC_TRY(weights_for(ctx, d_w, 27, 32, &wsw));
  PccProfScope prof(ctx, "sparse_conv", n_out, 32, 32, 27);
  launch_up<true>(ctx->stream, d_in, n_parents, d_nbr_parent, parent_pitch, wsw, d_bias, relu, nullptr, d_head_w, d_head_b,
                  d_head_out, d_rgb_w, d_rgb_b, d_rgb_out);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

// which kernel the dispatchers above pick for a shape (16-byte aligned tensors assumed): the tests assert that the widths
// a model config may name stay on the matrix cores
extern "C" const char* pcc_conv_kernel_name(int op, int k_vol, int cin, int cout) {
  if (force_scalar()) return op == 2 ? "k_convT_scalar" : "k_gconv_scalar";
  if (op == 2 && cin == 32 && cout == 32 && !convT_legacy()) return "k_convT16";
  if (op == 2) return convT_widths(cin, cout) ? "k_convT_mfma" : "k_convT_scalar";
  const bool sib = op == 1;   // pcc_sparse_conv_head: conv + 1-channel head, siblings first
  if (sib && cin == 32 && cout == 32 && (k_vol == 27 || k_vol == 8)) return "k_gconv16";   // its fused HEAD form
  if (!sib && cin == 32 && (cout == 32 || cout == 64) && (k_vol == 27 || k_vol == 8)) return "k_gconv16";
  if (convgen_widths(k_vol, cin, cout)) return "k_gconv_gen";
  if (!sib && cin == 4 && cout % 16 == 0 && cout <= 128) return "k_gconv_first";
  return "k_gconv_scalar";
}

extern "C" int pcc_convT_gen(pcc_ctx* ctx, const float* d_in, int64_t n_in, const float* d_w,
                             const float* d_bias, int cin, int cout, int relu, float* d_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_convT_gen: null ctx");
  PCC_REQUIRE(cin >= 1 && cin <= 128 && cout >= 1 && cout <= 256, PCC_E_ARG,
              "pcc_convT_gen: cin=%d cout=%d unsupported", cin, cout);
  if (n_in <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_w && d_bias && d_out, PCC_E_ARG, "pcc_convT_gen: null buffers");
  hipStream_t st = ctx->stream;
  PccProfScope prof(ctx, "convT_gen", n_in, cin, cout, 8);
  const bool aligned = ((uintptr_t)d_in % 16 == 0);
  if (!force_scalar() && aligned && cin == 32 && cout == 32 && (uintptr_t)d_out % 16 == 0 && n_in < ((int64_t)1 << 22) &&
      !convT_legacy()) {
    const float* wsw;
    PCC_TRY(weights_for(ctx, d_w, 8, 32, &wsw));
    launch_convT16(st, d_in, n_in, wsw, d_bias, relu, d_out, nullptr);
  } else if (!force_scalar() && aligned && convT_widths(cin, cout)) {
    launch_convT_any(st, d_in, n_in, d_w, d_bias, relu, d_out, nullptr, cin, cout);
  } else {
    hipLaunchKernelGGL(k_convT_scalar, dim3(nblk(n_in * 8 * cout, 256)), dim3(256), 0, st, d_in, n_in,
                       d_w, d_bias, cin, cout, relu, d_out);
  }
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_linear(pcc_ctx* ctx, const float* d_in, int64_t n, const float* d_w,
                          const float* d_bias, int cin, int cout, int relu, float* d_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_linear: null ctx");
  PCC_REQUIRE(cin >= 1 && cin <= 256 && cout >= 1 && cout <= 256, PCC_E_ARG,
              "pcc_linear: cin=%d cout=%d unsupported", cin, cout);
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_w && d_bias && d_out, PCC_E_ARG, "pcc_linear: null buffers");
  PccProfScope prof(ctx, "linear", n, cin, cout, 1);
  if (!force_scalar() && cin == 32 && cout <= 8 && ((uintptr_t)d_in % 16 == 0)) {
    hipLaunchKernelGGL((k_linear_rows<32>), dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_in, n, d_w,
                       d_bias, cout, relu, d_out, (const uint32_t*)nullptr);
  } else {
    hipLaunchKernelGGL(k_linear, dim3(nblk(n * cout, 256)), dim3(256), 0, ctx->stream, d_in, n, d_w,
                       d_bias, cin, cout, relu, d_out);
  }
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_convT_gen_gather(pcc_ctx* ctx, const float* d_in, const uint32_t* d_rows, int64_t n_in,
                                    const float* d_w, const float* d_bias, int relu, float* d_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_convT_gen_gather: null ctx");
  if (n_in <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_rows && d_w && d_bias && d_out && (uintptr_t)d_in % 16 == 0, PCC_E_ARG,
              "pcc_convT_gen_gather: null or misaligned buffers");
  PccProfScope prof(ctx, "convT_gen", n_in, 32, 32, 8);
  if ((uintptr_t)d_out % 16 == 0 && n_in < ((int64_t)1 << 22) && !convT_legacy()) {
    const float* wsw;
    PCC_TRY(weights_for(ctx, d_w, 8, 32, &wsw));
    launch_convT16(ctx->stream, d_in, n_in, wsw, d_bias, relu, d_out, d_rows);
  } else {
    launch_convT<32>(ctx->stream, d_in, n_in, d_w, d_bias, relu, d_out, d_rows, 32);
  }
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_linear_gather(pcc_ctx* ctx, const float* d_in, const uint32_t* d_rows, int64_t n, const float* d_w,
                                 const float* d_bias, int cout, int relu, float* d_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_linear_gather: null ctx");
  PCC_REQUIRE(cout >= 1 && cout <= 8, PCC_E_ARG, "pcc_linear_gather: cout=%d (1..8; cin is 32)", cout);
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_rows && d_w && d_bias && d_out && (uintptr_t)d_in % 16 == 0, PCC_E_ARG,
              "pcc_linear_gather: null or misaligned buffers");
  PccProfScope prof(ctx, "linear", n, 32, cout, 1);
  hipLaunchKernelGGL((k_linear_rows<32>), dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_in, n, d_w, d_bias, cout,
                     relu, d_out, d_rows);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

#if PCC_CONV_STAMP
// diagnostic builds only: the per-wave phase sums of the last k_gconv16 launch
extern "C" int pcc_debug_stamps(unsigned long long* h_out, int n) {
  if (n > 4096 * PCC_NSTAMP) n = 4096 * PCC_NSTAMP;
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(pcc_stamp_buf), (size_t)n * 8) == hipSuccess ? 0 : -2;
}
#endif
