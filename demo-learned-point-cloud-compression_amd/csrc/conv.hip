// conv.hip — sparse network layers: gather-convolution (3^3 stride 1 and 2^3
// stride 2 share one out-stationary kernel), generative transposed
// convolution, 1x1 linear.
//
// Replaces MinkowskiEngine's convolution / generative transposed convolution /
// linear forward kernels executed inside model.g_a, model.g_s, h_a, h_s
// (codec_pipeline.py:273,287,354; codec_parallel.py:302-303,376,469).
//
// Arithmetic contract (pcc.h): out = bias, then for k ascending over PRESENT
// neighbours, ci ascending: out = fmaf(x, w, out).  v_mfma_f32_32x32x2_f32 is
// bit-for-bit that chain (2 ci per instruction, k-ordered), so the MFMA kernel
// and the scalar-fmaf kernel give identical bits, and both equal the C oracle.
// No atomics: a wave owns 32 output rows (out-stationary), so results do not
// depend on scheduling.  An offset whose neighbour is absent for all 32 rows of
// the tile is skipped (wave-uniform ballot); rows are Morton-sorted, so a tile
// is spatially compact and on surface data most of the 27 offsets are skipped.
//
// Tile per wave: 32 rows x COUT(32|64) columns, accumulators in registers
// (16 | 32 VGPRs).  Neighbour rows are gathered with 16-B lane loads (8 lanes
// per 128-B row, coalesced per row) into a wave-private LDS tile with pitch
// CIN+1 floats, which makes both the ds_write_b32 pattern (bank = r + 4*chunk
// + j) and the MFMA-operand ds_read_b32 pattern (bank = i + 2s + h)
// conflict-free.  Weights W[k] (4-8 KB) are read straight from L1/L2 in the
// B-operand layout (two coalesced 128-B rows per load).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

#define GC_WAVES 4

template <int CIN, int NT>
__global__ __launch_bounds__(GC_WAVES * 64) void k_gconv_mfma(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int k_vol, int64_t pitch,
    int64_t n_out, const float* __restrict__ w, const float* __restrict__ bias, int relu,
    float* __restrict__ out) {
  constexpr int COUT = NT * 32;
  constexpr int PITCH = CIN + 1;
  __shared__ float a_lds[GC_WAVES][32 * PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * GC_WAVES + wave) * 32;
  if (row0 >= n_out) return;  // wave-uniform
  const int i = lane & 31, h = lane >> 5;
  float* a = a_lds[wave];

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float b = bias[t * 32 + i];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = b;
  }

  const bool row_ok = (row0 + i) < n_out;
  if constexpr (CIN == 4) {
    // The 4 -> 32 input layer (1M rows, 16-B feature rows): HBM-bound, and with one neighbour-index load and one
    // gather per offset in sequence every offset paid two dependent memory latencies.  All 27 neighbour indices of the
    // tile are fetched at once, and the rows and weights of offset k+1 are in flight while offset k is contracted.
    static_assert(NT == 1, "the 4-channel layer has 32 outputs");
    int32_t nbs[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) nbs[k] = (row_ok && k < k_vol) ? nbr[(int64_t)k * pitch + row0 + i] : -1;
    auto rows_of = [&](int32_t nb) -> float4 {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nb >= 0 && lane < 32) v = *reinterpret_cast<const float4*>(in + (int64_t)nb * CIN);
      return v;
    };
    float4 gn = rows_of(nbs[0]);
    float wn0 = w[(0 + h) * COUT + i], wn1 = w[(2 + h) * COUT + i];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const float4 gc = gn;
      const float wc0 = wn0, wc1 = wn1;
      if (k + 1 < 27) {
        gn = rows_of(nbs[k + 1]);  // -1 past k_vol: no load
        const int kn = k + 1 < k_vol ? k + 1 : k_vol - 1;
        wn0 = w[((int64_t)kn * CIN + 0 + h) * COUT + i];
        wn1 = w[((int64_t)kn * CIN + 2 + h) * COUT + i];
      }
      if (k < k_vol && __ballot(nbs[k] >= 0) != 0ull) {  // uniform: somebody in this tile has offset k
        if (lane < 32) {
          float* d = a + lane * PITCH;
          d[0] = gc.x; d[1] = gc.y; d[2] = gc.z; d[3] = gc.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * PITCH + 0 + h], wc0, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * PITCH + 2 + h], wc1, acc[0], 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else
  for (int k = 0; k < k_vol; ++k) {
    const int32_t nb = row_ok ? nbr[(int64_t)k * pitch + row0 + i] : -1;
    if (__ballot(nb >= 0) == 0ull) continue;  // nobody in this tile has offset k

    // ---- stage the gathered A tile (32 rows x CIN) into LDS
    if constexpr (CIN == 32) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), chunk = lane & 7;
        const int32_t src = __shfl(nb, r, 64);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src >= 0) v = *reinterpret_cast<const float4*>(in + (int64_t)src * CIN + chunk * 4);
        float* d = a + r * PITCH + chunk * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    } else {
      static_assert(CIN == 4, "CIN must be 4 or 32");
      if (lane < 32) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nb >= 0) v = *reinterpret_cast<const float4*>(in + (int64_t)nb * CIN);
        float* d = a + lane * PITCH;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- channel contraction on the matrix cores: 2 input channels / MFMA
    const float* wk = w + (int64_t)k * CIN * COUT;
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) {
      const float av = a[i * PITCH + 2 * s + h];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float bv = wk[(2 * s + h) * COUT + t * 32 + i];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int64_t g = row0 + row;
      if (g < n_out) {
        float v = acc[t][r];
        if (relu) v = fmaxf(v, 0.0f);
        out[g * COUT + t * 32 + i] = v;
      }
    }
  }
}

// Software-pipelined form for CIN = 32 (the kernel the roofline figure is quoted on).
//  - the tile's 27 (or 8) neighbour indices are read once, up front, into a wave-private LDS
//    table and reduced to a wave-uniform bit mask of the offsets present in the tile; the main
//    loop walks the set bits, so absent offsets cost nothing and no index load sits in the loop;
//  - while the 16*NT MFMAs of offset k run, the gathered rows (4 x dwordx4 per lane) and the
//    weight fragments (16*NT dwords per lane) of the NEXT present offset are already in flight
//    into registers; they are written to LDS / consumed at the top of the next iteration.
// Same arithmetic order as the simple form, hence the same bits.
// HEAD: additionally emits head_out[row] = head_b + sum_c fmaf(out[row][c], head_w[c]) (c ascending),
// the 1x1 occupancy logit of g_s, from the tile while it is still on chip (saves re-reading the
// whole feature tensor; same bits as pcc_linear on the stored output).
template <int NT, bool HEAD>
__global__ __launch_bounds__(GC_WAVES * 64) void k_gconv_mfma_pipe(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int k_vol, int64_t pitch,
    int64_t n_out, const float* __restrict__ w, const float* __restrict__ bias, int relu,
    float* __restrict__ out, const float* __restrict__ head_w, const float* __restrict__ head_b,
    float* __restrict__ head_out) {
  constexpr int CIN = 32;
  constexpr int COUT = NT * 32;
  constexpr int PITCH = CIN + 1;
  __shared__ float a_lds[GC_WAVES][32 * PITCH];
  __shared__ int32_t nb_lds[GC_WAVES][27 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * GC_WAVES + wave) * 32;
  if (row0 >= n_out) return;  // wave-uniform
  const int i = lane & 31, h = lane >> 5;
  float* a = a_lds[wave];
  int32_t* nbs = nb_lds[wave];

  // ---- neighbour table of the tile + mask of present offsets
  const bool row_ok = (row0 + i) < n_out;
  uint32_t present = 0;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    int32_t v = -1;
    if (k < k_vol && row_ok) v = nbr[(int64_t)k * pitch + row0 + i];
    if (h == 0) nbs[k * 32 + i] = v;
    if (__ballot(v >= 0) != 0ull) present |= 1u << k;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float b = bias[t * 32 + i];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = b;
  }

  const int grow = lane >> 3, chunk = lane & 7;  // this lane gathers rows grow + 8*it, 16-B chunk `chunk`
  float4 g[4];
  float bw[NT][CIN / 2];

  auto issue = [&](int k) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int32_t src = nbs[k * 32 + it * 8 + grow];
      g[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (src >= 0) g[it] = *reinterpret_cast<const float4*>(in + (int64_t)src * CIN + chunk * 4);
    }
    const float* wk = w + (int64_t)k * CIN * COUT;
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) bw[t][s] = wk[(2 * s + h) * COUT + t * 32 + i];
  };

  uint32_t todo = present;
  if (todo) issue(__builtin_ctz(todo));
  while (todo) {
    todo &= todo - 1;
    // ---- land the prefetched rows in LDS, keep the weight fragments of this offset
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float* d = a + (it * 8 + grow) * PITCH + chunk * 4;
      d[0] = g[it].x; d[1] = g[it].y; d[2] = g[it].z; d[3] = g[it].w;
    }
    float bc[NT][CIN / 2];
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) bc[t][s] = bw[t][s];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- next present offset: rows and weights go in flight now
    if (todo) issue(__builtin_ctz(todo));
    // ---- contraction of the current offset
    float av[CIN / 2];
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) av[s] = a[i * PITCH + 2 * s + h];
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bc[t][s], acc[t], 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int64_t gr = row0 + row;
      float v = acc[t][r];
      if (relu) v = fmaxf(v, 0.0f);
      if (gr < n_out) out[gr * COUT + t * 32 + i] = v;
      if constexpr (HEAD && NT == 1) a[row * PITCH + i] = v;  // the A tile buffer is free now
    }
  }
  if constexpr (HEAD && NT == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 32 && row0 + lane < n_out) {
      float hv = head_b[0];
#pragma unroll
      for (int c = 0; c < 32; ++c) hv = fmaf(a[lane * PITCH + c], head_w[c], hv);
      head_out[row0 + lane] = hv;
    }
  }
}

#include "conv_compact.h"
#include "conv16.h"

// scalar-fmaf reference path on the GPU (any cin/cout), same bits as the MFMA path
__global__ __launch_bounds__(256) void k_gconv_scalar(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int k_vol, int64_t pitch,
    int64_t n_out, const float* __restrict__ w, const float* __restrict__ bias, int cin, int cout,
    int relu, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t / cout;
  const int co = (int)(t - row * cout);
  if (row >= n_out) return;
  float acc = bias[co];
  for (int k = 0; k < k_vol; ++k) {
    const int32_t nb = nbr[(int64_t)k * pitch + row];
    if (nb < 0) continue;
    const float* x = in + (int64_t)nb * cin;
    const float* wk = w + ((int64_t)k * cin) * cout + co;
    for (int ci = 0; ci < cin; ++ci) acc = fmaf(x[ci], wk[(int64_t)ci * cout], acc);
  }
  if (relu) acc = fmaxf(acc, 0.0f);
  out[t] = acc;
}

// generative transposed convolution, kernel 2 stride 2: out[8p+o] = W[o]^T in[p] + b
// rows (nullable): parent p reads input row rows[p] — the up stage right after a pruning, on the kept rows in place
template <int NT>
__global__ __launch_bounds__(GC_WAVES * 64) void k_convT_mfma(
    const float* __restrict__ in, int64_t n_in, const float* __restrict__ w,
    const float* __restrict__ bias, int relu, float* __restrict__ out,
    const uint32_t* __restrict__ rows = nullptr) {
  constexpr int CIN = 32;
  constexpr int COUT = NT * 32;
  constexpr int PITCH = CIN + 1;
  __shared__ float a_lds[GC_WAVES][32 * PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * GC_WAVES + wave) * 32;
  if (row0 >= n_in) return;
  const int i = lane & 31, h = lane >> 5;
  float* a = a_lds[wave];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = it * 8 + (lane >> 3), chunk = lane & 7;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < n_in) {
      const int64_t src = rows ? (int64_t)rows[row0 + r] : row0 + r;
      v = *reinterpret_cast<const float4*>(in + src * CIN + chunk * 4);
    }
    float* d = a + r * PITCH + chunk * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float av[CIN / 2];
#pragma unroll
  for (int s = 0; s < CIN / 2; ++s) av[s] = a[i * PITCH + 2 * s + h];

  for (int o = 0; o < 8; ++o) {
    const float* wo = w + (int64_t)o * CIN * COUT;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float b = bias[t * 32 + i];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = b;
    }
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float bv = wo[(2 * s + h) * COUT + t * 32 + i];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int64_t p = row0 + row;
        if (p < n_in) {
          float v = acc[t][r];
          if (relu) v = fmaxf(v, 0.0f);
          out[(p * 8 + o) * COUT + t * 32 + i] = v;
        }
      }
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

// PCC_CONV_COMPACT = 0 | 64 | 128: rows per wave of the row-compacting 32->32 kernel (0 = dense tiles)
static int compact_rows() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PCC_CONV_COMPACT");
    v = e ? atoi(e) : 64;
    if (v != 0 && v != 64 && v != 128) v = 64;
  }
  return v;
}

static bool conv_simple() {
  static const bool simple = [] { const char* e = getenv("PCC_CONV_SIMPLE"); return e && e[0] == '1'; }();
  return simple;
}

bool pcc_conv_up_fused() {
  static const bool off = [] { const char* e = getenv("PCC_CONV_UP"); return e && e[0] == '0'; }();
  return !off && !force_scalar() && !conv_simple() && compact_rows() != 0;
}

// Layers of >= 200k rows run four 64-row windows per workgroup with the weights shared through LDS (conv_compact.h:
// -3 % on the 3.26M-row layer); smaller launches are one round of windows, where the per-offset workgroup barrier
// only lengthens the critical path.  PCC_CONV_W4=0 keeps every layer on the one-window-per-workgroup kernel, =1 forces
// the shared form for every size (tests).
static bool conv_w4(int64_t n_out) {
  static const int mode = [] { const char* e = getenv("PCC_CONV_W4"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
  return mode < 0 ? n_out >= 200000 : mode == 1;
}

// Launches of at most 64k rows use 32-row windows: such a launch is one partial round of windows and lasts as long as
// ONE window's 27 dependent offset steps, which are shorter with a single 32-slot group each (26k rows: 49 -> 34 us;
// at 106k rows the half windows no longer fit one round: 68 -> 86 us).  PCC_CONV_HALFW=0 disables, =1 extends the rule
// to every launch below 200k rows (tests).
static bool conv_halfw(int64_t n_out) {
  static const int mode = [] { const char* e = getenv("PCC_CONV_HALFW"); return e ? atoi(e) : -1; }();
  return mode < 0 ? n_out <= 65536 : (mode == 1 && n_out < 200000);
}

template <bool HEAD>
static void launch_compact(hipStream_t st, const float* d_in, const int32_t* d_nbr, int k_vol, int64_t pitch,
                           int64_t n_out, const float* d_w, const float* d_bias, int relu, float* d_out,
                           const float* hw, const float* hb, float* ho) {
  // grid rounded up to a multiple of 8: the kernel maps workgroup -> window per XCD
  if (compact_rows() == 64 && conv_w4(n_out))
    hipLaunchKernelGGL((k_gconv_mfma_compact_w4<HEAD, false>), dim3((nblk(n_out, 256) + 7) / 8 * 8), dim3(256), 0, st,
                       d_in, d_nbr, k_vol, pitch, n_out, d_w, d_bias, relu, d_out, hw, hb, ho);
  else if (compact_rows() == 64 && conv_halfw(n_out))
    hipLaunchKernelGGL((k_gconv_mfma_compact<1, HEAD, false, true>), dim3((nblk(n_out, 32) + 7) / 8 * 8), dim3(64), 0, st,
                       d_in, d_nbr, k_vol, pitch, n_out, d_w, d_bias, relu, d_out, hw, hb, ho);
  else if (compact_rows() == 64)
    hipLaunchKernelGGL((k_gconv_mfma_compact<1, HEAD>), dim3((nblk(n_out, 64) + 7) / 8 * 8), dim3(64), 0, st, d_in,
                       d_nbr, k_vol, pitch, n_out, d_w, d_bias, relu, d_out, hw, hb, ho);
  else
    hipLaunchKernelGGL((k_gconv_mfma_compact<2, HEAD>), dim3((nblk(n_out, 128) + 7) / 8 * 8), dim3(64), 0, st, d_in,
                       d_nbr, k_vol, pitch, n_out, d_w, d_bias, relu, d_out, hw, hb, ho);
}

extern "C" int pcc_sparse_conv(pcc_ctx* ctx, const float* d_in, int64_t n_in, const int32_t* d_nbr,
                               int k_vol, int64_t nbr_pitch, int64_t n_out, const float* d_w,
                               const float* d_bias, int cin, int cout, int relu, float* d_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_sparse_conv: null ctx");
  PCC_REQUIRE(k_vol == 27 || k_vol == 8 || k_vol == 1, PCC_E_ARG, "pcc_sparse_conv: k_vol=%d", k_vol);
  PCC_REQUIRE(cin >= 1 && cin <= 64 && cout >= 1 && cout <= 64, PCC_E_ARG,
              "pcc_sparse_conv: cin=%d cout=%d unsupported", cin, cout);
  if (n_out <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_nbr && d_w && d_bias && d_out && nbr_pitch >= n_out && n_in > 0, PCC_E_ARG,
              "pcc_sparse_conv: bad buffers (pitch %lld, n_out %lld, n_in %lld)", (long long)nbr_pitch,
              (long long)n_out, (long long)n_in);
  hipStream_t st = ctx->stream;
  PccProfScope prof(ctx, "sparse_conv", n_out, cin, cout, k_vol);
  const unsigned gm = nblk(n_out, 32 * GC_WAVES);
  const bool aligned = ((uintptr_t)d_in % 16 == 0);
  const bool simple = conv_simple();
  const float* nof = nullptr;
  float* nofo = nullptr;
  if (!force_scalar() && !simple && aligned && cin == 32 && cout == 32 && compact_rows() != 0 &&
      (uintptr_t)d_out % 16 == 0) {
    launch_compact<false>(st, d_in, d_nbr, k_vol, nbr_pitch, n_out, d_w, d_bias, relu, d_out, nof, nof, nofo);
  } else if (!force_scalar() && !simple && aligned && cin == 32 && cout == 64 && compact_rows() != 0 &&
             conv_halfw(n_out) && (uintptr_t)d_out % 16 == 0) {
    // a 32 -> 64 layer of at most 64k rows (h_s output layer at the latent's rows): 32-row windows, the two column
    // halves side by side as grid.y
    hipLaunchKernelGGL((k_gconv_mfma_compact<1, false, false, true, 64>), dim3((nblk(n_out, 32) + 7) / 8 * 8, 2),
                       dim3(64), 0, st, d_in, d_nbr, k_vol, nbr_pitch, n_out, d_w, d_bias, relu, d_out, nof, nof, nofo);
  } else if (!force_scalar() && !simple && aligned && cin == 32 && cout == 32) {
    hipLaunchKernelGGL((k_gconv_mfma_pipe<1, false>), dim3(gm), dim3(GC_WAVES * 64), 0, st, d_in, d_nbr,
                       k_vol, nbr_pitch, n_out, d_w, d_bias, relu, d_out, nof, nof, nofo);
  } else if (!force_scalar() && !simple && aligned && cin == 32 && cout == 64) {
    hipLaunchKernelGGL((k_gconv_mfma_pipe<2, false>), dim3(gm), dim3(GC_WAVES * 64), 0, st, d_in, d_nbr,
                       k_vol, nbr_pitch, n_out, d_w, d_bias, relu, d_out, nof, nof, nofo);
  } else if (!force_scalar() && aligned && cin == 32 && cout == 32) {
    hipLaunchKernelGGL((k_gconv_mfma<32, 1>), dim3(gm), dim3(GC_WAVES * 64), 0, st, d_in, d_nbr, k_vol,
                       nbr_pitch, n_out, d_w, d_bias, relu, d_out);
  } else if (!force_scalar() && aligned && cin == 32 && cout == 64) {
    hipLaunchKernelGGL((k_gconv_mfma<32, 2>), dim3(gm), dim3(GC_WAVES * 64), 0, st, d_in, d_nbr, k_vol,
                       nbr_pitch, n_out, d_w, d_bias, relu, d_out);
  } else if (!force_scalar() && aligned && cin == 4 && cout == 32) {
    hipLaunchKernelGGL((k_gconv_mfma<4, 1>), dim3(gm), dim3(GC_WAVES * 64), 0, st, d_in, d_nbr, k_vol,
                       nbr_pitch, n_out, d_w, d_bias, relu, d_out);
  } else {
    hipLaunchKernelGGL(k_gconv_scalar, dim3(nblk(n_out * cout, 256)), dim3(256), 0, st, d_in, d_nbr,
                       k_vol, nbr_pitch, n_out, d_w, d_bias, cin, cout, relu, d_out);
  }
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_sparse_conv_head(pcc_ctx* ctx, const float* d_in, int64_t n_in, const int32_t* d_nbr,
                                    int k_vol, int64_t nbr_pitch, int64_t n_out, const float* d_w,
                                    const float* d_bias, int cin, int cout, int relu, float* d_out,
                                    const float* d_head_w, const float* d_head_b, float* d_head_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_sparse_conv_head: null ctx");
  PCC_REQUIRE(d_head_w && d_head_b && d_head_out, PCC_E_ARG, "pcc_sparse_conv_head: null head buffers");
  const bool aligned = ((uintptr_t)d_in % 16 == 0);
  if (n_out > 0 && !force_scalar() && aligned && cin == 32 && cout == 32 && (k_vol == 27 || k_vol == 8) &&
      d_in && d_nbr && d_w && d_bias && d_out && nbr_pitch >= n_out && n_in > 0) {
    PccProfScope prof(ctx, "sparse_conv", n_out, cin, cout, k_vol);
    if (compact_rows() != 0 && (uintptr_t)d_out % 16 == 0)
      launch_compact<true>(ctx->stream, d_in, d_nbr, k_vol, nbr_pitch, n_out, d_w, d_bias, relu, d_out, d_head_w,
                           d_head_b, d_head_out);
    else
      hipLaunchKernelGGL((k_gconv_mfma_pipe<1, true>), dim3(nblk(n_out, 32 * GC_WAVES)), dim3(GC_WAVES * 64), 0,
                         ctx->stream, d_in, d_nbr, k_vol, nbr_pitch, n_out, d_w, d_bias, relu, d_out, d_head_w,
                         d_head_b, d_head_out);
    PCC_CHECK_LAUNCH();
    return PCC_OK;
  }
  // generic shapes: the two layers one after the other (same bits)
  PCC_TRY(pcc_sparse_conv(ctx, d_in, n_in, d_nbr, k_vol, nbr_pitch, n_out, d_w, d_bias, cin, cout, relu, d_out));
  return pcc_linear(ctx, d_out, n_out, d_head_w, d_head_b, cout, 1, 0, d_head_out);
}

extern "C" int pcc_sparse_conv_head_up(pcc_ctx* ctx, const float* d_in, int64_t n_parents, const int32_t* d_nbr_parent,
                                       int64_t parent_pitch, const float* d_w, const float* d_bias, int relu,
                                       float* d_out, const float* d_head_w, const float* d_head_b,
                                       float* d_head_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_sparse_conv_head_up: null ctx");
  if (n_parents <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_nbr_parent && d_w && d_bias && d_out && d_head_w && d_head_b && d_head_out &&
                  parent_pitch >= n_parents && n_parents < ((int64_t)1 << 27),
              PCC_E_ARG, "pcc_sparse_conv_head_up: bad buffers (pitch %lld, parents %lld)", (long long)parent_pitch,
              (long long)n_parents);
  PCC_REQUIRE((uintptr_t)d_in % 16 == 0 && (uintptr_t)d_out % 16 == 0, PCC_E_ARG,
              "pcc_sparse_conv_head_up: feature rows must be 16-byte aligned");
  PCC_REQUIRE(pcc_conv_up_fused(), PCC_E_ARG,
              "pcc_sparse_conv_head_up: only the row-compacting MFMA kernel has this form (PCC_FORCE_SCALAR / "
              "PCC_CONV_SIMPLE / PCC_CONV_COMPACT=0 / PCC_CONV_UP=0 select the explicit rule book: pcc_derive_map_up)");
  const int64_t n_out = 8 * n_parents;
  PccProfScope prof(ctx, "sparse_conv", n_out, 32, 32, 27);
  static const int conv16 = [] { const char* e = getenv("PCC_CONV16"); return e ? atoi(e) : 0; }();
  if (conv16) {
    pcc_arena_reset(ctx);
    PCC_TRY(pcc_arena_reserve(ctx, 27 * 1024 * 4 + 512));
    float* wsw = (float*)pcc_arena_alloc(ctx, 27 * 1024 * 4);
    if (!wsw) return PCC_E_NOMEM;
    hipLaunchKernelGGL(k_conv16_swizzle, dim3(27 * 4), dim3(256), 0, ctx->stream, d_w, 27, wsw);
    if (conv16 == 2)
      hipLaunchKernelGGL((k_gconv16<true, true, true>), dim3((nblk(n_out, 64) + 7) / 8 * 8), dim3(64), 0, ctx->stream,
                         d_in, d_nbr_parent, 27, parent_pitch, n_out, wsw, d_bias, relu, d_out, d_head_w, d_head_b,
                         d_head_out);
    else
      hipLaunchKernelGGL((k_gconv16<true, true, false>), dim3((nblk(n_out, 64) + 7) / 8 * 8), dim3(64), 0, ctx->stream,
                         d_in, d_nbr_parent, 27, parent_pitch, n_out, wsw, d_bias, relu, d_out, d_head_w, d_head_b,
                         d_head_out);
    PCC_CHECK_LAUNCH();
    return PCC_OK;
  }
  if (conv_w4(n_out))
    hipLaunchKernelGGL((k_gconv_mfma_compact_w4<true, true>), dim3((nblk(n_out, 256) + 7) / 8 * 8), dim3(256), 0,
                       ctx->stream, d_in, d_nbr_parent, 27, parent_pitch, n_out, d_w, d_bias, relu, d_out, d_head_w,
                       d_head_b, d_head_out);
  else
    hipLaunchKernelGGL((k_gconv_mfma_compact<1, true, true>), dim3((nblk(n_out, 64) + 7) / 8 * 8), dim3(64), 0,
                       ctx->stream, d_in, d_nbr_parent, 27, parent_pitch, n_out, d_w, d_bias, relu, d_out, d_head_w,
                       d_head_b, d_head_out);
  PCC_CHECK_LAUNCH();
  return PCC_OK;
}

extern "C" int pcc_convT_gen(pcc_ctx* ctx, const float* d_in, int64_t n_in, const float* d_w,
                             const float* d_bias, int cin, int cout, int relu, float* d_out) {
  PCC_REQUIRE(ctx, PCC_E_ARG, "pcc_convT_gen: null ctx");
  PCC_REQUIRE(cin >= 1 && cin <= 64 && cout >= 1 && cout <= 64, PCC_E_ARG,
              "pcc_convT_gen: cin=%d cout=%d unsupported", cin, cout);
  if (n_in <= 0) return PCC_OK;
  PCC_REQUIRE(d_in && d_w && d_bias && d_out, PCC_E_ARG, "pcc_convT_gen: null buffers");
  hipStream_t st = ctx->stream;
  PccProfScope prof(ctx, "convT_gen", n_in, cin, cout, 8);
  const bool aligned = ((uintptr_t)d_in % 16 == 0);
  if (!force_scalar() && aligned && cin == 32 && cout == 32) {
    hipLaunchKernelGGL((k_convT_mfma<1>), dim3(nblk(n_in, 32 * GC_WAVES)), dim3(GC_WAVES * 64), 0, st,
                       d_in, n_in, d_w, d_bias, relu, d_out, (const uint32_t*)nullptr);
  } else if (!force_scalar() && aligned && cin == 32 && cout == 64) {
    hipLaunchKernelGGL((k_convT_mfma<2>), dim3(nblk(n_in, 32 * GC_WAVES)), dim3(GC_WAVES * 64), 0, st,
                       d_in, n_in, d_w, d_bias, relu, d_out, (const uint32_t*)nullptr);
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
  hipLaunchKernelGGL((k_convT_mfma<1>), dim3(nblk(n_in, 32 * GC_WAVES)), dim3(GC_WAVES * 64), 0, ctx->stream, d_in,
                     n_in, d_w, d_bias, relu, d_out, d_rows);
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
// diagnostic builds only: the per-wave phase sums of the last k_gconv_mfma_compact_w4<.., UP> launch
extern "C" int pcc_debug_stamps(unsigned long long* h_out, int n) {
  if (n > 4096 * PCC_NSTAMP) n = 4096 * PCC_NSTAMP;
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(pcc_stamp_buf), (size_t)n * 8) == hipSuccess ? 0 : -2;
}
#endif
