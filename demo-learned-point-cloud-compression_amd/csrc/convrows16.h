// convrows16.h — the latent-sized 32 -> 32 / 32 -> 64 gather-convolutions of g_a, h_a and h_s (codec_pipeline.py:273,287,
// 354; codec_parallel.py:302-303) on an explicit rule book: launches of under one round of waves (included by
// convrows16.hip; the weights come in conv16.h's operand order).
//
// A launch of a few hundred to a few thousand 32-row windows of k_gconv16 lasts as long as ONE window: k_vol dependent
// steps, each a ballot, a compaction through LDS, a record read, a gather and an accumulator round trip — 0.9 us per step
// with nobody to hide behind (1.6k and 6.6k rows, 3^3: 24 us; 26k rows: 25 us alone, 32 in the step).  Such a launch is
// bound by the length of that chain, not by anything it moves or multiplies.  Here the chain is cut to the matrix work:
//
//   * a window is 16 rows = one item: slot n IS row row0 + n, whether or not it has the offset.  No ballot, no compaction,
//     no slot records, no LDS at all; the accumulators (two 16 x 16 tiles) stay in registers for the whole window;
//   * a row that lacks offset k gathers from beyond the buffer (zeros, no fetch) and keeps its accumulator by a select
//     behind the item's chains: its chain sees exactly the PRESENT neighbours, k ascending (include/pcc.h) — fmaf(0, w,
//     acc) would not do: it turns an accumulator of -0 into +0, and 0 x inf into a NaN;
//   * all k_vol neighbour indices of a row are requested up front, the offsets are unrolled, D gathers and WD weight
//     blocks are in flight ahead of the offset being contracted: nothing a step needs was requested in that step.
//
// The matrix pipe runs every offset on 16 slots (fill = pairs per row / k_vol, 0.3 .. 0.5 on these levels): 16 MFMAs per
// window and offset — 26k rows x 27 offsets = 23 M cycles over 1024 SIMDs = 11 us.  Measured (tools/bench_small_conv.py, one
// box, k_gconv16 -> this kernel; the figures include ~8 us of the caller per launch): 1.6k / 6.6k rows 3^3 24 -> 14 us,
// 2^3 12.5 -> 11.3; 26k rows 2^3 15.0 -> 11.9, 3^3 32 -> 32 25 -> 25 (32.1 -> 26.5 inside the step), 3^3 32 -> 64 35.5 -> 41.8
// (46.8 with a wave per column half).  So this form serves launches of at most two waves per SIMD — one for 32 -> 64 —
// (kRows16MaxWaves in conv.hip), and k_gconv16's compaction the others.
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// B operands of a slot in MFMA order from the two 16-B pieces lane (n, q) holds of its row (channels 8q .. 8q+7):
// transposes of the 4 x 4 blocks among the four q-lanes (conv16.h, PERM = false)
__device__ __forceinline__ void pcc_rows16_shape(const float4& g0, const float4& g1, float (&xv)[8]) {
  unsigned m[2][4] = {{__float_as_uint(g0.x), __float_as_uint(g0.y), __float_as_uint(g0.z), __float_as_uint(g0.w)},
                      {__float_as_uint(g1.x), __float_as_uint(g1.y), __float_as_uint(g1.z), __float_as_uint(g1.w)}};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    u32x2 p = __builtin_amdgcn_permlane32_swap(m[b][0], m[b][2], false, false);
    m[b][0] = p[0]; m[b][2] = p[1];
    p = __builtin_amdgcn_permlane32_swap(m[b][1], m[b][3], false, false);
    m[b][1] = p[0]; m[b][3] = p[1];
    p = __builtin_amdgcn_permlane16_swap(m[b][0], m[b][1], false, false);
    m[b][0] = p[0]; m[b][1] = p[1];
    p = __builtin_amdgcn_permlane16_swap(m[b][2], m[b][3], false, false);
    m[b][2] = p[0]; m[b][3] = p[1];
#pragma unroll
    for (int t = 0; t < 4; ++t) xv[2 * t + b] = __uint_as_float(m[b][t]);
  }
}

// COUT 32 or 64: a wave produces all COUT columns of its 16 rows (COUT / 32 pairs of accumulator tiles fed by ONE gathered
// and shaped operand — as two launches' worth of waves, one per column half, a 32 -> 64 layer of 26k rows took 46.8 us
// against k_gconv16's 35.5).  KV = k_vol, 27 or 8, known at compile time: the offsets are unrolled and the indices live in
// registers.  in_bytes: bytes of `in` (n_in * 128 < 2^32 - 128, conv16.h's narrow form)
template <int COUT, int KV>
__global__ __launch_bounds__(64) void k_gconv_rows16(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int64_t pitch, int64_t n_out,
    const float* __restrict__ wsw, const float* __restrict__ bias, int relu, float* __restrict__ out, uint32_t in_bytes) {
  constexpr int D = KV < 6 ? KV : 6;    // gathers in flight
  constexpr int WD = 2;                 // weight blocks in flight
  constexpr int NY = COUT / 32;
  const int lane = threadIdx.x, n = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  if (row0 >= n_out) return;
  const int64_t r = row0 + n;
  const bool row_ok = r < n_out;
  const int64_t rc = row_ok ? r : n_out - 1;

  // (the scheduling barriers pin every request where it is written: left alone, the compiler sinks each load to its
  // first use — 60 registers, and every step waits for the index, the row and the weights it has just asked for)
  int32_t nb[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) nb[k] = nbr[(int64_t)k * pitch + rc];
  __builtin_amdgcn_sched_barrier(0);

  constexpr uint32_t kPadOff = 0xFFFFFF80u;   // beyond the buffer: the load returns zeros without a fetch
  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00027000);
  const uint32_t qoff = (uint32_t)q * 32u;
  auto gather = [&](int k, float4& g0, float4& g1) {
    const uint32_t off = (row_ok && nb[k] >= 0) ? (((uint32_t)nb[k] << 7) | qoff) : kPadOff;
    const auto r0 = __builtin_amdgcn_raw_buffer_load_b128(in_rs, off, 0, 0);
    const auto r1 = __builtin_amdgcn_raw_buffer_load_b128(in_rs, off + 16u, 0, 0);
    g0 = make_float4(__uint_as_float(r0[0]), __uint_as_float(r0[1]), __uint_as_float(r0[2]), __uint_as_float(r0[3]));
    g1 = make_float4(__uint_as_float(r1[0]), __uint_as_float(r1[1]), __uint_as_float(r1[2]), __uint_as_float(r1[3]));
  };
  // the swizzled copy is [k][column half][lane][16] (conv16.h): NY consecutive 4-KB blocks per offset
  auto load_w = [&](float4 (&W)[NY][4], int k) {
    const float4* p = reinterpret_cast<const float4*>(
        reinterpret_cast<const char*>(wsw + (int64_t)k * NY * 1024) + (uint32_t)lane * 64u);
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int j = 0; j < 4; ++j) W[y][j] = p[y * 256 + j];
  };

  float4 Wd[WD + 1][NY][4];
#pragma unroll
  for (int k = 0; k < WD && k < KV; ++k) load_w(Wd[k], k);
  f32x4 lo[NY], hi[NY];
#pragma unroll
  for (int y = 0; y < NY; ++y) {
    const float* bp = bias + 32 * y + 4 * q;
    lo[y] = f32x4{bp[0], bp[1], bp[2], bp[3]};
    hi[y] = f32x4{bp[16], bp[17], bp[18], bp[19]};
  }
  __builtin_amdgcn_sched_barrier(0);
  float4 G[D][2];
#pragma unroll
  for (int d = 0; d < D; ++d) gather(d, G[d][0], G[d][1]);
  __builtin_amdgcn_sched_barrier(0);

#pragma unroll
  for (int k = 0; k < KV; ++k) {
    if (k + WD < KV) load_w(Wd[(k + WD) % (WD + 1)], k + WD);
    __builtin_amdgcn_sched_barrier(0);
    float xv[8];
    pcc_rows16_shape(G[k % D][0], G[k % D][1], xv);
    __builtin_amdgcn_sched_barrier(0);
    if (k + D < KV) gather(k + D, G[k % D][0], G[k % D][1]);
    __builtin_amdgcn_sched_barrier(0);
    const bool present = row_ok && nb[k] >= 0;
#pragma unroll
    for (int y = 0; y < NY; ++y) {
      const float4 (&Wc)[4] = Wd[k % (WD + 1)][y];
      const float wl[8] = {Wc[0].x, Wc[0].y, Wc[0].z, Wc[0].w, Wc[1].x, Wc[1].y, Wc[1].z, Wc[1].w};
      const float wh[8] = {Wc[2].x, Wc[2].y, Wc[2].z, Wc[2].w, Wc[3].x, Wc[3].y, Wc[3].z, Wc[3].w};
      f32x4 l2 = lo[y], h2 = hi[y];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        l2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xv[s], l2, 0, 0, 0);
        h2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xv[s], h2, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[y][j] = present ? l2[j] : lo[y][j];
        hi[y][j] = present ? h2[j] : hi[y][j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // lane (n, q) holds channels 32y + 4q .. 4q+3 and 32y + 16 + 4q .. of row row0 + n: two 16-B stores per column half
  if (row_ok) {
#pragma unroll
    for (int y = 0; y < NY; ++y) {
      float4 a = make_float4(lo[y][0], lo[y][1], lo[y][2], lo[y][3]), b = make_float4(hi[y][0], hi[y][1], hi[y][2], hi[y][3]);
      if (relu) {
        a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
        b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f);
      }
      float* op = out + r * COUT + 32 * y + 4 * q;
      *reinterpret_cast<float4*>(op) = a;
      *reinterpret_cast<float4*>(op + 16) = b;
    }
  }
}
