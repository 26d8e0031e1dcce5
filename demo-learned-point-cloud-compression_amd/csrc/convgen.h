// convgen.h — the gather-convolution of conv16.h for any channel widths that are multiples of 16 (up to 128): the layers of
// a model whose config.yaml names other widths than this build's default of 32 (ColorModel(config["model"]),
// sender/encoder/codec_pipeline.py:65).  Included by conv.hip after conv16.h.
//
// Same arithmetic contract (out = bias; k ascending over present neighbours — or siblings first, include/pcc.h — ci
// ascending: fmaf) and the same shape of the work as k_gconv16: one wave per 64-row window, the rows that have an offset
// packed into 16-slot items by ballot + mbcnt, accumulators in LDS, v_mfma_f32_16x16x4_f32 on the transposed product.
// What is general:
//   * C_in: the input channels of an offset are contracted in CHUNKS of CH = 32 (C_in a multiple of 32) or 16 channels:
//     chunk c of offset k is one more round over the offset's items with its own block of weights and its own piece of
//     the gathered rows — "virtual offsets" (k, c), c ascending inside k, so ci stays ascending;
//   * C_out: 32 columns per workgroup (grid.y column blocks); a last block of 16 columns runs the low plane only;
//   * the order of the offsets (siblings first or plain) is a launch argument: with other widths the g_s layers run as
//     this kernel + pcc_linear for the occupancy head (no fused epilogue, no in-kernel child rule book).
// It is written for correctness on the matrix pipe, not tuned like the 32-channel kernels: the gathers of a round are
// requested together in front of its items, and the next offset's neighbour index one round ahead.
#pragma once

// weights [k_vol][cin][cout] -> [k_vol][cin / CH][ceil(cout / 32)][64 lanes][CH / 2]: lane (m, q) of chunk c, column block y:
// s = 0 .. CH/4-1: W[c CH + 4s + q][32y + m], then the same for column 32y + 16 + m (0 beyond cout)
__global__ __launch_bounds__(256) void k_convgen_swizzle(const float* __restrict__ w, int k_vol, int cin, int cout, int ch,
                                                         float* __restrict__ wsw) {
  const int ny = (cout + 31) / 32, nc = cin / ch, per = ch / 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)k_vol * nc * ny * 64 * per) return;
  const int e = (int)(t % per), l = (int)((t / per) % 64), y = (int)((t / (per * 64)) % ny);
  const int c = (int)((t / ((int64_t)per * 64 * ny)) % nc), k = (int)(t / ((int64_t)per * 64 * ny * nc));
  const int m = l & 15, q = l >> 4, s = e % (ch / 4), hi = e / (ch / 4);
  const int col = 32 * y + 16 * hi + m;
  wsw[t] = col < cout ? w[((int64_t)k * cin + c * ch + 4 * s + q) * cout + col] : 0.f;
}

template <int CH>
__global__ __launch_bounds__(64) void k_gconv_gen(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int k_vol, int64_t pitch, int64_t n_out,
    const float* __restrict__ wsw, const float* __restrict__ bias, int cin, int cout, int relu, int sib,
    float* __restrict__ out) {
  static_assert(CH == 32 || CH == 16, "chunks of 32 or 16 input channels");
  constexpr int R = 64, HP = (R + 1) * 16, NI = R / 16, NS = CH / 4, NV = CH / 16;   // NS MFMA steps, NV 16-B pieces per lane
  __shared__ __attribute__((aligned(16))) float acc_lds[2 * HP];
  __shared__ __attribute__((aligned(8))) int2 rec[64];   // slot -> (input row, accumulator row address)
  const int lane = threadIdx.x;
  const int ny = (int)gridDim.y, ycol = (int)blockIdx.y, col0 = 32 * ycol;
  const bool hi_live = col0 + 16 < cout;   // a last block of 16 columns has no high plane
  const int nc = cin / CH;
  const int64_t window = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t row0 = window * R;
  if (row0 >= n_out) return;
  const int n = lane & 15, q = lane >> 4;
  const int grow = lane >> 3, chunk8 = lane & 7;
  auto acc_at = [&](int half, int row, int qq) -> int { return half * HP + row * 16 + (((qq + 2 * (row >> 2)) & 3) << 2); };
  auto acc_row = [](int row) -> int { return row * 64 + (((row >> 1) & 2) << 4); };
  const int a_own = acc_row(lane), a_sink = acc_row(R), q16 = q << 4;
  {  // accumulators start at the bias (columns beyond cout: 0)
    float4 b4;
    float* bb = reinterpret_cast<float*>(&b4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + chunk8 * 4 + j;
      bb[j] = col < cout ? bias[col] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < R / 8; ++it)
      *reinterpret_cast<float4*>(&acc_lds[acc_at(chunk8 >> 2, it * 8 + grow, chunk8 & 3)]) = b4;
    if (lane < 8) *reinterpret_cast<float4*>(&acc_lds[acc_at(lane >> 2, R, lane & 3)]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t r_own = row0 + lane;
  const bool row_ok = r_own < n_out;
  const int64_t rc = row_ok ? r_own : n_out - 1;
  auto acc_read = [&](int arow, f32x4& lo, f32x4& hi) {
    const float* base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(acc_lds) + (arow ^ q16));
    const float4 a = *reinterpret_cast<const float4*>(base);
    const float4 b = *reinterpret_cast<const float4*>(base + HP);
    lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; lo[3] = a.w;
    hi[0] = b.x; hi[1] = b.y; hi[2] = b.z; hi[3] = b.w;
  };
  auto acc_write = [&](int arow, const f32x4& lo, const f32x4& hi) {
    float* base = reinterpret_cast<float*>(reinterpret_cast<char*>(acc_lds) + (arow ^ q16));
    *reinterpret_cast<float4*>(base) = make_float4(lo[0], lo[1], lo[2], lo[3]);
    *reinterpret_cast<float4*>(base + HP) = make_float4(hi[0], hi[1], hi[2], hi[3]);
  };
  // B operands of a slot in MFMA order.  Lane (n, t) holds, per 16-channel group g of the chunk, channels 16 g + 4 t + e
  // (e = 0..3, one 16-B piece); step s = 4 g + e' needs channel 16 g + 4 e' + q in lane q: a 4 x 4 transpose per group
  // among the four q-lanes of the slot
  auto shape = [&](const float4 (&g)[NV], float (&xv)[NS]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      unsigned m0 = __float_as_uint(g[v].x), m1 = __float_as_uint(g[v].y), m2 = __float_as_uint(g[v].z), m3 = __float_as_uint(g[v].w);
      u32x2 p = __builtin_amdgcn_permlane32_swap(m0, m2, false, false);
      m0 = p[0]; m2 = p[1];
      p = __builtin_amdgcn_permlane32_swap(m1, m3, false, false);
      m1 = p[0]; m3 = p[1];
      p = __builtin_amdgcn_permlane16_swap(m0, m1, false, false);
      m0 = p[0]; m1 = p[1];
      p = __builtin_amdgcn_permlane16_swap(m2, m3, false, false);
      m2 = p[0]; m3 = p[1];
      xv[4 * v + 0] = __uint_as_float(m0); xv[4 * v + 1] = __uint_as_float(m1);
      xv[4 * v + 2] = __uint_as_float(m2); xv[4 * v + 3] = __uint_as_float(m3);
    }
  };

  const int passes = sib ? 2 : 1;
  int32_t nb_next = nbr[rc];   // offset 0
  for (int pass = 0; pass < passes; ++pass) {
    for (int k = 0; k < k_vol; ++k) {
      const int32_t nb = nb_next;
      {  // the neighbour index of the next offset (of the next pass: offset 0 again) is requested a round ahead
        const int kn = k + 1 < k_vol ? k + 1 : 0;
        nb_next = nbr[(int64_t)kn * pitch + rc];
      }
      bool p = row_ok && nb >= 0;
      if (sib) p = p && ((((int64_t)nb >> 3) == (rc >> 3)) == (pass == 0));
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(p);
      const int cnt = __popcll(bal);
      if (cnt == 0) continue;   // wave-uniform
      const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
      PCC16_SYNC();   // the records of the previous offset have been read
      rec[p ? rank : cnt + lane - rank] = p ? make_int2(nb, a_own) : make_int2(-1, a_sink);
      PCC16_SYNC();
      int rrow[NI], racc[NI];
#pragma unroll
      for (int g = 0; g < NI; ++g) {
        const int2 r = rec[g * 16 + n];
        rrow[g] = r.x;
        racc[g] = r.y;
      }
      for (int c = 0; c < nc; ++c) {
        // weights of (k, c, column block): CH / 2 floats per lane
        float wv[CH / 2];
        {
          const float4* wp = reinterpret_cast<const float4*>(wsw + ((((int64_t)k * nc + c) * ny + ycol) * 64 + lane) * (CH / 2));
#pragma unroll
          for (int j = 0; j < CH / 8; ++j) {
            const float4 t = wp[j];
            wv[4 * j] = t.x; wv[4 * j + 1] = t.y; wv[4 * j + 2] = t.z; wv[4 * j + 3] = t.w;
          }
        }
        float4 G[NI][NV];
#pragma unroll
        for (int g = 0; g < NI; ++g) {
          if (cnt > 16 * g) {   // wave-uniform
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
              if (rrow[g] >= 0) t = *reinterpret_cast<const float4*>(in + (int64_t)rrow[g] * cin + c * CH + 16 * v + 4 * q);
              G[g][v] = t;
            }
          }
        }
#pragma unroll
        for (int g = 0; g < NI; ++g) {
          if (cnt > 16 * g) {
            f32x4 lo, hi;
            acc_read(racc[g], lo, hi);
            float xv[NS];
            shape(G[g], xv);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              lo = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], xv[s], lo, 0, 0, 0);
              if (hi_live) hi = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[NS + s], xv[s], hi, 0, 0, 0);
            }
            acc_write(racc[g], lo, hi);
          }
        }
      }
    }
  }
  PCC16_SYNC();
  // epilogue: the window's rows are contiguous in `out`; lane (grow, chunk8) stores piece chunk8 (4 columns) of rows grow + 8 it
#pragma unroll
  for (int it = 0; it < R / 8; ++it) {
    const int r = it * 8 + grow;
    float4 v = *reinterpret_cast<const float4*>(&acc_lds[acc_at(chunk8 >> 2, r, chunk8 & 3)]);
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    const int col = col0 + chunk8 * 4;
    if (row0 + r < n_out && col < cout) *reinterpret_cast<float4*>(out + (row0 + r) * cout + col) = v;
  }
}
