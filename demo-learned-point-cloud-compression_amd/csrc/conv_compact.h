// conv_compact.h — row-compacting form of the 32->32 gather-convolution (included by conv.hip).
//
// The out-stationary kernels in conv.hip spend one 32-row MFMA tile on every offset that ANY of the
// tile's rows has, so on surface data roughly half of the matrix-core work multiplies zero rows
// (measured: SQ_INSTS_MFMA == 27 offsets x all tiles; useful pairs / issued pairs = 0.53).  Here a wave
// owns R = 64 or 128 consecutive (Morton-sorted) output rows and, offset by offset, packs only the
// rows that HAVE the neighbour into 32-row groups:
//
//   for k ascending:  slots = compact(rows of the window whose nbr[k] >= 0)      (ballot + mbcnt)
//                     for each group of 32 slots:  C  <- accumulators of those rows (LDS, 4 x b128)
//                                                  C  += W[k]^T  x  X_gathered^T   (16 MFMA 32x32x2)
//                                                  accumulators <- C
//
// The product is computed transposed (M = output channel, N = row slot) so that one lane holds 16
// channels of ONE row: reading / writing a row's accumulator is 4 ds_read_b128 / ds_write_b128 at
// row*36 floats (pitch 36 keeps 16 consecutive rows on distinct banks), about 8 KB of LDS traffic per
// 1024 MFMA cycles.  Per output row the arithmetic is still
//   out = bias; for k ascending over PRESENT neighbours, ci ascending: out = fmaf(x, w, out)
// (a row simply does not take part in the groups of an offset it lacks), so the bits equal those of
// the dense-tile kernels, the scalar kernel and the C oracle.  Pad slots (the rest of a partly filled
// group, and a group with no present row at all) gather row 0 of the input and accumulate into a
// sink row that is never stored: every load is unconditional, which keeps the loads out of control
// flow so that the compiler's s_waitcnt counts stay exact (vmcnt(N), not vmcnt(0)).
//
// Operand shaping without LDS: lane (i, h) loads the 64-B half h of the neighbour row of slot i
// (x[16h .. 16h+15]); the MFMA wants x[2s + h] in step s.  v_permlane32_swap of the even-j register's
// upper half with the odd-j register's lower half yields both ci = 2t+h (step t) and ci = 16+2t+h
// (step 8+t) in place: 8 swaps per group, which the compiler slots between the MFMAs.
//
// Pipeline per wave: neighbour indices of offset k+2 are in flight (registers) while offset k+1 is
// being compacted and offset k is contracted; the gathered rows of ALL groups of offset k+1 (4 x
// dwordx4 per lane and group) and its 16 weight dwords go in flight into a second register set
// (sets alternate between even and odd offsets) before the groups of offset k run.
//
// Measured on the 3,262,640-row layer of bench.py (tools/bench_conv.py, MI355X): dense tiles 1.69-1.80
// ms, this kernel 1.40-1.49 ms with R = 64 (R = 128 the same within 1 %: fewer weight re-reads,
// lower occupancy).  Variants that were measured and dropped: gathered rows staged through an LDS
// tile (1.70 ms: 16 ds_write + 16 ds_read per group), compaction by ds_permute with the slot lists in
// registers (1.83-1.87 ms) and a single weight register set reloaded behind its last reader (1.72 ms).
// Layers of >= 200k rows run k_gconv_mfma_compact_w4 at the end of this file (weights shared through LDS).
#pragma once

// Experiment switch (build-time): priority of a wave while it runs the MFMA chain of a group.  Waves of one SIMD that
// reach their chains together share the matrix pipe MFMA by MFMA and then also finish together — they fall into
// step, and bookkeeping time and matrix time add up instead of overlapping.  A wave that holds a raised priority for
// the length of its chain is served first: chains run one after the other and the waves drift apart.
#ifndef PCC_CONV_PRIO
#define PCC_CONV_PRIO 0
#endif
#if PCC_CONV_PRIO == 1
#define PCC_PRIO_BEGIN() __builtin_amdgcn_s_setprio(3)
#define PCC_PRIO_MID(s) do { } while (0)
#define PCC_PRIO_END() __builtin_amdgcn_s_setprio(0)
#elif PCC_CONV_PRIO == 2
#define PCC_PRIO_BEGIN() __builtin_amdgcn_s_setprio(1)
#define PCC_PRIO_MID(s) do { if ((s) == 5) __builtin_amdgcn_s_setprio(2); if ((s) == 10) __builtin_amdgcn_s_setprio(3); } while (0)
#define PCC_PRIO_END() __builtin_amdgcn_s_setprio(0)
#elif PCC_CONV_PRIO == 3   /* the reverse: bookkeeping first, chains at the lowest priority */
#define PCC_PRIO_BEGIN() __builtin_amdgcn_s_setprio(0)
#define PCC_PRIO_MID(s) do { } while (0)
#define PCC_PRIO_END() __builtin_amdgcn_s_setprio(3)
#else
#define PCC_PRIO_BEGIN() do { } while (0)
#define PCC_PRIO_MID(s) do { } while (0)
#define PCC_PRIO_END() do { } while (0)
#endif
#if PCC_CONV_PRIO == 3
#define PCC_PRIO_KERNEL() __builtin_amdgcn_s_setprio(3)
#elif PCC_CONV_PRIO == 4   /* static: workgroups that share a SIMD get different priorities */
#define PCC_PRIO_KERNEL() do { switch ((blockIdx.x >> 3) & 3) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); } } while (0)
#else
#define PCC_PRIO_KERNEL() do { } while (0)
#endif

// Diagnostic build only (-DPCC_CONV_STAMP=1; results are unchanged, timing is not): cycle stamps (s_memtime) at the
// phase boundaries of an offset step of k_gconv_mfma_compact_w4, summed per wave over its 27 steps and written to a
// buffer of their own (pcc_debug_stamps).  MI355X_MICROARCH.md, "in-kernel stamps".
#ifndef PCC_CONV_STAMP
#define PCC_CONV_STAMP 0
#endif
#if PCC_CONV_STAMP
#define PCC_NSTAMP 10
__device__ unsigned long long pcc_stamp_buf[4096 * PCC_NSTAMP];
#define PCC_STAMP(i)                                                                  \
  do {                                                                                \
    unsigned long long t_;                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
    st_sum[i] += t_ - st_last;                                                        \
    st_last = t_;                                                                     \
  } while (0)
#else
#define PCC_STAMP(i) do { } while (0)
#endif

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// UP: the rows are the 8 generative children (row 8p + o) of the rows of a PARENT level and `nbr` / `pitch` are
// the parent level's rule book: the child's neighbour at offset k is child o' of parent-neighbour kp (per axis
// t = o + d, D = floor(t / 2), o' = t & 1 — pcc_derive_map_up's rule), formed here from one parent-book load, so
// the 27 x 8N child rule book is never written to or read from HBM.
// HALFW: 32-row windows (lanes 32..63 own no row): every offset is ONE group, so a window's 27 dependent steps are
// shorter — for launches of a single round of windows, whose duration is one window's latency, not throughput.
// COUT: output channels of the layer (32 or 64).  A workgroup always produces 32 of them, columns [32 blockIdx.y,
// 32 blockIdx.y + 32): a 32 -> 64 layer launches grid.y = 2 and the two halves of a window run side by side (used for
// the h_s output layer evaluated at the latent's rows only — a launch far smaller than one round of windows).
template <int RCH, bool HEAD, bool UP = false, bool HALFW = false, int COUT = 32>
__global__ __launch_bounds__(64) void k_gconv_mfma_compact(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int k_vol, int64_t pitch,
    int64_t n_out, const float* __restrict__ w, const float* __restrict__ bias, int relu,
    float* __restrict__ out, const float* __restrict__ head_w, const float* __restrict__ head_b,
    float* __restrict__ head_out) {
  static_assert(!HALFW || RCH == 1, "half windows are a variant of the 64-row kernel");
  static_assert(COUT == 32 || (COUT == 64 && !HEAD), "the fused head reads all channels of a row");
  const int col0 = COUT == 32 ? 0 : 32 * (int)blockIdx.y;
  constexpr int R = HALFW ? 32 : 64 * RCH;  // output rows of this wave
  constexpr int AP = 36;       // accumulator row pitch (floats), 16-B aligned rows
  __shared__ __attribute__((aligned(16))) float acc_lds[(R + 1) * AP];  // row R = sink for pad slots
  __shared__ int32_t slot_in[2][R];
  __shared__ uint8_t slot_row[2][R];

  const int lane = threadIdx.x;
  PCC_PRIO_KERNEL();
  // XCD-aware window order: workgroups are dealt round-robin to the 8 XCDs, so workgroup b works on
  // window (b % 8) * (grid / 8) + b / 8: every XCD walks one contiguous eighth of the (Morton-sorted)
  // rows, and the neighbour rows that adjacent windows share are fetched into ONE L2 instead of eight
  const int64_t window = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t row0 = window * R;
  if (row0 >= n_out) return;  // the grid is rounded up to a multiple of 8
  const int i = lane & 31, h = lane >> 5;
  const int grow = lane >> 3, chunk = lane & 7;

  // ---- accumulators start at the bias
  {
    const float* bp = bias + col0 + chunk * 4;
    const float4 b4 = make_float4(bp[0], bp[1], bp[2], bp[3]);
#pragma unroll
    for (int it = 0; it < R / 8; ++it)
      *reinterpret_cast<float4*>(&acc_lds[(it * 8 + grow) * AP + chunk * 4]) = b4;
    if (lane < 8) *reinterpret_cast<float4*>(&acc_lds[R * AP + lane * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  int32_t nbreg[RCH];
  auto load_nb = [&](int k) {
#pragma unroll
    for (int c = 0; c < RCH; ++c) {  // branch-free (clamped address + select): keeps the loads out of control flow
      const int64_t r = row0 + c * 64 + lane;
      const int kk = k < k_vol ? k : k_vol - 1;
      const int64_t rc = r < n_out ? r : n_out - 1;
      if constexpr (UP) {
        const int o = (int)(rc & 7);
        const int tx = ((o >> 2) & 1) + (kk / 9) - 1, ty = ((o >> 1) & 1) + ((kk / 3) % 3) - 1, tz = (o & 1) + (kk % 3) - 1;
        const int kp = ((tx + 2) >> 1) * 9 + ((ty + 2) >> 1) * 3 + ((tz + 2) >> 1);
        const int op = ((tx & 1) << 2) | ((ty & 1) << 1) | (tz & 1);
        const int32_t pr = nbr[(int64_t)kp * pitch + (rc >> 3)];
        nbreg[c] = (k < k_vol && r < n_out && pr >= 0 && (!HALFW || lane < 32)) ? ((pr << 3) | op) : -1;
      } else {
        const int32_t v = nbr[(int64_t)kk * pitch + rc];
        nbreg[c] = (k < k_vol && r < n_out && (!HALFW || lane < 32)) ? v : -1;
      }
    }
  };
  // pack the rows that have the offset held in nbreg into slot lists `b`; returns their count
  auto compact = [&](int b) -> int {
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < RCH; ++c) {
      const bool p = nbreg[c] >= 0;
      const unsigned long long bal = __ballot(p);
      const int rank = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
      if (p) {
        slot_in[b][rank] = nbreg[c];
        slot_row[b][rank] = (uint8_t)(c * 64 + lane);
      }
      cnt += __popcll(bal);
    }
    // every remaining slot is a pad: row 0 of `in` (any valid row), accumulated into the sink row
#pragma unroll
    for (int c = 0; c < RCH; ++c) {
      const int sl = cnt + c * 64 + lane;
      if (sl < R) {
        slot_in[b][sl] = 0;
        slot_row[b][sl] = (uint8_t)R;
      }
    }
    return cnt;
  };

  constexpr int NG = R / 32;  // groups an offset can have
  // gathered rows of every group of an offset: two register sets, even offsets use gA, odd ones gB,
  // so the set of offset k+1 fills (whole offset in flight) while the groups of offset k are contracted
  float4 gA[NG][4], gB[NG][4];
  float bw[16];
  auto load_w = [&](int k) {
    const float* wk = w + (int64_t)k * 32 * COUT + col0;
#pragma unroll
    for (int s = 0; s < 16; ++s) bw[s] = wk[(2 * s + h) * COUT + i];
  };

#define PCC_WAVE_SYNC()                                      \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)
// pad slots carry row 0 (any valid row: their products go to the sink row)
#define PCC_GATHER(G, b, cnt)                                                                      \
  do {                                                                                             \
    _Pragma("unroll") for (int grp = 0; grp < NG; ++grp) {                                         \
      const float* xr = in + (int64_t)slot_in[b][grp * 32 + i] * 32 + h * 16;                      \
      _Pragma("unroll") for (int it = 0; it < 4; ++it)                                             \
        G[grp][it] = *reinterpret_cast<const float4*>(xr + it * 4);                                \
    }                                                                                              \
  } while (0)
// one offset: compact offset k+1, put its rows in flight into GN, contract the groups held in GC
#define PCC_STEP(GC, GN, cur, k)                                                                   \
  do {                                                                                             \
    const int cnt_next = compact((cur) ^ 1); /* offset k+1 (all absent past the last offset) */    \
    load_nb((k) + 2);                                                                              \
    PCC_WAVE_SYNC();                                                                               \
    float bc[16];                                                                                  \
    _Pragma("unroll") for (int s = 0; s < 16; ++s) bc[s] = bw[s];                                  \
    load_w((k) + 1 < k_vol ? (k) + 1 : (k));                                                       \
    PCC_GATHER(GN, (cur) ^ 1, cnt_next);                                                           \
    _Pragma("unroll") for (int grp = 0; grp < NG; ++grp) {                                         \
      if (grp * 32 < cnt_cur) {                                                                    \
        const int arow = (int)slot_row[cur][grp * 32 + i];                                         \
        float* ap = &acc_lds[arow * AP + h * 4];                                                   \
        f32x16 acc;                                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
          const float4 c4 = *reinterpret_cast<const float4*>(ap + j * 8);                          \
          acc[4 * j + 0] = c4.x; acc[4 * j + 1] = c4.y; acc[4 * j + 2] = c4.z; acc[4 * j + 3] = c4.w; \
        }                                                                                          \
        /* lane (i,h) holds x[slot i][16h .. 16h+15]; the MFMA wants x[slot i][2s+h]: swapping the */ \
        /* upper half of the even-j register with the lower half of the odd-j one gives both       */ \
        /* ci = 2t+h (s = t) and ci = 16+2t+h (s = 8+t) without touching LDS                       */ \
        float xv[16];                                                                              \
        _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                         \
          const u32x2 p0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(GC[grp][it].x),        \
                                                            __float_as_uint(GC[grp][it].y), false, false); \
          const u32x2 p1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(GC[grp][it].z),        \
                                                            __float_as_uint(GC[grp][it].w), false, false); \
          xv[2 * it] = __uint_as_float(p0[0]);     xv[8 + 2 * it] = __uint_as_float(p0[1]);        \
          xv[2 * it + 1] = __uint_as_float(p1[0]); xv[8 + 2 * it + 1] = __uint_as_float(p1[1]);    \
        }                                                                                          \
        PCC_PRIO_BEGIN();                                                                          \
        _Pragma("unroll") for (int s = 0; s < 16; ++s) {                                           \
          PCC_PRIO_MID(s);                                                                         \
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[s], xv[s], acc, 0, 0, 0);                  \
        }                                                                                          \
        PCC_PRIO_END();                                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                              \
          *reinterpret_cast<float4*>(ap + j * 8) =                                                 \
              make_float4(acc[4 * j + 0], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);         \
        PCC_WAVE_SYNC();                                                                           \
      }                                                                                            \
    }                                                                                              \
    cnt_cur = cnt_next;                                                                            \
  } while (0)

  load_nb(0);
  int cnt_cur = compact(0);
  load_nb(1);
  PCC_WAVE_SYNC();
  PCC_GATHER(gA, 0, cnt_cur);
  load_w(0);

  // D[co][slot] += sum_ci W[ci][co] * x[slot][ci], 2 ci per MFMA, ci ascending
  for (int k = 0; k < k_vol; k += 2) {
    PCC_STEP(gA, gB, 0, k);
    if (k + 1 < k_vol) PCC_STEP(gB, gA, 1, k + 1);
  }
#undef PCC_STEP
#undef PCC_GATHER

  // ---- epilogue: the window's rows are contiguous in `out`: coalesced 16-B stores
#pragma unroll
  for (int it = 0; it < R / 8; ++it) {
    const int r = it * 8 + grow;
    float4 v = *reinterpret_cast<const float4*>(&acc_lds[r * AP + chunk * 4]);
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (row0 + r < n_out) *reinterpret_cast<float4*>(out + (row0 + r) * COUT + col0 + chunk * 4) = v;
  }
  if constexpr (HEAD) {
#pragma unroll
    for (int c = 0; c < RCH; ++c) {
      const int r = c * 64 + lane;
      float hv = head_b[0];
#pragma unroll
      for (int ch = 0; ch < 32; ++ch) {
        float v = acc_lds[(r < R ? r : 0) * AP + ch];
        if (relu) v = fmaxf(v, 0.f);
        hv = fmaf(v, head_w[ch], hv);
      }
      if (row0 + r < n_out && r < R) head_out[row0 + r] = hv;
    }
  }
#undef PCC_WAVE_SYNC
}


// ---------------------------------------------------------------------------------------------------------------
// Four windows per workgroup, weights shared through LDS.
//
// In the kernel above every wave fetches the 4 KB weight matrix of every offset for itself: as many bytes through
// the texture path as its useful gathered rows.  With the weight loads removed that kernel runs 14 % faster
// (ablation, tools/bench_conv.py: 1.61 -> 1.38 ms on the 3.26M-row layer).  Here a workgroup is four such waves
// (four consecutive 64-row windows); each wave fetches a QUARTER of W[k+2] (one dwordx4 per lane), the quarters meet
// in a double-buffered LDS tile, and every wave reads its MFMA operands of W[k] from there.  One s_barrier per
// offset; it waits for LDS only (lgkmcnt), never for the gathers in flight.
//   iteration k:  bc <- wbuf[k & 1]            (W[k]: complete since the barrier that closed iteration k-1)
//                 wbuf[(k+1) & 1] <- wq        (own quarter of W[k+1]; every wave read W[k-1] from it before that barrier)
//                 wq <- global quarter of W[k+2]
//                 ... compact / gather / contract as above ...
//                 barrier
template <bool HEAD, bool UP = false>
__global__ __launch_bounds__(256) void k_gconv_mfma_compact_w4(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int k_vol, int64_t pitch,
    int64_t n_out, const float* __restrict__ w, const float* __restrict__ bias, int relu,
    float* __restrict__ out, const float* __restrict__ head_w, const float* __restrict__ head_b,
    float* __restrict__ head_out) {
  constexpr int RCH = 1;
  constexpr int R = 64;
  constexpr int AP = 36;
  __shared__ __attribute__((aligned(16))) float acc_all[4][(R + 1) * AP];
  __shared__ int32_t slot_in_all[4][2][R];
  __shared__ uint8_t slot_row_all[4][2][R];
  __shared__ __attribute__((aligned(16))) float wbuf[2][32 * 32];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  PCC_PRIO_KERNEL();
  float* const acc_lds = acc_all[wave];
  int32_t(*const slot_in)[R] = slot_in_all[wave];
  uint8_t(*const slot_row)[R] = slot_row_all[wave];
  // XCD-aware order over workgroups (see above); a workgroup owns four consecutive windows.  A wave whose window
  // lies past the end keeps running (all rows absent): the others need its quarter of the weights and its barriers
  const int64_t wg = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t row0 = (wg * 4 + wave) * R;
  const int i = lane & 31, h = lane >> 5;
  const int grow = lane >> 3, chunk = lane & 7;

  {
    const float4 b4 = make_float4(bias[chunk * 4], bias[chunk * 4 + 1], bias[chunk * 4 + 2], bias[chunk * 4 + 3]);
#pragma unroll
    for (int it = 0; it < R / 8; ++it)
      *reinterpret_cast<float4*>(&acc_lds[(it * 8 + grow) * AP + chunk * 4]) = b4;
    if (lane < 8) *reinterpret_cast<float4*>(&acc_lds[R * AP + lane * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  int32_t nbreg[RCH];
  auto load_nb = [&](int k) {
    const int64_t r = row0 + lane;
    const int kk = k < k_vol ? k : k_vol - 1;
    const int64_t rc = r < n_out ? r : n_out - 1;
    if constexpr (UP) {
      const int o = (int)(rc & 7);
      const int tx = ((o >> 2) & 1) + (kk / 9) - 1, ty = ((o >> 1) & 1) + ((kk / 3) % 3) - 1, tz = (o & 1) + (kk % 3) - 1;
      const int kp = ((tx + 2) >> 1) * 9 + ((ty + 2) >> 1) * 3 + ((tz + 2) >> 1);
      const int op = ((tx & 1) << 2) | ((ty & 1) << 1) | (tz & 1);
      const int32_t pr = nbr[(int64_t)kp * pitch + (rc >> 3)];
      nbreg[0] = (k < k_vol && r < n_out && pr >= 0) ? ((pr << 3) | op) : -1;
    } else {
      const int32_t v = nbr[(int64_t)kk * pitch + rc];
      nbreg[0] = (k < k_vol && r < n_out) ? v : -1;
    }
  };
  auto compact = [&](int b) -> int {
    const bool p = nbreg[0] >= 0;
    const unsigned long long bal = __ballot(p);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
    const int cnt = __popcll(bal);
    if (p) {
      slot_in[b][rank] = nbreg[0];
      slot_row[b][rank] = (uint8_t)lane;
    }
    const int sl = cnt + lane;
    if (sl < R) {
      slot_in[b][sl] = 0;
      slot_row[b][sl] = (uint8_t)R;
    }
    return cnt;
  };

  constexpr int NG = R / 32;
  float4 gA[NG][4], gB[NG][4];
  // this thread's 16 B of a weight matrix: quarter `wave`, float4 index `lane` of it
  float4 wq;
  auto load_wq = [&](int k) {
    const int kk = k < k_vol ? k : k_vol - 1;
    wq = *reinterpret_cast<const float4*>(w + (int64_t)kk * 1024 + threadIdx.x * 4);
  };

#define PCC_WAVE_SYNC()                                      \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)
// workgroup barrier that waits for this wave's LDS traffic only: loads in flight (gathers, next weights, next
// neighbour indices) stay in flight across it
#define PCC_WG_SYNC()                                        \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */     \
    __builtin_amdgcn_s_barrier();                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)
#define PCC_GATHER(G, b)                                                                           \
  do {                                                                                             \
    _Pragma("unroll") for (int grp = 0; grp < NG; ++grp) {                                         \
      const float* xr = in + (int64_t)slot_in[b][grp * 32 + i] * 32 + h * 16;                      \
      _Pragma("unroll") for (int it = 0; it < 4; ++it)                                             \
        G[grp][it] = *reinterpret_cast<const float4*>(xr + it * 4);                                \
    }                                                                                              \
  } while (0)
#define PCC_STEP(GC, GN, cur, k)                                                                   \
  do {                                                                                             \
    PCC_STAMP(0);                                                                                  \
    const int cnt_next = compact((cur) ^ 1);                                                       \
    PCC_STAMP(1);                                                                                  \
    load_nb((k) + 2);                                                                              \
    PCC_STAMP(2);                                                                                  \
    *reinterpret_cast<float4*>(&wbuf[((k) + 1) & 1][threadIdx.x * 4]) = wq;                        \
    load_wq((k) + 2);                                                                              \
    PCC_WAVE_SYNC();                                                                               \
    PCC_GATHER(GN, (cur) ^ 1);                                                                     \
    float bc[16]; /* read behind the gathers' issue: its LDS latency overlaps the accumulator reads */ \
    _Pragma("unroll") for (int s = 0; s < 16; ++s) bc[s] = wbuf[(k) & 1][(2 * s + h) * 32 + i];    \
    PCC_STAMP(3);                                                                                  \
    _Pragma("unroll") for (int grp = 0; grp < NG; ++grp) {                                         \
      if (grp * 32 < cnt_cur) {                                                                    \
        const int arow = (int)slot_row[cur][grp * 32 + i];                                         \
        float* ap = &acc_lds[arow * AP + h * 4];                                                   \
        f32x16 acc;                                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
          const float4 c4 = *reinterpret_cast<const float4*>(ap + j * 8);                          \
          acc[4 * j + 0] = c4.x; acc[4 * j + 1] = c4.y; acc[4 * j + 2] = c4.z; acc[4 * j + 3] = c4.w; \
        }                                                                                          \
        PCC_STAMP(4);                                                                              \
        float xv[16];                                                                              \
        _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                         \
          const u32x2 p0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(GC[grp][it].x),        \
                                                            __float_as_uint(GC[grp][it].y), false, false); \
          const u32x2 p1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(GC[grp][it].z),        \
                                                            __float_as_uint(GC[grp][it].w), false, false); \
          xv[2 * it] = __uint_as_float(p0[0]);     xv[8 + 2 * it] = __uint_as_float(p0[1]);        \
          xv[2 * it + 1] = __uint_as_float(p1[0]); xv[8 + 2 * it + 1] = __uint_as_float(p1[1]);    \
        }                                                                                          \
        PCC_STAMP(5); /* gathered rows arrived (vmcnt), operands shaped */                        \
        PCC_PRIO_BEGIN();                                                                          \
        _Pragma("unroll") for (int s = 0; s < 16; ++s) {                                           \
          PCC_PRIO_MID(s);                                                                         \
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[s], xv[s], acc, 0, 0, 0);                  \
        }                                                                                          \
        PCC_PRIO_END();                                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                              \
          *reinterpret_cast<float4*>(ap + j * 8) =                                                 \
              make_float4(acc[4 * j + 0], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);         \
        PCC_WAVE_SYNC();                                                                           \
        PCC_STAMP(6);                                                                              \
      }                                                                                            \
    }                                                                                              \
    cnt_cur = cnt_next;                                                                            \
    PCC_STAMP(7);                                                                                  \
    PCC_WG_SYNC();                                                                                 \
    PCC_STAMP(8);                                                                                  \
  } while (0)

#if PCC_CONV_STAMP
  unsigned long long st_sum[PCC_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
  load_nb(0);
  int cnt_cur = compact(0);
  load_nb(1);
  load_wq(0);
  *reinterpret_cast<float4*>(&wbuf[0][threadIdx.x * 4]) = wq;
  load_wq(1);
  PCC_WG_SYNC();
  PCC_GATHER(gA, 0);

  for (int k = 0; k < k_vol; k += 2) {
    PCC_STEP(gA, gB, 0, k);
    if (k + 1 < k_vol) PCC_STEP(gB, gA, 1, k + 1);
  }
#undef PCC_STEP
#undef PCC_GATHER
  PCC_STAMP(9);
#if PCC_CONV_STAMP
  if (lane == 0 && UP && blockIdx.x >= 4096 && blockIdx.x < 4096 + 1024) {
    for (int q = 0; q < PCC_NSTAMP; ++q) pcc_stamp_buf[((blockIdx.x - 4096) * 4 + wave) * PCC_NSTAMP + q] = st_sum[q];
  }
#endif

#pragma unroll
  for (int it = 0; it < R / 8; ++it) {
    const int r = it * 8 + grow;
    float4 v = *reinterpret_cast<const float4*>(&acc_lds[r * AP + chunk * 4]);
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (row0 + r < n_out) *reinterpret_cast<float4*>(out + (row0 + r) * 32 + chunk * 4) = v;
  }
  if constexpr (HEAD) {
    const int r = lane;
    float hv = head_b[0];
#pragma unroll
    for (int ch = 0; ch < 32; ++ch) {
      float v = acc_lds[r * AP + ch];
      if (relu) v = fmaxf(v, 0.f);
      hv = fmaf(v, head_w[ch], hv);
    }
    if (row0 + r < n_out) head_out[row0 + r] = hv;
  }
#undef PCC_WAVE_SYNC
#undef PCC_WG_SYNC
}
