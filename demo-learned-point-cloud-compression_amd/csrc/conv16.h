// conv16.h — row-compacting 32 -> 32 gather-convolution on 16-slot items (included by conv.hip).
//
// Arithmetic contract of conv.hip (out = bias; for k ascending over PRESENT neighbours, ci ascending:
// out = fmaf(x, w, out); v_mfma_f32_16x16x4_f32 is that chain four ci at a time), different shape of the work:
//
//   * the rows of a 64-row window that have offset k are packed into ITEMS of 16 slots (not groups of 32): a partly
//     filled item wastes at most 15 slots instead of 31 — issued / useful matrix work on bench.py's dominant layer
//     1.18 instead of 1.38 (simulated on its geometry and counted by SQ_INSTS_MFMA);
//   * an item is two independent chains of 8 v_mfma_f32_16x16x4_f32 (output channels 0..15 and 16..31) issued
//     alternately: the 40-cycle dependent latency of one chain hides behind the other chain's 32-cycle issue;
//   * one wave per workgroup and no workgroup barrier: a wave reads the weights of an offset as its MFMA operands
//     straight from a pre-swizzled copy (pcc_conv16_swizzle: [k][lane][16] floats — four coalesced dwordx4 per
//     lane and offset instead of sixteen dword loads), so waves never wait for each other;
//   * nothing is waited for right after it is issued.  In the stamped build of the 32-slot, four-windows-per-workgroup
//     kernel this one replaces (round 1; tools/stamp_conv.py) a wave spent 17 % of its time waiting for the neighbour index it had just asked for,
//     16 % in the per-offset barrier, 10 % in the two dependent LDS round trips slot -> row -> accumulator and 12 %
//     around the slot-list read in front of the gathers, against 31 % in its matrix chains.  Here the neighbour index
//     of offset k+2 is only CONSUMED one step later (the child-index arithmetic of the UP form is deferred to the
//     compaction), the slot records of offset k+1 (input row, accumulator row) are read into registers right after
//     the compaction, the gathered rows of offset k+1's item g are requested into the registers item g of offset k
//     has just been consumed from (a whole offset of prefetch distance on ONE register set), the accumulator tile of
//     item g+1 is read while the chains of item g run (items of one offset touch disjoint rows), and the bookkeeping of
//     a step is written INTO the chains of its first item (one basic block): it issues in the shadow of the MFMAs.
//
// Lane (n, q) = (lane & 15, lane >> 4).  MFMA operands (D = A x B, A = W^T block 16 co x 4 ci, B = x^T block 4 ci x
// 16 slots): A lane (m, q) holds W[4s + q][m (+16)], B lane (n, q) holds x[slot n][4s + q], D lane (n, q) holds
// channels 4q .. 4q+3 (+16) of slot n — a row's accumulator is two 16-B pieces, one per 16-channel plane.
//
// PERM: the input rows are stored channel-permuted, position 8q + s holding channel 4s + q: lane (n, q) reads its
// eight B operands with two dwordx4 and no shaping.  The native decoder stores the output of the generative up
// stages that way (permuted weight columns, codec.hip).  Natural layout (PERM = false): lane (n, q) reads channels
// 8q .. 8q+7 and the four q-lanes of a slot transpose two 4 x 4 blocks with v_permlane32_swap / v_permlane16_swap.
//
// HEAD (the conv3 + occupancy head layers of g_s): the neighbours are visited SIBLINGS FIRST (include/pcc.h) — here as two
// passes over the offsets, the first taking the neighbours inside the output row's own block of 8 rows (UP: the
// offsets that stay inside the row's parent), the second the others.  The shipped decoder runs these layers on
// k_gconv_up (convup.h), which contracts the sibling pass as a dense product; this form serves explicit rule books
// (pcc_sparse_conv_head) and tensors beyond k_gconv_up's 32-bit offsets.
//
// COUT = 64: a 32 -> 64 layer runs as grid.y = 2, workgroup (x, y) producing columns [32 y, 32 y + 32) of window x
// (the h_s output layer, evaluated at the latent's rows only).
//
// Measured on bench.py's dominant launch (3,262,640 candidate rows, tools/bench_conv.py, same box): the 32-slot kernel
// with four windows per workgroup 1.54 ms, this kernel 1.29 ms on natural rows and 1.23 ms on permuted rows (52 % of
// the f32 matrix peak).  What was tried on top and changed nothing or made it slower (tools/ab_conv.sh): write-back of
// an item delayed behind the next item's first MFMA pair, register renaming instead of copies between steps,
// wave priorities (raised during a chain, lowered during a chain, static per workgroup: +11 .. +16 %), four waves
// per SIMD (123 registers, no spill: +16 %), two or fewer waves per SIMD (+9 .. +22 %), weights or gathered rows
// served from L1-resident addresses (+5 .. +12 %: the launch is not waiting for memory).
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));


// Diagnostic build only (-DPCC_CONV_STAMP=1; results are unchanged, timing is not): cycle stamps (s_memtime) at the
// phase boundaries of an offset step, summed per wave over its steps and written to a buffer of their own
// (pcc_debug_stamps, tools/stamp_conv.py).  MI355X_MICROARCH.md, "in-kernel stamps".
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

#define PCC16_SYNC()                                         \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

// (k / div) % 3 for k = 0 .. 26 as 2-bit fields of one word
__host__ __device__ constexpr uint64_t digits3(int div) {
  uint64_t v = 0;
  for (int k = 0; k < 27; ++k) v |= (uint64_t)((k / div) % 3) << (2 * k);
  return v;
}

// weights [k_vol][32][cout] (ci major) -> [k_vol][cout / 32][64 lanes][16]: lane (m, q) of column half y:
// s = 0..7: W[4s+q][32y + m], then W[4s+q][32y + 16 + m]
__global__ __launch_bounds__(256) void k_conv16_swizzle(const float* __restrict__ w, int k_vol, int cout,
                                                        float* __restrict__ wsw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ny = cout >> 5;
  if (t >= k_vol * ny * 1024) return;
  const int k = t / (ny * 1024), y = (t >> 10) % ny, l = (t >> 4) & 63, e = t & 15;
  const int m = l & 15, q = l >> 4, s = e & 7, hi = e >> 3;
  wsw[t] = w[(k * 32 + 4 * s + q) * cout + 32 * y + 16 * hi + m];
}

// R (rows of a window, 64 or 32): a launch of a few hundred 64-row windows is one partial round of waves on 1024 SIMDs
// — it lasts as long as ONE window (27 dependent steps) whatever its size.  Such launches run on 32-row windows: twice
// the waves, about half the records (and items) per step and wave; lanes 32..63 own no row and only help to carry.
//
// WIDE = false (every launch whose input tensor is below 4 GB, i.e. < 2^25 rows, and whose parent rule book has a pitch
// below 2^24): a slot record holds the BYTE offset of its input row and the LDS byte address of its accumulator row, so
// a gather is `global_load v, voffset, s[base]` on a 32-bit offset (one v_or per gather instead of a sign extension, a
// 64-bit shift and a 64-bit add), an accumulator address one v_xor with the lane's piece (instead of five instructions,
// twice per item), and the parent-book index of the UP form one v_mul_u32_u24 (instead of a 64-bit product built from
// three quarter-rate multiplies).  f32 MFMAs share the vector ALUs, so every vector instruction saved is
// matrix time gained.  WIDE = true keeps 64-bit row arithmetic for tensors beyond those bounds.
template <bool HEAD, bool UP, bool PERM, int COUT = 32, int R = 64, bool WIDE = false>
__global__ __launch_bounds__(64) void k_gconv16(
    const float* __restrict__ in, const int32_t* __restrict__ nbr, int k_vol, int64_t pitch,
    int64_t n_out, const float* __restrict__ wsw, const float* __restrict__ bias, int relu,
    float* __restrict__ out, const float* __restrict__ head_w, const float* __restrict__ head_b,
    float* __restrict__ head_out, const float* __restrict__ rgb_w = nullptr, const float* __restrict__ rgb_b = nullptr,
    float* __restrict__ rgb_out = nullptr, uint32_t in_bytes = 0u) {
  static_assert(R == 64 || R == 32, "window of 64 or 32 rows");
  // Accumulators in LDS: two planes (channels 0..15, 16..31) of 16-float rows; the 16-B piece q of a row sits at
  // position (q + 2 (row >> 2)) & 3 of its plane row.  The hardware serves a ds_read_b128 / ds_write_b128 in groups of 16
  // lanes that hold 16 different slots and two values of q.  The placement makes an item of 16 CONSECUTIVE rows
  // conflict-free; the rows of a compacted list are not consecutive, and the counters say so: SQ_LDS_BANK_CONFLICT /
  // SQ_LDS_IDX_ACTIVE = 52 % for this kernel (profiles/r02q_pmc_sq_conv.json), 58 % for k_gconv_up, which uses the same
  // placement (profiles/r03p_pmc_sq_conv.json) — any fixed placement leaves 16 arbitrary rows to collide by chance.  It is
  // not what limits either kernel (no accumulator traffic at all: no gain).
  constexpr int HP = (R + 1) * 16;   // floats per plane (row R = sink of the pad slots)
  constexpr int NI = R / 16;   // items an offset can have
  __shared__ __attribute__((aligned(16))) float acc_lds[2 * HP];
  __shared__ __attribute__((aligned(8))) int2 rec[2][64];                // slot -> (input row [WIDE] or its byte offset, accumulator row address acc_row); every LANE writes one

  static_assert(COUT == 32 || (COUT == 64 && !HEAD), "the fused head reads all channels of a row");
  const int lane = threadIdx.x;
  const int ny = COUT / 32, ycol = COUT == 32 ? 0 : (int)blockIdx.y, col0 = 32 * ycol;
  // XCD-aware window order: workgroups are dealt round-robin to the 8 XCDs, so workgroup b takes window (b % 8) *
  // (grid / 8) + b / 8: every XCD walks one contiguous eighth of the Morton-sorted rows, and the neighbour rows that
  // adjacent windows share are fetched into ONE L2 instead of eight
  const int64_t window = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t row0 = window * R;
  if (row0 >= n_out) return;
  const int n = lane & 15, q = lane >> 4;
  const int grow = lane >> 3, chunk = lane & 7;

  // float index of channel block (plane `half`, piece qq) of `row`
  auto acc_at = [&](int half, int row, int qq) -> int { return half * HP + row * 16 + (((qq + 2 * (row >> 2)) & 3) << 2); };
  // the same as a byte address in plane 0, in two parts: (qq + 2 (row >> 2)) & 3 == qq ^ ((row >> 1) & 2), so
  // 4 * acc_at(0, row, qq) == acc_row(row) ^ (qq << 4): the row's part travels in the slot record, the lane's is a constant
  auto acc_row = [](int row) -> int { return row * 64 + (((row >> 1) & 2) << 4); };
  const int a_own = acc_row(lane), a_sink = acc_row(R), q16 = q << 4;
  const uint32_t qoff = (uint32_t)q * 32u;   // bytes: channels 8q .. of an input row
  // !WIDE: the rows are gathered with raw buffer loads over [in, in + in_bytes).  A pad slot's offset lies beyond the
  // buffer: such a lane returns zeros without a fetch, and the gather of an item no row of the next offset needs (issued
  // all the same, to keep the compiler's vmcnt counts exact) costs a quarter of a real one (tools/micro/mix.hip)
  constexpr uint32_t kPadOff = 0xFFFFFF80u;
  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00027000);
  {  // accumulators start at the bias: lane (grow, chunk) fills piece `chunk` (channels 4 chunk ..) of rows grow + 8 it
    const float* bp = bias + col0 + chunk * 4;
    const float4 b4 = make_float4(bp[0], bp[1], bp[2], bp[3]);
#pragma unroll
    for (int it = 0; it < R / 8; ++it)
      *reinterpret_cast<float4*>(&acc_lds[acc_at(chunk >> 2, it * 8 + grow, chunk & 3)]) = b4;
    if (lane < 8) *reinterpret_cast<float4*>(&acc_lds[acc_at(lane >> 2, R, lane & 3)]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- neighbour index of this lane's row: requested two offsets ahead, turned into a row one offset later
  const int64_t r_own = row0 + lane;
  const bool row_ok = lane < R && r_own < n_out;
  const int64_t rc = row_ok ? r_own : n_out - 1;
  const uint32_t rc3 = (uint32_t)(rc >> 3);
  // UP: byte d of an axis word = (parent-offset digit * weight of the axis in kp) | (octant bit of the axis << 5) for
  // a step of d - 1 along the axis from this lane's octant bit `ob`
  auto up_axis = [](int ob, int weight, int opbit) -> uint32_t {
    uint32_t v = 0u;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int t = ob + d - 1;
      v |= (uint32_t)((((t + 2) >> 1) * weight) | ((t & 1) ? (opbit << 5) : 0)) << (8 * d);
    }
    return v;
  };
  const int oct = (int)(rc & 7);
  const uint32_t up_x = up_axis((oct >> 2) & 1, 9, 4), up_y = up_axis((oct >> 1) & 1, 3, 2), up_z = up_axis(oct & 1, 1, 1);
  constexpr bool SIB = HEAD;   // siblings-first order: offsets 0 .. k_vol-1 twice (virtual offsets 0 .. 2 k_vol - 1)
  const int kv_tot = SIB ? 2 * k_vol : k_vol;
  int32_t nb_raw = -1;   // UP: row of the parent-level neighbour; else the neighbour row itself
  int nb_op = 0;         // UP: octant of the neighbour inside that parent
  bool nb_live = false;  // offset exists and the lane owns a row
  bool nb_first = true;  // SIB: the request belongs to the sibling pass
  bool nb_own = false;   // SIB && UP: the offset stays inside the row's parent
  auto request_nb = [&](int k) {
    const int kc = k < kv_tot ? k : kv_tot - 1;
    nb_first = !SIB || kc < k_vol;
    const int kk = nb_first ? kc : kc - k_vol;
    nb_live = k < kv_tot && row_ok;
    if constexpr (UP) {
      // t = octant bit + step of the offset along an axis, in -1 .. 2: the parent-level offset is (t + 2) >> 1, the octant
      // bit of the neighbour inside that parent t & 1.  Per axis the three answers sit in one register (up_axis below)
      // and the offset's step selects a byte: three bit-field extracts and an add instead of a dozen instructions.
      // digits of kk in base 3 from 2-bit fields of three constants (a shift and a mask instead of two divisions)
      constexpr uint64_t DX = digits3(9), DY = digits3(3), DZ = digits3(1);
      const uint32_t sh = 2u * (uint32_t)kk;
      const uint32_t comb = __builtin_amdgcn_ubfe(up_x, ((uint32_t)(DX >> sh) & 3u) << 3, 8u) +
                            __builtin_amdgcn_ubfe(up_y, ((uint32_t)(DY >> sh) & 3u) << 3, 8u) +
                            __builtin_amdgcn_ubfe(up_z, ((uint32_t)(DZ >> sh) & 3u) << 3, 8u);
      const int kp = (int)(comb & 31u);
      nb_op = (int)(comb >> 5);
      nb_own = kp == 13;
      if constexpr (WIDE) {
        nb_raw = nbr[(int64_t)kp * pitch + (rc >> 3)];
      } else {
        // pitch < 2^24 and kp < 27: the index and its byte offset fit 32 bits
        const uint32_t off = (__umul24((uint32_t)kp, (uint32_t)pitch) + rc3) << 2;
        nb_raw = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(nbr) + off);
      }
    } else {
      nb_raw = nbr[(int64_t)kk * pitch + rc];
    }
  };
  // pack the rows that have the requested offset into slot records `b`; returns their count.  Every lane writes
  // exactly one record (no divergent branch): a present row the record of its rank, an absent one a pad record (row 0 of
  // `in`, accumulated into the sink row) in slot count + number of absent lanes in front of it
  auto compact = [&](int b) -> int {
    bool p = nb_live && nb_raw >= 0;
    if constexpr (SIB) p = p && ((UP ? nb_own : (nb_raw >> 3) == (int32_t)rc3) == nb_first);
    const int32_t src = UP ? ((nb_raw << 3) | nb_op) : nb_raw;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(p);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
    const int cnt = __popcll(bal);
    const int32_t sx = WIDE ? src : (int32_t)((uint32_t)src << 7);
    rec[b][p ? rank : cnt + lane - rank] = p ? make_int2(sx, a_own) : make_int2(WIDE ? 0 : (int32_t)kPadOff, a_sink);
    return cnt;
  };

  float4 G[NI][2];              // gathered rows (B operands) of the items of one offset
  float4 W0[4], W1[4];          // A operands of the even / odd offsets
  int rin[NI], ra0[NI], ra1[NI];  // input rows of the next offset's slots; accumulator rows (byte offsets) even / odd
  const uint32_t lane64 = (uint32_t)lane * 64u;
  auto load_w = [&](float4 (&W)[4], int k) {
    const int kc = k < kv_tot ? k : kv_tot - 1;
    const int kk = (SIB && kc >= k_vol) ? kc - k_vol : kc;
    const char* base = reinterpret_cast<const char*>(wsw + ((int64_t)kk * ny + ycol) * 1024);   // uniform
    const float4* p = reinterpret_cast<const float4*>(base + lane64);
#pragma unroll
    for (int j = 0; j < 4; ++j) W[j] = p[j];
  };
  auto read_records = [&](int b, int (&racc)[NI]) {
#pragma unroll
    for (int g = 0; g < NI; ++g) {
      const int2 r = rec[b][g * 16 + n];
      rin[g] = r.x;
      racc[g] = r.y;
    }
  };
  auto gather = [&](int g) {
    if constexpr (WIDE) {
      const float* xr = in + (int64_t)rin[g] * 32 + q * 8;
      G[g][0] = *reinterpret_cast<const float4*>(xr);
      G[g][1] = *reinterpret_cast<const float4*>(xr + 4);
    } else {
      const uint32_t off = (uint32_t)rin[g] | qoff;
      const auto r0 = __builtin_amdgcn_raw_buffer_load_b128(in_rs, off, 0, 0);
      const auto r1 = __builtin_amdgcn_raw_buffer_load_b128(in_rs, off + 16u, 0, 0);
      G[g][0] = make_float4(__uint_as_float(r0[0]), __uint_as_float(r0[1]), __uint_as_float(r0[2]), __uint_as_float(r0[3]));
      G[g][1] = make_float4(__uint_as_float(r1[0]), __uint_as_float(r1[1]), __uint_as_float(r1[2]), __uint_as_float(r1[3]));
    }
  };
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
  // B operands of item g in MFMA order
  auto shape = [&](int g, float (&xv)[8]) {
    if constexpr (PERM) {
      xv[0] = G[g][0].x; xv[1] = G[g][0].y; xv[2] = G[g][0].z; xv[3] = G[g][0].w;
      xv[4] = G[g][1].x; xv[5] = G[g][1].y; xv[6] = G[g][1].z; xv[7] = G[g][1].w;
    } else {
      // lane (n, t) holds channels 8t + e: xv[2t'] of lane q must become channel 8t' + q, xv[2t'+1] channel 8t' + 4 + q:
      // transpose the 4 x 4 blocks (lane t, element e) of the low and of the high four elements
      unsigned m[2][4] = {{__float_as_uint(G[g][0].x), __float_as_uint(G[g][0].y), __float_as_uint(G[g][0].z), __float_as_uint(G[g][0].w)},
                          {__float_as_uint(G[g][1].x), __float_as_uint(G[g][1].y), __float_as_uint(G[g][1].z), __float_as_uint(G[g][1].w)}};
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
  };

#if PCC_CONV_STAMP
  unsigned long long st_sum[PCC_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last;
#endif
  int cnt_cur;
  // One offset step.  Wc / rc: operands and accumulator rows of offset k; Wn / rn take those of offset k+1.
  // Item g: [tile of item g+1 requested] chains of item g, with the write-back of item g-1 behind their first pair
  // (its results have long arrived by then; straight behind its own chains it would wait out their latency);
  // the last item of the step writes itself back.  Items of one offset touch disjoint rows, so the order of these
  // LDS accesses inside a step is free; across steps program order keeps every write in front of the next read.
  auto step = [&](int k, float4 (&Wc)[4], float4 (&Wn)[4], int (&rc_)[NI], int (&rn)[NI]) {
    PCC_STAMP(0);
    f32x4 lo0, hi0, lo1, hi1;
    const float wl[8] = {Wc[0].x, Wc[0].y, Wc[0].z, Wc[0].w, Wc[1].x, Wc[1].y, Wc[1].z, Wc[1].w};
    const float wh[8] = {Wc[2].x, Wc[2].y, Wc[2].z, Wc[2].w, Wc[3].x, Wc[3].y, Wc[3].z, Wc[3].w};
    // Item 0 is unconditional (an offset without a present row runs it on pad slots into the sink row), so that its
    // chains and the bookkeeping of the step — compaction of offset k+1, request of offset k+2's indices, weights of
    // offset k+1, slot records — are ONE basic block, written interleaved: the bookkeeping issues in the shadow of the
    // MFMAs instead of in front of them (-9 % on the dominant launch).
    acc_read(rc_[0], lo0, hi0);
    acc_read(rc_[1], lo1, hi1);   // item 1's tile (the sink row's when there is no item 1)
    float xv0[8];
    shape(0, xv0);
    lo0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[0], xv0[0], lo0, 0, 0, 0);
    hi0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[0], xv0[0], hi0, 0, 0, 0);
    const int cnt_next = compact((k + 1) & 1);   // offset k+1 (no row has an offset past the last)
    lo0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[1], xv0[1], lo0, 0, 0, 0);
    hi0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[1], xv0[1], hi0, 0, 0, 0);
    request_nb(k + 2);
    lo0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[2], xv0[2], lo0, 0, 0, 0);
    hi0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[2], xv0[2], hi0, 0, 0, 0);
    load_w(Wn, k + 1);
    lo0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[3], xv0[3], lo0, 0, 0, 0);
    hi0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[3], xv0[3], hi0, 0, 0, 0);
    PCC16_SYNC();
    read_records((k + 1) & 1, rn);
#pragma unroll
    for (int s = 4; s < 8; ++s) {
      lo0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xv0[s], lo0, 0, 0, 0);
      hi0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xv0[s], hi0, 0, 0, 0);
    }
    PCC_STAMP(1);
    if (cnt_cur <= 16) acc_write(rc_[0], lo0, hi0);
    // The gathers are unconditional (an item without a present row reads row 0, one cache line for the whole wave):
    // loads under a branch would make the compiler's s_waitcnt vmcnt counts inexact, and an inexact count in front
    // of an item's chains waits for the neighbour index and the weights requested at the top of this very step
    gather(0);
    // (the tile read for item g+1 is unconditional: when the offset has no item g+1 its slots hold pad records, whose
    // accumulator row is the sink row)
    // Item g >= 1: [write-back of item g-1 behind the first MFMA pair, tile of item g+1 requested] chains of item g;
    // the last item of the step writes itself back.  Items of one offset touch disjoint rows, so the order of these
    // LDS accesses inside a step is free; across steps program order keeps every write in front of the next read.
#define PCC16_ITEM(g, LO, HI, PLO, PHI)                                                              \
    if (cnt_cur > 16 * (g)) {                                                                          \
      const bool more = (g) + 1 < NI && cnt_cur > 16 * ((g) + 1);                                      \
      float xv[8];                                                                                     \
      shape(g, xv);                                                                                    \
      LO = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[0], xv[0], LO, 0, 0, 0);                            \
      HI = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[0], xv[0], HI, 0, 0, 0);                            \
      acc_write(rc_[(g) - 1], PLO, PHI);                                                               \
      if constexpr ((g) + 1 < NI) acc_read(rc_[(g) + 1 < NI ? (g) + 1 : 0], PLO, PHI);                 \
      _Pragma("unroll") for (int s = 1; s < 8; ++s) {                                                  \
        LO = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xv[s], LO, 0, 0, 0);                          \
        HI = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xv[s], HI, 0, 0, 0);                          \
      }                                                                                                \
      if (!more) acc_write(rc_[g], LO, HI);                                                            \
    }                                                                                                  \
    gather(g)
    PCC16_ITEM(1, lo1, hi1, lo0, hi0);
    if constexpr (NI > 2) {
      PCC16_ITEM(2, lo0, hi0, lo1, hi1);
      PCC16_ITEM(3, lo1, hi1, lo0, hi0);
    }
#undef PCC16_ITEM
    PCC_STAMP(2);
    cnt_cur = cnt_next;
  };

  // ---- prologue: offset 0 compacted and gathered, offset 1 requested
  request_nb(0);
  cnt_cur = compact(0);
  request_nb(1);
  load_w(W0, 0);
  PCC16_SYNC();
  read_records(0, ra0);
#pragma unroll
  for (int g = 0; g < NI; ++g) gather(g);
#if PCC_CONV_STAMP
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
  for (int k = 0; k < kv_tot; k += 2) {
    step(k, W0, W1, ra0, ra1);
    if (k + 1 < kv_tot) step(k + 1, W1, W0, ra1, ra0);
  }
  PCC16_SYNC();
#if PCC_CONV_STAMP
  if (lane == 0 && blockIdx.x >= 16384 && blockIdx.x < 16384 + 4096) {
    for (int j = 0; j < PCC_NSTAMP; ++j) pcc_stamp_buf[(blockIdx.x - 16384) * PCC_NSTAMP + j] = st_sum[j];
  }
#endif

  // ---- epilogue: the window's rows are contiguous in `out`: coalesced 16-B stores (out == nullptr with the colour
  // head below: the last stage of g_s, whose rows nothing else reads)
  if (out != nullptr)
#pragma unroll
  for (int it = 0; it < R / 8; ++it) {
    const int r = it * 8 + grow;
    float4 v = *reinterpret_cast<const float4*>(&acc_lds[acc_at(chunk >> 2, r, chunk & 3)]);
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (row0 + r < n_out) *reinterpret_cast<float4*>(out + (row0 + r) * COUT + col0 + chunk * 4) = v;
  }
  if constexpr (HEAD) {
    float hv = head_b[0];
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const float4 v4 = *reinterpret_cast<const float4*>(&acc_lds[acc_at(c4 >> 2, lane < R ? lane : R, c4 & 3)]);
      const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = vv[j];
        if (relu) v = fmaxf(v, 0.f);
        hv = fmaf(v, head_w[4 * c4 + j], hv);
      }
    }
    if (row_ok) head_out[r_own] = hv;
    // the 1x1 colour head of g_s (32 -> 3, no activation) on the same rows: rgb[j] = b[j]; c ascending: fmaf(x[c],
    // w[c][j], rgb[j]) — the chain of pcc_linear; the rows themselves then never leave the LDS
    if (rgb_out != nullptr) {
      float cv[3] = {rgb_b[0], rgb_b[1], rgb_b[2]};
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 v4 = *reinterpret_cast<const float4*>(&acc_lds[acc_at(c4 >> 2, lane < R ? lane : R, c4 & 3)]);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = vv[j];
          if (relu) v = fmaxf(v, 0.f);
#pragma unroll
          for (int o = 0; o < 3; ++o) cv[o] = fmaf(v, rgb_w[(4 * c4 + j) * 3 + o], cv[o]);
        }
      }
      if (row_ok) {
        rgb_out[3 * r_own] = cv[0];
        rgb_out[3 * r_own + 1] = cv[1];
        rgb_out[3 * r_own + 2] = cv[2];
      }
    }
  }
}
