// convup.h — the conv3 + occupancy head (+ colour head) layer of a g_s stage (receiver/decoder/codec_parallel.py:469),
// 32 -> 32 channels, on the 8 N generative children of a level, given only THAT level's 27-offset rule book
// (included by conv.hip after conv16.h, whose operand order of the weights and whose helpers it shares).
//
// On a generative level every parent has all of its 8 children, so the 27 neighbours of child row 8p + o fall in two
// classes that the arithmetic contract (include/pcc.h, "siblings first") visits one after the other:
//
//   1. the 8 SIBLINGS (offsets that stay inside parent p, the row itself included) always exist: 8 of the ~14 pairs of
//      a row.  For 16 parents they are one dense product Y[16 x 256] = X[16 x 256] . Wbig[256 x 256], Wbig's (o', o)
//      block being W[k(o' - o)]: no rule book, no ballot, no compaction, no pad slot, no accumulator traffic through
//      LDS.  A wave holds X (16 parents x 8 children x 32 channels: 64 registers) and the 8 x 2 accumulator tiles
//      (64 registers) and walks the 27 offsets once: offset k feeds the (8, 4, 2 or 1) octant pairs with o' - o = d(k),
//      so each octant's chain sees its siblings o' ascending = k ascending, and the weights of an offset are loaded once
//      per window (108 16-byte loads instead of 256).
//   2. the neighbours under OTHER parents exist where the parent-level neighbour exists: the irregular remainder,
//      handled like k_gconv16 (rows that have the offset packed into 16-slot items by ballot + mbcnt, accumulators
//      in LDS, everything requested one or two steps ahead) — but on a window of 128 rows = 16 parents with two rows per
//      lane: one compaction step serves twice the rows (vector / scalar bookkeeping per row about halved; f32 MFMAs
//      share the vector ALUs, DESIGN.md §4, so that is matrix time), and the lists of an offset are twice as long
//      (issued / useful slots of the remainder 1.45 on 64-row windows, 1.22 on 128-row ones; the whole layer 1.10
//      against k_gconv16's 1.18 on bench.py's dominant launch, simulated on its geometry).
//
// One wave per workgroup, no workgroup barrier.  LDS: 129 accumulator rows x 128 B + the slot records + book slice + step table = 19.5 KB:
// eight waves per CU (two per SIMD), so registers are plentiful (__launch_bounds__(64, 2)).
//
// Offsets of the remainder: j = 0 .. 25 <-> k = j + (j >= 13) (the centre offset has no other-parent pair).  An octant
// whose neighbour at offset k is a sibling sits that step out.  Up to 8 items per offset can occur (7 of 8 octants x 16
// parents); four ride the software pipeline (records, gathered rows and accumulator tiles requested ahead), the rest —
// windows in densely occupied regions — are contracted at the end of the step without prefetch.
#pragma once

// Offset step j (k = j + (j >= 13)) and octant o of an output row -> byte (kp | o' << 5): the offset of the parent-level
// neighbour that holds the row's neighbour (kp, 0 .. 26) and the neighbour's octant o' inside it; kp = 13 where the
// offset stays inside the row's own parent (siblings: the dense product) and for the steps past the last (row 26).
struct PccUpLut {
  unsigned char b[27 * 8];
};
__host__ __device__ constexpr PccUpLut pcc_up_lut() {
  PccUpLut t{};
  for (int j = 0; j < 27; ++j)
    for (int o = 0; o < 8; ++o) {
      int kp = 13, op = 0;
      if (j < 26) {
        const int k = j + (j >= 13 ? 1 : 0);
        const int d[3] = {k / 9 - 1, (k / 3) % 3 - 1, k % 3 - 1}, ob[3] = {(o >> 2) & 1, (o >> 1) & 1, o & 1}, w[3] = {9, 3, 1};
        kp = 0;
        for (int a = 0; a < 3; ++a) {
          const int tt = ob[a] + d[a];   // -1 .. 2
          kp += ((tt + 2) >> 1) * w[a];
          op |= (tt & 1) << (2 - a);
        }
        if (kp == 13) op = 0;
      }
      t.b[j * 8 + o] = (unsigned char)(kp | (op << 5));
    }
  return t;
}
__device__ const PccUpLut kPccUpLut = pcc_up_lut();

template <bool PERM>
__global__ __launch_bounds__(64, 2) void k_gconv_up(
    const float* __restrict__ in, const int32_t* __restrict__ nbrp, int64_t pitch, int64_t n_par,
    const float* __restrict__ wsw, const float* __restrict__ bias, int relu, float* __restrict__ out,
    const float* __restrict__ head_w, const float* __restrict__ head_b, float* __restrict__ head_out,
    const float* __restrict__ rgb_w, const float* __restrict__ rgb_b, float* __restrict__ rgb_out, uint32_t in_bytes) {
  constexpr int R = 128;             // rows of a window: 16 parents
  constexpr int HP = (R + 1) * 16;   // floats per accumulator plane (row R = sink of the pad slots)
  constexpr int NI = 4;              // pipelined items of an offset
  constexpr int NJ = 26;             // offsets of the remainder
  __shared__ __attribute__((aligned(16))) float acc_lds[2 * HP];
  // slot -> (byte offset of the input row, accumulator row address): every lane writes the records of its two rows, a
  // present row the record of its rank, an absent one a pad record behind the list — all R slots every step, no branch.
  // One buffer: the four pipelined items of a step hold their records in registers, the items beyond them read theirs at
  // the top of the step, before the compaction of the next offset overwrites the list
  __shared__ __attribute__((aligned(8))) int2 rec[R];
  // the window's slice of the parent rule book, [27][16] (-1: no such parent or neighbour).  Fetched once, in front of the
  // sibling product: the steps of the remainder then issue no load whose latency they cannot plan for — as two dword
  // loads per step straight from the book (44 MB, streamed once: HBM latency) these sat in the in-order load queue in
  // front of the gathers, and every step lasted one such latency (2.4k cycles for 0.5k - 2k cycles of matrix work)
  __shared__ int32_t pb[27 * 16];
  // kPccUpLut (216 bytes): row 13 of pb is filled with -1 — an offset that stays inside the parent, or lies past the last
  // step, reads "no neighbour" there, with no compare in the step
  __shared__ __attribute__((aligned(4))) unsigned char lut[27 * 8];

  const int lane = threadIdx.x;
  // Windows are taken from the END of the tensor, in dispatch order: the up stage in front of this layer (k_convT16p,
  // persistent waves over ascending tiles) has just written the rows in ascending order, 417 MB at the large stage, so
  // the rows it wrote last are the ones still in the 256-MB Infinity Cache — read in ascending order instead, every
  // window evicts rows that are still to come.  1.059 / 1.060 ms against 1.068 / 1.070 in the timed region of bench.py
  // on one box (two interleaved passes of 30 steps; round 3's order — one contiguous eighth of the windows per XCD,
  // ascending — against this one); neighbouring windows now run on different XCDs, which measures as nothing (round 3:
  // runs of 1 / 16 / 128 windows per XCD).
  const int64_t window = (int64_t)gridDim.x - 1 - (int64_t)blockIdx.x;
  const int64_t par0 = window * 16;
  if (par0 >= n_par) return;
  const int64_t row0 = par0 * 8, n_out = n_par * 8;
  const int n = lane & 15, q = lane >> 4;
  const int grow = lane >> 3, chunk = lane & 7;

  auto acc_at = [&](int half, int row, int qq) -> int { return half * HP + row * 16 + (((qq + 2 * (row >> 2)) & 3) << 2); };
  auto acc_row = [](int row) -> int { return row * 64 + (((row >> 1) & 2) << 4); };   // 4 * acc_at(0, row, qq) == acc_row(row) ^ (qq << 4)
  const int a_own0 = acc_row(lane), a_own1 = acc_row(lane + 64), a_sink = acc_row(R), q16 = q << 4;
  const uint32_t qoff = (uint32_t)q * 32u;
  constexpr uint32_t kPadOff = 0xFFFFFF80u;   // beyond the buffer: the load returns zeros without a fetch
  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00027000);

  auto load_w = [&](float4 (&W)[4], int k) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(wsw + (int64_t)k * 1024) + (uint32_t)lane * 64u);
#pragma unroll
    for (int j = 0; j < 4; ++j) W[j] = p[j];
  };
  auto load_row = [&](uint32_t off, float4& g0, float4& g1) {
    const auto r0 = __builtin_amdgcn_raw_buffer_load_b128(in_rs, off, 0, 0);
    const auto r1 = __builtin_amdgcn_raw_buffer_load_b128(in_rs, off + 16u, 0, 0);
    g0 = make_float4(__uint_as_float(r0[0]), __uint_as_float(r0[1]), __uint_as_float(r0[2]), __uint_as_float(r0[3]));
    g1 = make_float4(__uint_as_float(r1[0]), __uint_as_float(r1[1]), __uint_as_float(r1[2]), __uint_as_float(r1[3]));
  };
  // B operands in MFMA order from the two 16-B pieces lane (n, q) holds of its slot's row (conv16.h, PERM)
  auto shape = [&](const float4& g0, const float4& g1, float (&xv)[8]) {
    if constexpr (PERM) {
      xv[0] = g0.x; xv[1] = g0.y; xv[2] = g0.z; xv[3] = g0.w;
      xv[4] = g1.x; xv[5] = g1.y; xv[6] = g1.z; xv[7] = g1.w;
    } else {
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
  // the records of the pipelined items are resolved for the reading lane once per step, in one group (input offset | qoff,
  // accumulator address ^ q16): tile_read / tile_write take the resolved address
  auto tile_read = [&](int addr, f32x4& lo, f32x4& hi) { acc_read(addr ^ q16, lo, hi); };
  auto tile_write = [&](int addr, const f32x4& lo, const f32x4& hi) { acc_write(addr ^ q16, lo, hi); };

  // ---- remainder, first requests (their latency hides behind the sibling product): the two rows of lane l are
  // children `oct` of parents par0 + (l >> 3) and par0 + 8 + (l >> 3)
  const int oct = lane & 7;
  int32_t pbv[7];   // element e = lane + 64 i of the slice: offset e >> 4, parent e & 15
  // (unconditional loads off a descriptor with 32-bit offsets — the book is below 27 x 2^24 entries —: an entry that does
  // not exist is requested beyond the book and replaced by -1 where the slice is written to LDS)
  const __amdgpu_buffer_rsrc_t book_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(nbrp), 0, (int)(uint32_t)(27 * pitch * 4), 0x00027000);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int e = lane + 64 * i;
    const int64_t pp = par0 + (e & 15);
    const bool have = e < 27 * 16 && (e >> 4) != 13 && pp < n_par;
    pbv[i] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(book_rs, have ? (uint32_t)(((int64_t)(e >> 4) * pitch + pp) * 4) : 0xFFFFFFF0u, 0, 0);
  }
  int32_t nb0 = -1, nb1 = -1;   // rows of the parent-level neighbours of the lane's two parents (-1: none, or a sibling offset)
  uint32_t nb_op7 = 0u;         // octant of the neighbour inside that parent, << 7
  uint32_t cb = 13u;            // kPccUpLut byte of the step after the one nb0 / nb1 belong to
  // two LDS reads in a row, one step apart: the byte of step j, then (next call) the book entries it names
  auto request_lut = [&](int j) { cb = lut[(j < 26 ? j : 26) * 8 + oct]; };
  auto request_nb = [&]() {
    nb_op7 = (cb & 0xE0u) << 2;
    const int32_t* pr = pb + ((cb & 31u) << 4) + (lane >> 3);
    nb0 = pr[0];
    nb1 = pr[8];
  };
  // pack the rows that have the requested offset (under another parent) into the slot records; returns their count.
  // (Pad records written to every lane's own two slots first, then the present rows' records over them under EXEC: 10
  // vector instructions instead of 22, two more LDS writes and two EXEC regions — 0.6 % slower.)
  auto compact = [&]() -> int {
    const bool p0 = nb0 >= 0, p1 = nb1 >= 0;
    const unsigned long long bal0 = __builtin_amdgcn_ballot_w64(p0), bal1 = __builtin_amdgcn_ballot_w64(p1);
    const int c0 = __popcll(bal0), cnt = c0 + __popcll(bal1);
    const int r0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal0, 0u));
    const int r1 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal1, 0u));
    // present rows: ranks 0 .. cnt-1 (first rows, then second rows); absent ones: the pads cnt .. R-1 in the same order
    const int s0 = p0 ? r0 : cnt + lane - r0;
    const int s1 = p1 ? c0 + r1 : cnt + 64 - c0 + lane - r1;
    rec[s0] = p0 ? make_int2((int32_t)(((uint32_t)nb0 << 10) | nb_op7), a_own0) : make_int2((int32_t)kPadOff, a_sink);
    rec[s1] = p1 ? make_int2((int32_t)(((uint32_t)nb1 << 10) | nb_op7), a_own1) : make_int2((int32_t)kPadOff, a_sink);
    return cnt;
  };
#if PCC_CONV_STAMP
  unsigned long long st_sum[PCC_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
  unsigned long long rt0, mt0 = st_last;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
#endif

  // ---- 1. siblings: dense product over the window's 16 parents (slot n = parent par0 + n)
  {
    float xs[8][8];
    {
      const int64_t pn = par0 + n;
      const uint32_t rbase = pn < n_par ? (((uint32_t)pn << 10) | qoff) : kPadOff;   // child 0 of the slot's parent
#pragma unroll
      for (int op = 0; op < 8; ++op) {
        float4 g0, g1;
        load_row(pn < n_par ? rbase + ((uint32_t)op << 7) : kPadOff, g0, g1);
        shape(g0, g1, xs[op]);
      }
    }
    f32x4 lo[8], hi[8];
    {
      const float* bp = bias + 4 * q;
      const f32x4 bl = {bp[0], bp[1], bp[2], bp[3]}, bh = {bp[16], bp[17], bp[18], bp[19]};
#pragma unroll
      for (int o = 0; o < 8; ++o) { lo[o] = bl; hi[o] = bh; }
    }
    PCC_STAMP(0);   // issue of the X loads
    // weights of an offset: one 4-KB block per wave from L2 (the 108 KB of a layer's weights do not fit the 32-KB L1).
    // An offset's chains last 16 .. 128 MFMAs (0.25 .. 2 us); requested WD offsets ahead, and pinned there by the
    // scheduling barriers, the blocks arrive behind matrix work instead of in front of it
    constexpr int WD = 3;
    float4 Wd[WD + 1][4];
#pragma unroll
    for (int k = 0; k < WD; ++k) load_w(Wd[k], k);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      if (k + WD < 27) load_w(Wd[(k + WD) % (WD + 1)], k + WD);
      __builtin_amdgcn_sched_barrier(0);
      const float4 (&Wc)[4] = Wd[k % (WD + 1)];
      const float wl[8] = {Wc[0].x, Wc[0].y, Wc[0].z, Wc[0].w, Wc[1].x, Wc[1].y, Wc[1].z, Wc[1].w};
      const float wh[8] = {Wc[2].x, Wc[2].y, Wc[2].z, Wc[2].w, Wc[3].x, Wc[3].y, Wc[3].z, Wc[3].w};
      const int dx = k / 9 - 1, dy = (k / 3) % 3 - 1, dz = k % 3 - 1;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int tx = ((o >> 2) & 1) + dx, ty = ((o >> 1) & 1) + dy, tz = (o & 1) + dz;
        if (tx < 0 || tx > 1 || ty < 0 || ty > 1 || tz < 0 || tz > 1) continue;   // the offset leaves the parent for this octant
        const int op = tx * 4 + ty * 2 + tz;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          lo[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xs[op][s], lo[o], 0, 0, 0);
          hi[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xs[op][s], hi[o], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    PCC_STAMP(2);   // sibling product
    // the tiles seed the accumulators of the remainder: lane (n, q) holds channels 4q .. (+16) of row 8 n + o
#pragma unroll
    for (int o = 0; o < 8; ++o) acc_write(acc_row(8 * n + o), lo[o], hi[o]);
    if (lane < 8) *reinterpret_cast<float4*>(&acc_lds[acc_at(lane >> 2, R, lane & 3)]) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int e = lane + 64 * i;
      const bool have = (e >> 4) != 13 && par0 + (e & 15) < n_par;
      if (e < 27 * 16) pb[e] = have ? pbv[i] : -1;
    }
    if (lane < 54) reinterpret_cast<uint32_t*>(lut)[lane] = reinterpret_cast<const uint32_t*>(kPccUpLut.b)[lane];
  }

  // ---- 2. remainder
  float4 G[NI][2];
  float4 W0[4], W1[4];
  int rin[NI], ra0[NI], ra1[NI];
  auto load_wj = [&](float4 (&W)[4], int j) {
    const int jj = j < NJ ? j : NJ - 1;
    load_w(W, jj + (jj >= 13 ? 1 : 0));
  };
  auto read_records = [&](int (&racc)[NI]) {
#pragma unroll
    for (int g = 0; g < NI; ++g) {
      const int2 r = rec[g * 16 + n];
      rin[g] = r.x;
      racc[g] = r.y;
    }
  };
  auto gather = [&](int g) { load_row((uint32_t)rin[g], G[g][0], G[g][1]); };
  auto resolve = [&](int (&racc)[NI]) {
#pragma unroll
    for (int g = 0; g < NI; ++g) {
      rin[g] = (int)((uint32_t)rin[g] | qoff);
      racc[g] ^= q16;
      // materialised here: sunk to its use behind a conditional item, the value would be waited for with lgkmcnt(0) at the
      // join — behind every LDS operation of the items in between
      asm volatile("" : "+v"(rin[g]), "+v"(racc[g]));
    }
    asm volatile("" : "+v"(nb0), "+v"(nb1), "+v"(cb));   // requested with the records, arrived with them
  };
#define UP_LIVE(g) (cnt_cur > 16 * (g))
  int cnt_cur;
  auto step = [&](int j, float4 (&Wc)[4], float4 (&Wn)[4], int (&rc_)[NI], int (&rn)[NI]) {
#if PCC_CONV_STAMP
    unsigned long long tq[8];
#define UP_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(tq[i])::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define UP_STAMP(i) do { } while (0)
#endif
    UP_STAMP(0);
    f32x4 lo0, hi0, lo1, hi1;
    const float wl[8] = {Wc[0].x, Wc[0].y, Wc[0].z, Wc[0].w, Wc[1].x, Wc[1].y, Wc[1].z, Wc[1].w};
    const float wh[8] = {Wc[2].x, Wc[2].y, Wc[2].z, Wc[2].w, Wc[3].x, Wc[3].y, Wc[3].z, Wc[3].w};
    // items beyond the pipeline (more than 64 rows of the window have the offset) first, while the list is still in
    // LDS (items of one offset touch disjoint rows: their order is free).
    if (cnt_cur > 16 * NI) {
      // the next item's record and rows are requested before this item's chains (a pad record behind the last item: its
      // load returns zeros without a fetch) — one item at a time, each lasted a record read + a row fetch + a tile read
      int2 r = rec[NI * 16 + n];
      float4 g0, g1;
      load_row((uint32_t)r.x | qoff, g0, g1);
      for (int g = NI; 16 * g < cnt_cur; ++g) {
        const int2 rn2 = rec[(g + 1 < R / 16 ? g + 1 : R / 16 - 1) * 16 + n];
        float4 h0, h1;
        load_row((uint32_t)rn2.x | qoff, h0, h1);
        f32x4 lo, hi;
        acc_read(r.y, lo, hi);
        float xv[8];
        shape(g0, g1, xv);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          lo = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xv[s], lo, 0, 0, 0);
          hi = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xv[s], hi, 0, 0, 0);
        }
        acc_write(r.y, lo, hi);
        r = rn2;
        g0 = h0;
        g1 = h1;
      }
    }
    UP_STAMP(6);
    // item 0 is unconditional (pad slots into the sink row when the offset has no row).  The step's bookkeeping sits in
    // front of its MFMAs as ONE group of vector instructions, behind the request for the tiles (their LDS round trip runs
    // under it): a vector instruction between two MFMAs costs a lone wave 16 cycles, in a group 4 (tools/micro/issue.hip),
    // and f32 MFMAs share the vector ALU — whatever the partner wave does, this time is not hidden.  The records arrive
    // under the first four MFMA pairs; they are resolved and the first gather of the next offset issued in the middle.
    tile_read(rc_[0], lo0, hi0);
    tile_read(rc_[1], lo1, hi1);
    __builtin_amdgcn_sched_barrier(0);
    const int cnt_next = compact();   // offset j + 1
    request_nb();                     // book entries of offset j + 2
    request_lut(j + 3);
    load_wj(Wn, j + 1);
    PCC16_SYNC();
    read_records(rn);
    __builtin_amdgcn_sched_barrier(0);
    float xv0[8];
    shape(G[0][0], G[0][1], xv0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      lo0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xv0[s], lo0, 0, 0, 0);
      hi0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xv0[s], hi0, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    resolve(rn);
    gather(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 4; s < 8; ++s) {
      lo0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xv0[s], lo0, 0, 0, 0);
      hi0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xv0[s], hi0, 0, 0, 0);
    }
    if (!UP_LIVE(1)) tile_write(rc_[0], lo0, hi0);
    UP_STAMP(5);   // item 0 with the bookkeeping
    // Item g >= 1: [write-back of item g-1 behind the first MFMA pair, tile of item g+1 requested] chains of item g; the
    // last item of the step writes itself back.  The gathers of the next offset's four items are issued whether or not
    // the item exists (conv16.h: exact vmcnt counts).  (Four straight-line tails selected by the number of items instead
    // of a branch around every item were built: 5 % slower.)
#define UP_ITEM(g, LO, HI, PLO, PHI)                                                                \
    if (UP_LIVE(g)) {                                                                               \
      const bool more = (g) + 1 < NI && UP_LIVE((g) + 1);                                              \
      float xv[8];                                                                                     \
      shape(G[g][0], G[g][1], xv);                                                                     \
      LO = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[0], xv[0], LO, 0, 0, 0);                            \
      HI = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[0], xv[0], HI, 0, 0, 0);                            \
      tile_write(rc_[(g) - 1], PLO, PHI);                                                              \
      if constexpr ((g) + 1 < NI) { if (more) tile_read(rc_[(g) + 1 < NI ? (g) + 1 : 0], PLO, PHI); } \
      _Pragma("unroll") for (int s = 1; s < 8; ++s) {                                                  \
        LO = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[s], xv[s], LO, 0, 0, 0);                          \
        HI = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], xv[s], HI, 0, 0, 0);                          \
      }                                                                                                \
      if (!more) tile_write(rc_[g], LO, HI);                                                           \
    }                                                                                                  \
    gather(g)
    UP_ITEM(1, lo1, hi1, lo0, hi0);
    UP_ITEM(2, lo0, hi0, lo1, hi1);
    UP_ITEM(3, lo1, hi1, lo0, hi0);
#undef UP_ITEM
    UP_STAMP(1);
#if PCC_CONV_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    st_sum[3] += tq[0] - st_last;   // loop back
    st_sum[8] += tq[6] - tq[0];     // overflow items
    st_sum[4] += tq[5] - tq[6];     // item 0 + bookkeeping
    st_sum[5] += tq[1] - tq[5];     // items 1 .. 3, gathers
    st_sum[9] += (unsigned long long)((cnt_cur + 15) >> 4);   // items of the step
    st_last = tq[1];
#endif
    cnt_cur = cnt_next;
  };

  // prologue of the remainder: offset 0 compacted and gathered, offset 1 requested
  PCC16_SYNC();
  request_lut(0);
  request_nb();      // offset 0
  request_lut(1);
  cnt_cur = compact();
  request_nb();      // offset 1
  request_lut(2);
  load_wj(W0, 0);
  PCC16_SYNC();
  read_records(ra0);
  resolve(ra0);
#pragma unroll
  for (int g = 0; g < NI; ++g) gather(g);
  for (int j = 0; j < NJ; j += 2) {
    step(j, W0, W1, ra0, ra1);
    step(j + 1, W1, W0, ra1, ra0);
  }
  PCC16_SYNC();

  // ---- epilogue: the window's rows are contiguous in `out` (nullptr with the colour head: the last stage of g_s).  Vector
  // instructions are what the epilogue costs (they share the ALU with the partner wave's MFMAs): one v_max per element
  // (no canonicalising copy in front of it), LDS and store addresses as immediates off one register, rows past the end
  // dropped by the store's buffer bounds instead of a compare per row
  auto relu1 = [&](float v) -> float {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));   // +0 for -0, 0 for a quiet NaN: v > 0 ? v : 0
    return r;
  };
  if (out != nullptr) {
    const int64_t rows_left = n_out - row0;
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
        out + row0 * 32, 0, (int)((rows_left < R ? rows_left : R) * 128), 0x00027000);
    // row it * 8 + grow, piece chunk: the swizzle of acc_at does not depend on `it` (8 rows = 2 swizzle periods)
    const float* lp = &acc_lds[acc_at(chunk >> 2, grow, chunk & 3)];
    const uint32_t voff = (uint32_t)grow * 128u + (uint32_t)chunk * 16u;
    auto store_rows = [&](auto with_relu) {   // one straight-line body per case: the reads of all rows in flight together
      float4 v[R / 8];
#pragma unroll
      for (int it = 0; it < R / 8; ++it) v[it] = *reinterpret_cast<const float4*>(lp + it * 8 * 16);
#pragma unroll
      for (int it = 0; it < R / 8; ++it) {
        float4 t = v[it];
        if constexpr (decltype(with_relu)::value) { t.x = relu1(t.x); t.y = relu1(t.y); t.z = relu1(t.z); t.w = relu1(t.w); }
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t d = {__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w)};
        __builtin_amdgcn_raw_buffer_store_b128(d, out_rs, voff, it * 1024, 0);
      }
    };
    if (relu) store_rows(std::true_type{}); else store_rows(std::false_type{});
  }
  // occupancy head (32 -> 1) and, at the last stage, the colour head (32 -> 3, no activation) of the lane's two rows:
  // c ascending fmaf chains, the bits of pcc_linear on the stored rows
  auto heads = [&](auto with_relu) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = lane + 64 * h;
    const int64_t gr = row0 + r;
    float hv = head_b[0];
    float cv[3] = {0.f, 0.f, 0.f};
    const bool with_rgb = rgb_out != nullptr;
    if (with_rgb) { cv[0] = rgb_b[0]; cv[1] = rgb_b[1]; cv[2] = rgb_b[2]; }
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const float4 v4 = *reinterpret_cast<const float4*>(&acc_lds[acc_at(c4 >> 2, r, c4 & 3)]);
      const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = vv[j];
        if constexpr (decltype(with_relu)::value) v = relu1(v);
        hv = fmaf(v, head_w[4 * c4 + j], hv);
        if (with_rgb) {
#pragma unroll
          for (int o = 0; o < 3; ++o) cv[o] = fmaf(v, rgb_w[(4 * c4 + j) * 3 + o], cv[o]);
        }
      }
    }
    if (gr < n_out) {
      head_out[gr] = hv;
      if (with_rgb) {
        rgb_out[3 * gr] = cv[0];
        rgb_out[3 * gr + 1] = cv[1];
        rgb_out[3 * gr + 2] = cv[2];
      }
    }
  }
  };
  if (relu) heads(std::true_type{}); else heads(std::false_type{});
#if PCC_CONV_STAMP
  {   // shader clock over the wave's life: s_memtime ticks per 100-MHz s_memrealtime tick, x 1000 (st_sum[1] = kHz / 100)
    unsigned long long rt1, mt1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt1), "=s"(rt1)::"memory");
    st_sum[1] = (mt1 - mt0) * 1000ull / (rt1 - rt0 ? rt1 - rt0 : 1ull);
    st_sum[6] = rt0;   // life of the wave on the 100-MHz clock: launch makespan and resident waves from the sample
    st_sum[7] = rt1;
  }
  if (lane == 0 && blockIdx.x % 6 == 0 && blockIdx.x / 6 < 4096) {   // every sixth window: a uniform sample of a 25k-window launch
    for (int j = 0; j < PCC_NSTAMP; ++j) pcc_stamp_buf[(blockIdx.x / 6) * PCC_NSTAMP + j] = st_sum[j];
  }
#endif
}
