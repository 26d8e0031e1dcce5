// What does a single wave pay for the non-matrix instructions of k_gconv_up's item between its v_mfma_f32_16x16x4_f32?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/issue.hip -o tools/micro/issue && tools/micro/issue
// One wave on a CU.  Loop body = one "item": 16 MFMAs in two alternating chains (512 cycles of the matrix pipe), plus,
// by mode bit: 1 accumulator tile from LDS in front / back to LDS behind (the tile of the NEXT iteration requested
// behind the first MFMA pair, as the kernel does); 2 two 16-B buffer loads per iteration, consumed one iteration later
// as the B operands; 4 eight independent VALU adds spread between the MFMAs; 8 a compaction (v_cmp, s_bcnt1, v_mbcnt x 2,
// v_cndmask, ds_write_b64, wave barrier, ds_read_b64); 16 a taken branch around a dead block; 32 eight SALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int mode>
__global__ __launch_bounds__(64, 2) void k(int iters, const float* __restrict__ src, int src_bytes, unsigned long long* out,
                                        float* sink, int never) {
  __shared__ __attribute__((aligned(16))) float acc[2 * 129 * 16];
  __shared__ __attribute__((aligned(8))) int2 rec[128];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2 * 129 * 16; i += 64) acc[i] = 0.f;
  rec[lane] = make_int2(lane * 128, lane * 64);
  rec[lane + 64] = make_int2(lane * 128 + 8192, lane * 64 + 4096);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00027000);
  float w[16];
  for (int j = 0; j < 16; ++j) w[j] = 0.001f * (lane + j);
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.5f + j;
  u32x4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
  int va[8];
  for (int j = 0; j < 8; ++j) va[j] = lane + j;
  int sa = never, recx = lane * 128;
  unsigned arow = (unsigned)(lane & 15) * 64u + (unsigned)(lane >> 4) * 16u, anext = arow;
  f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0}, nlo = lo, nhi = hi;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if (mode & 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[j] = __uint_as_float(g0[j]); x[4 + j] = __uint_as_float(g1[j]); }
    }
    if (mode & 1) { lo = nlo; hi = nhi; }
    __builtin_amdgcn_sched_barrier(0);
    lo = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], x[0], lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8], x[0], hi, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (mode & 1) {   // tile of the next item (another row: i-dependent address)
      anext = (arow + 1024u * (unsigned)(i & 7)) & 8191u;
      const float* b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(acc) + anext);
      const float4 a4 = *reinterpret_cast<const float4*>(b), b4 = *reinterpret_cast<const float4*>(b + 129 * 16);
      nlo = f32x4{a4.x, a4.y, a4.z, a4.w};
      nhi = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
    if (mode & 64) {
#pragma unroll
      for (int s = 1; s < 8; ++s) va[s] += va[(s + 3) & 7];
    }
    if (mode & 128) {
#pragma unroll
      for (int s = 1; s < 8; ++s) va[s] += va[(s + 3) & 7];
#pragma unroll
      for (int s = 1; s < 8; ++s) va[s] ^= va[(s + 5) & 7];
    }
    if (mode & 256) {
#pragma unroll
      for (int s = 1; s < 8; ++s) va[0] = va[0] * 3 + s;
    }
    if (mode & 8) {
      const bool p = (va[0] & 1) == 0;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(p);
      const int cnt = __popcll(bal);
      const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
      rec[p ? r : cnt + lane - r] = make_int2(p ? va[1] : -128, p ? lane * 64 : 8192);
      va[0] += cnt;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 1; s < 8; ++s) {
      lo = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], x[s], lo, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8 + s], x[s], hi, 0, 0, 0);
      if (mode & 4) { __builtin_amdgcn_sched_barrier(0); va[s] += va[(s + 3) & 7]; __builtin_amdgcn_sched_barrier(0); }
      if ((mode & 32) && s < 5) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_add_u32 %0, %0, 3\n\ts_lshl_b32 %0, %0, 1" : "+s"(sa)); __builtin_amdgcn_sched_barrier(0); }
      if ((mode & 8) && s == 4) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        recx = rec[lane & 15].x;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (mode & 1) {   // write-back of this item
      float* b = reinterpret_cast<float*>(reinterpret_cast<char*>(acc) + arow);
      *reinterpret_cast<float4*>(b) = make_float4(lo[0], lo[1], lo[2], lo[3]);
      *reinterpret_cast<float4*>(b + 129 * 16) = make_float4(hi[0], hi[1], hi[2], hi[3]);
      arow = anext;
    }
    if (mode & 2) {
      const unsigned off = ((unsigned)(recx + i * 128) & 0x3F80u) | ((unsigned)(lane >> 4) * 32u);
      g0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      g1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u, 0, 0);
    }
    if (mode & 16) {
      __builtin_amdgcn_sched_barrier(0);
      if (sa == 12345 + i) { sink[lane] = lo[0]; }   // never: the branch around it is taken every iteration
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[0] = t1 - t0;
  float s = lo[0] + hi[1] + nlo[2] + nhi[3] + __uint_as_float(g0[0]) + __uint_as_float(g1[1]) + (float)recx + (float)sa;
  for (int j = 0; j < 8; ++j) s += (float)va[j] + x[j];
  sink[lane] = s;
}

int main() {
  unsigned long long* out;
  float *sink, *src;
  hipMalloc(&out, 16);
  hipMalloc(&sink, 4096);
  hipMalloc(&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  const int iters = 4000;
#define CASE(M, NAME) do { unsigned long long h = 0; for (int rep = 0; rep < 2; ++rep) { \
      hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, iters, src, 1 << 20, out, sink, 0); \
      hipDeviceSynchronize(); hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); } \
    printf("%-36s %8.1f cycles / iteration\n", NAME, (double)h / iters); } while (0)
  CASE(0, "16 MFMAs");
  CASE(1, "+ tile from / to LDS");
  CASE(2, "+ two 16-B buffer loads");
  CASE(4, "+ 7 VALU adds between");
  CASE(64, "+ 7 VALU adds in one group");
  CASE(128, "+ 14 VALU in one group");
  CASE(256, "+ 7 dependent VALU (mad) in one group");
  CASE(8, "+ compaction + record round trip");
  CASE(16, "+ a compare and branch");
  CASE(32, "+ 8 SALU");
  CASE(3, "+ tile + loads");
  CASE(1 + 2 + 16, "item: tile + loads + branch");
  CASE(1 + 2 + 4 + 8 + 16 + 32, "item 0: everything");
  return 0;
}
