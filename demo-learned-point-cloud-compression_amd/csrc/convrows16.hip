// convrows16.hip — k_gconv_rows16 (convrows16.h) in a translation unit of its own: it is compiled with
// -mllvm -amdgpu-mfma-vgpr-form (Makefile).  The kernel keeps its accumulator tiles in registers across all offsets and
// selects per row between the tile before and after an offset's chains; with the accumulators in the matrix registers
// (the compiler's choice for long-lived MFMA operands) every step moved them out and back: 16 of its ~45 vector
// instructions, on the ALU the f32 MFMAs share.  The option is per translation unit, and conv.hip's kernels do not want
// it (k_gconv16 grows by a dozen vector instructions under it).
#include "common.h"
#include "convrows16.h"

void pcc_rows16_launch(hipStream_t st, int cout, int k_vol, unsigned n_windows, const float* d_in, const int32_t* d_nbr,
                       int64_t pitch, int64_t n_out, const float* wsw, const float* d_bias, int relu, float* d_out,
                       uint32_t in_bytes) {
  const dim3 grid(n_windows), block(64);
#define ROWS16(CO, KV) \
  hipLaunchKernelGGL((k_gconv_rows16<CO, KV>), grid, block, 0, st, d_in, d_nbr, pitch, n_out, wsw, d_bias, relu, d_out, in_bytes)
  if (cout == 32) {
    if (k_vol == 27) ROWS16(32, 27); else ROWS16(32, 8);
  } else {
    if (k_vol == 27) ROWS16(64, 27); else ROWS16(64, 8);
  }
#undef ROWS16
}
