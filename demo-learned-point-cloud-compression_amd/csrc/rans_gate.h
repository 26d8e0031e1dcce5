// rans_gate.h — internal (not part of the C-ABI): chunk gates of the host rANS coders.
#pragma once
#include <stdint.h>
#include <vector>

// Host rANS coders working through a buffer that is still crossing PCIe (rans_host.cpp; used by codec.hip).
// The flattened symbol array is cut into chunks; `fn(user, c)` is called
//   encoder: BEFORE the first access to chunk c — chunks are visited from the end of the array to its start:
//            chunk c = [bound[c], c == 0 ? n : bound[c-1]), bound descending, bound[n_chunks-1] == 0;
//   decoder: AFTER the last symbol of chunk c has been written — chunks in array order:
//            chunk c = [c == 0 ? 0 : bound[c-1], bound[c]), bound ascending, bound[n_chunks-1] == n.
struct PccRansGate {
  int n_chunks;
  const int64_t* bound;
  void (*fn)(void* user, int chunk);
  void* user;
};
// Coder tables of a CDF set (per-symbol reciprocals for the encoder, bucket LUTs for the decoder), built once: for the
// 64 Gaussian tables (27k entries) building them costs ~0.2 ms, which a codec should not pay on every frame.
// `tables` may be NULL in the calls below (then they are built for the call).
struct PccRansTables;
PccRansTables* pcc_rans_tables_build(const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                                     const int32_t* h_offsets, int n_cdf);
void pcc_rans_tables_free(PccRansTables* t);
int pcc_rans_encode16_gated(const int16_t* h_sym, const uint8_t* h_idx, int64_t n, const int32_t* h_cdfs,
                            int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                            uint8_t* h_out, int64_t cap, int64_t* h_len, const PccRansGate* gate,
                            const PccRansTables* tables);
// Seek points of a host-coded stream (codec.hip's "PCSK" trailer): index[k] ascending symbol positions, filled by the
// encoder with the coder's state and the number of 32-bit words a decoder has consumed when symbol index[k] is next.
// A decoder that holds them decodes the pieces between the points on as many threads, each piece checked against the
// point behind it (pcc_rans_decode8_range returns where it ended).
struct PccRansSeek {
  int n;
  const int64_t* index;
  uint64_t* state;
  int64_t* word;
};
int pcc_rans_encode16_seek(const int16_t* h_sym, const uint8_t* h_idx, int64_t n, const int32_t* h_cdfs,
                           int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                           uint8_t* h_out, int64_t cap, int64_t* h_len, const PccRansGate* gate,
                           const PccRansTables* tables, PccRansSeek* seek);
int pcc_rans_decode8_range(const uint8_t* h_in, int64_t len, const uint8_t* h_idx, int64_t n, const int32_t* h_cdfs,
                           int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf, int32_t* h_sym,
                           const PccRansTables* tables, int64_t i_lo, int64_t i_hi, uint64_t state_in, int64_t word_in,
                           uint64_t* state_out, int64_t* word_out);
int pcc_rans_decode8_gated(const uint8_t* h_in, int64_t len, const uint8_t* h_idx, int64_t n, const int32_t* h_cdfs,
                           int cdf_pitch, const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                           int32_t* h_sym, const PccRansGate* gate, const PccRansTables* tables);


// octree_host.cpp: pcc_octree_unpack_levels into a vector that is sized by what the stream actually DECODED, not by
// the point count its header announces (decoders of untrusted containers; h_level_n needs 16 entries).
int pcc_octree_unpack_vec(const uint8_t* h_in, int64_t len, std::vector<int32_t>* pts, int64_t* h_level_n);
// octree_host.cpp, blob version 3 (parts coded and decoded side by side): the parts of a blob (K = 1 and the blob itself
// for versions 1 / 2; arrays of 16), one part decoded, the decoded parts checked against each other and put together
// (cells = Morton cell indexes relative to the root cube), cells -> points, and the envelope around finished parts
struct PccOctPart {
  std::vector<uint64_t> cells;
  int64_t level_n[16];
};
int pcc_octree_parts(const uint8_t* h_in, int64_t len, int* K, const uint8_t** part, int64_t* part_len);
int pcc_octree_unpack_part(const uint8_t* h_in, int64_t len, PccOctPart* out);
int pcc_octree_merge_parts(const uint8_t* h_in, int64_t len, PccOctPart* parts, int K, std::vector<uint64_t>* cells,
                           int64_t* h_level_n);
void pcc_octree_cells_to_points(const uint64_t* cells, int64_t n, const int32_t origin[3], int32_t* out);
int pcc_octree_join_parts(int depth, const int32_t origin[3], int64_t n_points, const std::vector<uint8_t>* parts, int K,
                          uint8_t* h_out, int64_t cap, int64_t* h_len);

// rans_gpu.hip: pcc_rans_encode_dev without its read-back (see there)
struct pcc_ctx;
struct pcc_rans_dev;
int pcc_rans_encode_dev_async(pcc_ctx* ctx, const pcc_rans_dev* tables, const int32_t* d_sym, const uint8_t* d_idx,
                              int64_t idx_run, int64_t n, int n_streams, uint8_t* d_out, int64_t cap_each,
                              long long* d_lens, int attempt);

// octree2.hip: blob version 2 of the geometry slot (levels, entropy coder and decoder on the GPU)
int pcc_octree2_encode(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int key_shift, int depth, const int32_t origin[3],
                       uint8_t* h_out, int64_t cap, int64_t* h_len);
int pcc_octree2_decode(pcc_ctx* ctx, const uint8_t* h_in, int64_t len, int32_t* d_points, int32_t* h_points, int64_t cap_points,
                       int64_t* h_n_points, int64_t* h_level_n);
