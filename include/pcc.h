/*
 * pcc.h — C-ABI of libpcc_hip.so, the MI355X (gfx950) codec hot path.
 *
 * This is the drop-in boundary for the path
 *     CompressionPipeline.compress()    sender/encoder/codec_pipeline.py:196
 *     DecompressionPipeline.decompress() receiver/decoder/codec_parallel.py:141
 * of ikt-luh/Demo-Learned-Point-Cloud-Compression.  In the reference that path
 * crosses into native code through four third-party bindings that are NOT in
 * the reference tree (MinkowskiEngine pybind, CompressAI's rANS pybind, the
 * tmc3 subprocess, the `bitstream` Cython module).
 * Each entry point below names the reference call site it serves.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures;
 *   - `d_` prefix  = pointer into device (HBM) memory, `h_` = host memory;
 *   - every device op is enqueued on the ctx stream and is asynchronous unless
 *     it returns a count through a host pointer (then it synchronises the
 *     stream once before returning);
 *   - return value: 0 = ok, negative = error (pcc_last_error() has the text);
 *     the library never throws across the ABI;
 *   - caller owns inputs and outputs; scratch comes from a ctx-owned arena
 *     that is valid until the next call on the same ctx;
 *   - one ctx per in-flight compress()/decompress() call ("slot"): the
 *     reference runs up to 3 concurrent calls (sender/encoder/encoder.py:50).
 *
 * Data layout (HBM)
 *   coords  int32 [N,4]  rows (b,x,y,z), coordinates multiples of the tensor
 *                        stride, each in [-32768,32767], b in [0,65535]
 *   keys    uint64 [N]   Morton key: b<<48 | interleave(x+32768,y+32768,z+32768)
 *                        (x is the top bit of each triple).  Rows of every
 *                        sparse tensor are kept sorted by this key.
 *   feats   float32 [N,C] row-major
 *   nbr     int32 [K,N]  rule book, out-stationary: nbr[k*N+n] = input row
 *                        feeding output row n through kernel offset k, -1 if
 *                        absent.  3^3: k=(dx+1)*9+(dy+1)*3+(dz+1); 2^3: k=octant
 *                        = xbit<<2|ybit<<1|zbit.
 *   weights float32 [K,Cin,Cout], bias float32 [Cout]
 *
 * Arithmetic contract (what the parity tests check bit-for-bit)
 *   out[n][co] = bias[co]; for k ascending (present neighbours only), for ci
 *   ascending: out = fmaf(in[nbr[k][n]][ci], W[k][ci][co], out); then ReLU if
 *   asked.  Evaluated with v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32
 *   (exact k-ordered fmaf chains) or scalar fmaf; never atomics.
 *   The conv3 + occupancy-head layers of g_s (pcc_sparse_conv_head*, which run
 *   on the 8 N generative children of a level) visit the neighbours in
 *   SIBLINGS-FIRST order: first those inside the output row's own aligned block
 *   of 8 rows (nbr >> 3 == n >> 3: the children of the row's parent, itself
 *   included), k ascending, then the others, k ascending.  MinkowskiEngine's
 *   own order depends on atomics, so the order is this build's definition; it
 *   makes a parent's 8 x 8 sibling pairs one dense register-resident product
 *   (csrc/convup.h).  oracle/pcc_oracle.c states both orders.
 */
#ifndef PCC_H
#define PCC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCC_ABI_VERSION 1

/* error codes */
#define PCC_OK 0
#define PCC_E_ARG (-1)      /* bad argument / unsupported shape            */
#define PCC_E_HIP (-2)      /* HIP runtime error                           */
#define PCC_E_RANGE (-3)    /* coordinate / batch index out of range       */
#define PCC_E_DUP (-4)      /* duplicate coordinates in input              */
#define PCC_E_STREAM (-5)   /* corrupt or truncated bitstream              */
#define PCC_E_NOMEM (-6)

typedef struct pcc_ctx pcc_ctx;

int pcc_abi_version(void);
const char* pcc_last_error(void);

/* ctx = device + stream + scratch arena.  `stream` is a hipStream_t passed as
 * void* (NULL = default stream). */
pcc_ctx* pcc_create(int device, void* stream);
void pcc_destroy(pcc_ctx* ctx);
int pcc_set_stream(pcc_ctx* ctx, void* stream);
int pcc_sync(pcc_ctx* ctx);
/* start/stop a hipEvent pair on the ctx stream; elapsed returns ms of the last
 * completed pair (used by bench.py for the per-kernel roofline figure). */
int pcc_timer_start(pcc_ctx* ctx);
int pcc_timer_stop(pcc_ctx* ctx);
int pcc_timer_elapsed_ms(pcc_ctx* ctx, float* h_ms);

/* per-launch profiler: when enabled, every device entry point below brackets
 * its launches with a hipEvent pair on the ctx stream.  pcc_prof_get returns
 * the op name, its elapsed ms and four shape numbers (op-specific; for
 * sparse_conv: n_out, cin, cout, k_vol).  Used by bench.py for the roofline
 * figure of the dominant kernel. */
int pcc_prof_enable(pcc_ctx* ctx, int on);
/* restrict the records to entry points whose op name starts with the prefix
 * ("sparse_conv", "convT_gen", ...; NULL or "" = all) and, when d0 >= 0, whose
 * first recorded dimension (output rows) equals d0.  An event pair costs a few
 * microseconds of stream bubble, so a timed run brackets only what it reads. */
int pcc_prof_only(pcc_ctx* ctx, const char* h_op_prefix, int64_t d0);
int pcc_prof_count(pcc_ctx* ctx);
int pcc_prof_get(pcc_ctx* ctx, int i, char* h_op, int cap, float* h_ms,
                 int64_t* h_dims);
/* number of entries >= 0 (active pairs of a rule book) */
int pcc_count_nonneg(pcc_ctx* ctx, const int32_t* d_p, int64_t n,
                     int64_t* h_count);

/* ---- coordinate keys and ordering ------------------------------------- */

/* replaces: MinkowskiEngine coordinate-map insertion inside every
 * ME.SparseTensor(...) ctor (codec_pipeline.py:262,308; codec_parallel.py:296,
 * 309,411).  d_flag (int32[1], device) is OR-ed with 1 if any coordinate is
 * out of range. */
int pcc_morton_keys(pcc_ctx* ctx, const int32_t* d_coords, int64_t n,
                    uint64_t* d_keys, int32_t* d_flag);
/* inverse: keys -> (b,x,y,z) */
int pcc_keys_to_coords(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                       int32_t* d_coords);
/* replaces: the sortable value of shared/utils.py:131-132 / :160-161
 * (key = b*1e15 + x*1e10 + y*1e5 + z, int64) */
int pcc_linear_keys(pcc_ctx* ctx, const int32_t* d_coords, int64_t n,
                    int64_t* d_keys);
/* stable LSD radix sort of 64-bit keys (unsigned order; pass is_signed=1 for
 * int64 order) carrying the original row index.  d_keys is sorted in place,
 * d_perm[i] = original index of the i-th smallest key. */
int pcc_sort_pairs(pcc_ctx* ctx, uint64_t* d_keys, uint32_t* d_perm, int64_t n,
                   int is_signed);
/* replaces: utils.sort_tensor / utils.sort_points (shared/utils.py:116-165):
 * d_perm = argsort of the linear key. */
int pcc_sort_coords(pcc_ctx* ctx, const int32_t* d_coords, int64_t n,
                    uint32_t* d_perm);
/* dst[i,:] = src[perm[i],:], rows of row_bytes (multiple of 4) */
int pcc_gather_rows(pcc_ctx* ctx, const void* d_src, const uint32_t* d_perm,
                    int64_t n, int row_bytes, void* d_dst);
/* h_dup = 1 if two adjacent sorted keys are equal */
int pcc_check_unique(pcc_ctx* ctx, const uint64_t* d_sorted_keys, int64_t n,
                     int* h_dup);
/* per-batch row offsets of a key-sorted tensor: h_offsets[b] = first row with
 * batch >= b, for b in [0, n_batch]; (h_offsets[n_batch] == n) */
int pcc_batch_offsets(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                      int n_batch, int64_t* h_offsets);

/* ---- coordinate pyramid ------------------------------------------------ */

/* replaces: the output coordinate map of every stride-2 kernel-2 convolution
 * (g_a / h_a down stages, g_s.down_conv codec_parallel.py:302-303):
 * parents = unique(floor(c / 2ts) * 2ts).  child_shift = 3*log2(ts).
 * Outputs: d_pkeys (capacity n_cap >= n), d_nbr8 (capacity 8*n_cap) laid out
 * [8, M] with row pitch M = *h_n_out, d_parent_of[i] = parent row of input i.
 * Synchronises to return *h_n_out. */
int pcc_down_coords(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                    int child_shift, uint64_t* d_pkeys, int32_t* d_nbr8,
                    int64_t n_cap, int32_t* d_parent_of /* [n], nullable */,
                    int64_t* h_n_out);
/* The sizes of the pyramid above a sorted key set in one pass and one
 * synchronisation: h_counts[l] = number of distinct (key >> (child_shift +
 * 3 (l+1))), l = 0..levels-1, i.e. the *h_n_out of `levels` successive
 * pcc_down_coords calls; *h_dup (nullable) = 1 if two neighbouring keys are equal
 * (pcc_check_unique).  pcc_down_coords_known is pcc_down_coords with the parent
 * count m supplied, and does not synchronise. */
int pcc_level_counts(pcc_ctx* ctx, const uint64_t* d_sorted_keys, int64_t n,
                     int child_shift, int levels, int64_t* h_counts, int* h_dup);
int pcc_down_coords_known(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                          int child_shift, uint64_t* d_pkeys, int32_t* d_nbr8,
                          int64_t n_cap, int32_t* d_parent_of, int64_t m);
/* replaces: the generative transposed-convolution coordinate map (up stages
 * of h_s and g_s): children key = parent | o << (3*log2(ts/2)), row 8p+o. */
int pcc_up_coords(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                  int child_shift, uint64_t* d_ckeys);
/* the keys of the listed children only: d_ckeys[i] = key of row d_rows[i] = 8p + o of
 * pcc_up_coords' output (every d_rows[i] < 8n — the caller's contract, as for
 * pcc_gather_rows), without the 8n keys being written.  What the decoder keeps of
 * an up stage's candidates (codec_parallel.py:465-472: the pruned tensor's
 * coordinates). */
int pcc_up_coords_rows(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                       int child_shift, const uint32_t* d_rows, int64_t m,
                       uint64_t* d_ckeys);

/* ---- rule book (kernel map) and lookup --------------------------------- */

/* The same rule book WITHOUT hashing, derived from the parent level's rule
 * book: a child (parent p, octant o) has its neighbour at offset d in the child
 * o' of parent-neighbour p+D with, per axis, t = o+d, D = floor(t/2), o' = t&1.
 *  - up:   children are the 8 generative children of n_parents rows (child row
 *          8j+o).  If the parent level is a pruned subset of the level the
 *          rule book d_nbr_parent was built on, d_parent_rows[j] is its row
 *          there and d_remap maps rows of that level back to pruned rows (-1
 *          if dropped); pass both NULL when parent rows == rule-book rows.
 *  - down: children/parents linked by pcc_down_coords (d_parent_of, d_nbr8).
 *  - pcc_inverse_rows: remap[rows[j]] = j, -1 elsewhere (n entries; rows distinct,
 *    so with m == n every entry is written and nothing is preset). */
int pcc_derive_map_up(pcc_ctx* ctx, const int32_t* d_nbr_parent,
                      int64_t parent_pitch, const uint32_t* d_parent_rows,
                      const int32_t* d_remap, int64_t n_parents, int32_t* d_nbr);
int pcc_derive_map_down(pcc_ctx* ctx, const int32_t* d_nbr_parent,
                        int64_t n_parent, const int32_t* d_nbr8,
                        const int32_t* d_parent_of, const uint64_t* d_keys,
                        int64_t n, int child_shift, int32_t* d_nbr);
int pcc_inverse_rows(pcc_ctx* ctx, const uint32_t* d_rows, int64_t m, int64_t n,
                     int32_t* d_remap);

/* replaces: ME kernel-map generation for 3^3 stride-1 convolutions.
 * stride = tensor stride ts; d_nbr is [27, n].  d_keys = the rows of a coordinate
 * set: Morton-sorted, distinct (as pcc_sort_pairs / pcc_down_coords / pcc_up_coords
 * leave them); sets of <= 4096 rows are searched in LDS, larger ones hashed. */
int pcc_build_map(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n, int stride,
                  int32_t* d_nbr);
/* replaces: SparseTensor.features_at_coordinates (codec_pipeline.py:401,
 * codec_parallel.py:387) — exact-lattice lookup; d_rows[i] = row of query i in
 * the key set or -1. */
int pcc_lookup(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
               const uint64_t* d_qkeys, int64_t m, int32_t* d_rows);
/* dst[i,:] = rows[i] >= 0 ? src[rows[i],:] : 0 */
int pcc_gather_rows_or_zero(pcc_ctx* ctx, const float* d_src,
                            const int32_t* d_rows, int64_t m, int c,
                            float* d_dst);

/* ---- sparse network layers -------------------------------------------- */

/* replaces: MinkowskiConvolution forward (3^3 stride 1 with K=27 and a rule
 * book from pcc_build_map; 2^3 stride 2 with K=8 and the rule book from
 * pcc_down_coords).  On the matrix cores: cin = 4 or a multiple of 16 up to 128
 * with cout a multiple of 16 (pcc_conv_kernel_name); any (cin <= 128, cout <= 256)
 * on the scalar-fmaf path (same results). */
int pcc_sparse_conv(pcc_ctx* ctx, const float* d_in, int64_t n_in,
                    const int32_t* d_nbr, int k_vol, int64_t nbr_pitch,
                    int64_t n_out, const float* d_w, const float* d_bias,
                    int cin, int cout, int relu, float* d_out);
/* Diagnostic: name of the kernel pcc_sparse_conv (op 0), the siblings-first conv of
 * pcc_sparse_conv_head's generic form (op 1) or pcc_convT_gen (op 2) runs for a
 * shape with 16-byte aligned tensors: "k_gconv16" (32 -> 32 / 64; the name stands for
 * the family: op 0 launches of at most 2048 sixteen-row windows, 1024 for 32 -> 64, run
 * its no-compaction form k_gconv_rows16 — the name does not depend on the row count),
 * "k_gconv_gen"
 * (any widths that are multiples of 16: C_in <= 128, C_out <= 256), "k_gconv_first"
 * (4 -> multiples of 16), "k_convT16" (the up stage 32 -> 32), "k_convT_mfma" (other
 * multiples of 16 up to 128) — all on the
 * matrix cores — or the scalar-fmaf kernels ("k_gconv_scalar", "k_convT_scalar":
 * other shapes, PCC_FORCE_SCALAR=1).  Same results whichever runs. */
const char* pcc_conv_kernel_name(int op, int k_vol, int cin, int cout);
/* Layer weights, prepared once.  The (32,32) and (32,64) layers read their weights
 * as matrix-core operands from a copy in operand order (csrc/conv16.h).  A weight
 * tensor [k_vol][cin][cout] registered here — what load_model / model.to(device)
 * do once per layer (codec_pipeline.py:56-72) — is re-arranged now and found again
 * by its device pointer in every later pcc_sparse_conv* call on this ctx; for an
 * unregistered pointer the copy is made in front of each launch (same results).
 * Call again after the tensor's contents changed; pcc_conv_forget before its
 * memory is freed or reused.  Shapes without such a form: PCC_E_ARG. */
int pcc_conv_prepare(pcc_ctx* ctx, const float* d_w, int k_vol, int cin, int cout);
int pcc_conv_forget(pcc_ctx* ctx, const float* d_w);
/* the conv3 layer of a g_s stage (codec_parallel.py:469) with the 1x1 occupancy
 * head fused into its epilogue: d_head_out[n] = head_b[0] + sum_c fmaf(out[n][c],
 * head_w[c]) (c ascending) — bit-identical to pcc_linear(cout -> 1) applied to
 * d_out, without re-reading it.  Neighbours are visited SIBLINGS FIRST (the
 * arithmetic contract at the top of this file): first the pairs with
 * (input row >> 3) == (output row >> 3), k ascending, then the others, k
 * ascending.  On a generative level (input rows = output rows, in aligned blocks
 * of the 8 children of a parent — what g_s feeds it) that is "the row's own
 * parent first"; on any other input it is the same comparison of index values,
 * as deterministic, and what the oracle computes too — no check rejects it. */
int pcc_sparse_conv_head(pcc_ctx* ctx, const float* d_in, int64_t n_in,
                         const int32_t* d_nbr, int k_vol, int64_t nbr_pitch,
                         int64_t n_out, const float* d_w, const float* d_bias,
                         int cin, int cout, int relu, float* d_out,
                         const float* d_head_w, const float* d_head_b,
                         float* d_head_out);
/* building block of every compaction on the path (stride-2 parent dedup, top-k
 * pruning, radix-sort offsets, octree levels — the work torch.unique / ME's
 * coordinate manager do inside the reference's ME.SparseTensor and pruning
 * calls): exclusive prefix sum, d_out[i] = sum d_in[0..i), wrapping mod 2^32;
 * d_in and d_out may alias; d_total (device, nullable) receives the grand total.
 * Reduce-then-scan over 2048-element tiles. */
int pcc_exclusive_scan_u32(pcc_ctx* ctx, const uint32_t* d_in, uint32_t* d_out,
                           int64_t n, uint32_t* d_total);

/* The g_s stage form of the two above with the rule book formed on the fly.
 * The conv3 of a synthesis stage runs on the 8 generative children (row 8p+o) of
 * the n_parents rows of the level below; d_nbr_parent is THAT level's 27-offset
 * rule book ([27, parent_pitch]), and the child rule book (27 x 8 n_parents
 * int32, 352 MB for the bench frame) is never materialised: same values as
 * pcc_derive_map_up(parent book) followed by pcc_sparse_conv_head, bit for bit.
 * cin = cout = 32, K = 27 only (the shape of the model's g_s stages).
 * pcc_subset_map_up gives the rule book of the level that survives the top-k
 * pruning of those children (rows d_keep, ascending; d_remap from
 * pcc_inverse_rows over the 8 n_parents candidates), again from the parent
 * book: d_nbr is [27, n_keep]. */
int pcc_sparse_conv_head_up(pcc_ctx* ctx, const float* d_in, int64_t n_parents,
                            const int32_t* d_nbr_parent, int64_t parent_pitch,
                            const float* d_w, const float* d_bias, int relu,
                            float* d_out, const float* d_head_w,
                            const float* d_head_b, float* d_head_out);
int pcc_subset_map_up(pcc_ctx* ctx, const int32_t* d_nbr_parent,
                      int64_t parent_pitch, const uint32_t* d_keep,
                      const int32_t* d_remap, int64_t n_keep, int32_t* d_nbr);

/* rule book of a SUBSET of the output rows of a layer: d_nbr_out[k][j] =
 * d_nbr[k][d_rows[j]], or -1 where d_rows[j] < 0 ([k_vol, m], pitch m);
 * d_self (nullable) [m] = j where d_rows[j] >= 0, else -1.  With it a conv whose
 * output is only sampled afterwards (h_s followed by features_at_coordinates at
 * the latent's coordinates, codec_pipeline.py:401 / codec_parallel.py:387) is
 * evaluated at the sampled rows alone: same bits per row, a quarter of the rows;
 * pcc_gather_rows_or_zero(out, d_self) then zeroes the rows that were absent. */
int pcc_gather_map_columns(pcc_ctx* ctx, const int32_t* d_nbr, int k_vol,
                           int64_t pitch, const int32_t* d_rows, int64_t m,
                           int32_t* d_nbr_out, int32_t* d_self);

/* replaces: MinkowskiGenerativeConvolutionTranspose forward (kernel 2,
 * stride 2): out[8p+o] = W[o]^T in[p] + bias. */
int pcc_convT_gen(pcc_ctx* ctx, const float* d_in, int64_t n_in,
                  const float* d_w, const float* d_bias, int cin, int cout,
                  int relu, float* d_out);
/* 1x1 convolution (MinkowskiLinear / kernel-1 conv): occupancy and colour
 * heads of g_s. */
int pcc_linear(pcc_ctx* ctx, const float* d_in, int64_t n, const float* d_w,
               const float* d_bias, int cin, int cout, int relu, float* d_out);
/* pcc_convT_gen (cin = cout = 32) whose parent p is row d_rows[p] of d_in: the up
 * stage that follows a pruning, on the kept rows in place (same bits as
 * pcc_gather_rows followed by pcc_convT_gen). */
int pcc_convT_gen_gather(pcc_ctx* ctx, const float* d_in, const uint32_t* d_rows,
                         int64_t n_in, const float* d_w, const float* d_bias,
                         int relu, float* d_out);
/* pcc_linear (cin = 32, cout <= 8) applied to the rows d_rows[0..n) of d_in:
 * d_out[j] = linear(d_in[d_rows[j]]) — the colour head on the voxels kept by the
 * last pruning, without materialising their gathered feature rows (same bits as
 * pcc_gather_rows followed by pcc_linear). */
int pcc_linear_gather(pcc_ctx* ctx, const float* d_in, const uint32_t* d_rows,
                      int64_t n, const float* d_w, const float* d_bias, int cout,
                      int relu, float* d_out);

/* replaces: the per-frame top-k occupancy pruning inside model.g_s(y_hat,k=ks)
 * (codec_parallel.py:469): within each batch segment keep the k[b] rows with
 * the largest logit (ties: lower row first).  d_keep_rows receives the kept
 * row indices in ascending order; *h_n_keep their number.  h_offsets as from
 * pcc_batch_offsets (n_batch+1 entries), h_k n_batch entries.  The number of
 * kept rows is sum_f min(k[f], rows of frame f) by construction (no read-back);
 * h_n_keep may be NULL.  The call does not synchronise the stream for GOPs of
 * up to 8 frames. */
int pcc_topk_prune(pcc_ctx* ctx, const float* d_logits, int64_t n, int n_batch,
                   const int64_t* h_offsets, const int64_t* h_k,
                   uint32_t* d_keep_rows, int64_t* h_n_keep);
/* the same, and d_remap[r] (n entries, nullable) = position of row r in
 * d_keep_rows, or -1 for a row that is not kept: what pcc_inverse_rows makes of
 * d_keep_rows, written by the placement itself (the rule book of the pruned
 * level wants it, pcc_subset_map_up). */
int pcc_topk_prune_map(pcc_ctx* ctx, const float* d_logits, int64_t n, int n_batch,
                       const int64_t* h_offsets, const int64_t* h_k,
                       uint32_t* d_keep_rows, int64_t* h_n_keep, int32_t* d_remap);

/* ---- entropy-model device kernels ------------------------------------- */

/* replaces: EntropyBottleneck.compress symbol formation + dequantised z_hat
 * (codec_pipeline.py:303-306): sym[c*n+i] = rint(z[i][c] - med[c]) (int32,
 * channel-major), zhat[i][c] = sym + med[c]. */
int pcc_factorized_quant(pcc_ctx* ctx, const float* d_z, int64_t n, int c,
                         const float* d_med, int32_t* d_sym, float* d_zhat);
/* decoder side: zhat[i][c] = sym[c*n+i] + med[c] (codec_parallel.py:307-314) */
int pcc_factorized_dequant(pcc_ctx* ctx, const int32_t* d_sym, int64_t n,
                           int c, const float* d_med, float* d_zhat);
/* replaces: build_indexes + quantize of GaussianConditional.compress for Q
 * quality settings at once (codec_pipeline.py:407-430).
 *   params [n, 2c] = (scales_hat | means_hat) rows aligned with y rows
 *   scale  [q, c]  = scale_nn(q)+eps per quality
 *   table  [n_tab] ascending scale table (n_tab <= 64)
 *   sym,idx int32 [q, c, n] channel-major
 * sym = rint(y*s - mean*s), idx = (n_tab-1) - #{t in table[:-1] : max(sc*s,
 * table[0]) <= t}. */
int pcc_gaussian_quant(pcc_ctx* ctx, const float* d_y, const float* d_params,
                       int64_t n, int c, const float* d_scale, int q,
                       const float* d_table, int n_tab, int32_t* d_sym,
                       int32_t* d_idx);
/* element-wise forms behind the CompressAI-shaped tensor methods
 * gaussian_conditional.build_indexes(t) and EntropyModel.quantize(t, "symbols",
 * means) (any shape, n elements): idx = build_indexes(scales); sym = rint(x -
 * means) (means nullable). */
int pcc_build_indexes(pcc_ctx* ctx, const float* d_scales, int64_t n,
                      const float* d_table, int n_tab, int32_t* d_idx);
int pcc_quantize_symbols(pcc_ctx* ctx, const float* d_x, const float* d_means,
                         int64_t n, int32_t* d_sym);
/* compact form of the same: int16 symbols, uint8 indexes (3 instead of 8 bytes
 * per symbol across PCIe).  d_flag (int32[1]) is OR-ed with 1 if a symbol does
 * not fit int16; the caller then falls back to pcc_gaussian_quant. */
int pcc_gaussian_quant16(pcc_ctx* ctx, const float* d_y, const float* d_params,
                         int64_t n, int c, const float* d_scale, int q,
                         const float* d_table, int n_tab, int16_t* d_sym,
                         uint8_t* d_idx, int32_t* d_flag);
/* the same with int32 symbols and uint8 indexes, the element types the GPU
 * coder reads (pcc_rans_encode_dev): nothing overflows, nothing crosses PCIe */
int pcc_gaussian_quant_dev(pcc_ctx* ctx, const float* d_y, const float* d_params,
                           int64_t n, int c, const float* d_scale, int q,
                           const float* d_table, int n_tab, int32_t* d_sym,
                           uint8_t* d_idx);
int pcc_gaussian_indexes8(pcc_ctx* ctx, const float* d_params, int64_t n, int c,
                          const float* d_scale, const float* d_table, int n_tab,
                          uint8_t* d_idx);
/* decoder side (codec_parallel.py:394-409): indexes for one quality */
int pcc_gaussian_indexes(pcc_ctx* ctx, const float* d_params, int64_t n, int c,
                         const float* d_scale, const float* d_table, int n_tab,
                         int32_t* d_idx);
/* decoder side de-quantisation with offsets (codec_parallel.py:401-409):
 *   sigma = max(sc*s, bound); off = -(a / (b + sigma)), 0 where sym == 0
 *   yhat = sign(sym) * (|sym| + off) * (1/s) + mean */
int pcc_gaussian_dequant(pcc_ctx* ctx, const int32_t* d_sym,
                         const float* d_params, int64_t n, int c,
                         const float* d_scale, float bound, float off_a,
                         float off_b, float* d_yhat);

/* ---- host coders (no GPU needed) --------------------------------------- */

/* replaces: compressai.ans.RansEncoder.encode_with_indexes / RansDecoder.
 * decode_with_indexes (CompressAI 1.2.4, called from entropy_bottleneck /
 * gaussian_conditional .compress/.decompress: codec_pipeline.py:305-306,
 * 426-430; codec_parallel.py:307,400).
 *   cdfs   int32 [n_cdf, cdf_pitch] quantised CDFs (16-bit precision)
 *   sizes  int32 [n_cdf] cdf lengths, offsets int32 [n_cdf]
 * encode: returns bytes written into h_out (cap bytes) via *h_len. */
int pcc_rans_encode(const int32_t* h_sym, const int32_t* h_idx, int64_t n,
                    const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                    const int32_t* h_offsets, int n_cdf, uint8_t* h_out,
                    int64_t cap, int64_t* h_len);
int pcc_rans_decode(const uint8_t* h_in, int64_t len, const int32_t* h_idx,
                    int64_t n, const int32_t* h_cdfs, int cdf_pitch,
                    const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                    int32_t* h_sym);
/* the same for n_streams independent streams coded on n_streams host threads
 * (the Q quality settings of gaussian_model_step_batched). */
int pcc_rans_encode_multi(const int32_t* h_sym, const int32_t* h_idx, int64_t n,
                          int n_streams, const int32_t* h_cdfs, int cdf_pitch,
                          const int32_t* h_sizes, const int32_t* h_offsets,
                          int n_cdf, uint8_t* h_out, int64_t cap_each,
                          int64_t* h_lens);

/* the same coders for the compact element widths (int16 symbols / uint8
 * indexes on encode, uint8 indexes on decode) */
int pcc_rans_encode_multi16(const int16_t* h_sym, const uint8_t* h_idx,
                            int64_t n, int n_streams, const int32_t* h_cdfs,
                            int cdf_pitch, const int32_t* h_sizes,
                            const int32_t* h_offsets, int n_cdf, uint8_t* h_out,
                            int64_t cap_each, int64_t* h_lens);
int pcc_rans_decode8(const uint8_t* h_in, int64_t len, const uint8_t* h_idx,
                     int64_t n, const int32_t* h_cdfs, int cdf_pitch,
                     const int32_t* h_sizes, const int32_t* h_offsets, int n_cdf,
                     int32_t* h_sym);

/* Seek points of a host-coded stream (round 4; pcc_codec_set_seek_points puts them into a container trailer).
 * _encode_seek: pcc_rans_encode, and for every h_seek_index[k] (ascending) inside (0, n) the coder's state and the
 * number of 32-bit words of the stream a decoder has consumed when symbol h_seek_index[k] is the next it decodes
 * (0 / 0 for an index outside (0, n)).  _decode_range: the symbols [i_lo, i_hi) from such a point (state_in, word_in;
 * i_lo == 0 with word_in == 0: the head of the stream), written to h_sym[i_lo .. i_hi); *h_state_out / *h_word_out =
 * where the decoder stands behind symbol i_hi - 1 — equal to the next point's values when stream and points agree.
 * The pieces between seek points are independent: any number of threads may decode them at once. */
int pcc_rans_encode_seek(const int32_t* h_sym, const int32_t* h_idx, int64_t n,
                         const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                         const int32_t* h_offsets, int n_cdf, uint8_t* h_out, int64_t cap,
                         int64_t* h_len, const int64_t* h_seek_index, int n_seek,
                         uint64_t* h_seek_state, int64_t* h_seek_word);
int pcc_rans_decode_range(const uint8_t* h_in, int64_t len, const int32_t* h_idx, int64_t n,
                          const int32_t* h_cdfs, int cdf_pitch, const int32_t* h_sizes,
                          const int32_t* h_offsets, int n_cdf, int32_t* h_sym, int64_t i_lo,
                          int64_t i_hi, uint64_t state_in, int64_t word_in,
                          uint64_t* h_state_out, int64_t* h_word_out);

/* ---- range-ANS on the GPU: container version 1 (flagged extension) ------ */

/* Same call sites as the host coders above (codec_pipeline.py:305-306,426-430;
 * codec_parallel.py:307,398-400), different stream: the symbols of an array are
 * dealt to 64 rANS states per wave (csrc/rans_gpu.hip gives the layout), so the
 * coder runs on the device next to the kernels that make / consume the symbols
 * and the step has no serial host coder on its critical path.  The arithmetic of
 * a coding step (64-bit state, 16-bit CDFs, escape + 4-bit bypass) is that of
 * the reference's coder; a container whose y / z strings have this form carries
 * PCC_CONTAINER_V1 in the top byte of its first word (codec.hip) — the reference
 * decoder cannot read it, this library's decoders read both versions.
 *   tables : the CDF set in HBM (the arguments of pcc_rans_encode), made once
 *   d_sym  : int32 [n_streams, n]; d_idx uint8 [n_streams, n] table index per
 *            symbol, or NULL: index = position / idx_run (channel-major arrays)
 *   d_out  : n_streams streams at d_out + s * cap_each (device, 4-byte aligned);
 *            pcc_rans_dev_bound(n) bytes always suffice; h_lens[s] = bytes
 * encode synchronises the stream once (the lengths); decode does not: it takes
 * the header fields pcc_rans_stream_info checked on the host copy of the stream
 * and ORs d_status (int32, device) with 1 / 2 / 4 if a chunk runs out of words /
 * an escape is malformed / a table index is not below n_cdf (the last row is
 * used instead) — test it after the next synchronisation.  encode clamps table
 * indexes the same way; with d_idx == NULL it refuses n > idx_run * n_cdf.
 * The coding loops read d_idx and the stream as aligned 32-bit words: the
 * device allocations behind d_in (decode) and d_idx (both) must be readable up
 * to the next multiple of 4 bytes beyond their last byte (any hipMalloc'd or
 * framework-allocated buffer is). */
typedef struct pcc_rans_dev pcc_rans_dev;
pcc_rans_dev* pcc_rans_dev_create(const int32_t* h_cdfs, int cdf_pitch,
                                  const int32_t* h_sizes, const int32_t* h_offsets,
                                  int n_cdf);
void pcc_rans_dev_destroy(pcc_rans_dev* tables);
int64_t pcc_rans_dev_bound(int64_t n);
int pcc_rans_encode_dev(pcc_ctx* ctx, const pcc_rans_dev* tables,
                        const int32_t* d_sym, const uint8_t* d_idx, int64_t idx_run,
                        int64_t n, int n_streams, uint8_t* d_out, int64_t cap_each,
                        int64_t* h_lens);
int pcc_rans_stream_info(const uint8_t* h_in, int64_t len, int64_t* h_n,
                         int64_t* h_steps, int64_t* h_chunks);
int pcc_rans_decode_dev(pcc_ctx* ctx, const pcc_rans_dev* tables, const uint8_t* d_in,
                        int64_t len, int64_t n, int64_t steps, int64_t n_chunks,
                        const uint8_t* d_idx, int64_t idx_run, int32_t* d_sym,
                        int32_t* d_status);

/* replaces: utils.gpcc_encode / gpcc_decode (shared/utils.py:169-240), i.e.
 * the tmc3 subprocess: lossless octree occupancy coding of one frame's latent
 * coordinates.  The blob is opaque to the container (length-prefixed slot) and
 * is NOT tmc3-compatible.
 * Device part: occupancy bytes of every octree level of one frame.
 *   d_keys    the frame's rows of a Morton-sorted key array (stride-ts tensor)
 *   key_shift 3*log2(ts): leaf = (key >> key_shift) & (2^(3*depth) - 1)
 *   depth     log2 of the side of the aligned root cube (host computes it from
 *             the first and last key: floor(msb(first^last)/3)+1, min 1)
 *   d_occ     uint8  [sum of level node counts], root level first
 *   h_level_n int64  [depth] nodes per level (root first)
 * cap = capacity of d_occ in bytes (n*depth is always enough). */
int pcc_octree_levels(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                      int key_shift, int depth, uint8_t* d_occ, int64_t cap,
                      int64_t* h_level_n);
/* host part: adaptive binary range-ANS of the occupancy bytes */
int pcc_octree_pack(const uint8_t* h_occ, const int64_t* h_level_n, int depth,
                    int64_t n_points, const int32_t* h_origin, uint8_t* h_out,
                    int64_t cap, int64_t* h_len);
/* parse blob header: n_points, depth, origin[3] */
int pcc_octree_peek(const uint8_t* h_in, int64_t len, int64_t* h_n_points,
                    int* h_depth, int32_t* h_origin);
/* decode blob to Morton-ordered points int32 [n_points,3] (origin added) */
int pcc_octree_unpack(const uint8_t* h_in, int64_t len, int32_t* h_points,
                      int64_t cap_points);
/* the same, and the node count of every octree level (h_level_n[0 .. depth-1],
 * 16 entries): level depth-1 = the leaves' parents, depth-2 their grandparents,
 * i.e. the sizes of the stride-2 coordinate sets above the decoded points. */
int pcc_octree_unpack_levels(const uint8_t* h_in, int64_t len, int32_t* h_points,
                             int64_t cap_points, int64_t* h_level_n);

/* one-call forms of the geometry slot: pcc_octree_encode = root cube from the
 * first / last key + pcc_octree_levels + pcc_octree_pack (utils.gpcc_encode,
 * shared/utils.py:169-207); pcc_octree_decode = pcc_octree_peek (+ unpack when
 * h_points is given) (utils.gpcc_decode, shared/utils.py:210-240, without the
 * `* 8`). */
int pcc_octree_encode(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                      int key_shift, uint8_t* h_out, int64_t cap, int64_t* h_len);
int pcc_octree_decode(const uint8_t* h_in, int64_t len, int32_t* h_points,
                      int64_t cap_points, int64_t* h_n_points);

/* Blob version 2 (round 4; csrc/octree2.hip gives the layout): the occupancy
 * ENTROPY coder on the GPU too — the adaptive binary model of version 1, coded
 * by 64 rANS states per wave over runs of consecutive nodes, every state's
 * model starting from the frame's average probabilities (header).  BASELINE.json
 * configs[2] (geometry-only coding of a ~100k-point LiDAR sweep) is this path.
 * pcc_octree_encode writes version 2 for sets above PCC_OCTREE_V2_MIN_LEAVES
 * leaves and version 1 (serial host coder: fewer bytes on small sets, and what
 * the latent-sized slots of the codec use) below; _version forces one
 * (0 = that rule).  Version-2 blobs are DEcoded by the GPU as well and have no
 * host decoder: pcc_octree_decode / pcc_octree_unpack* refuse them with
 * PCC_E_STREAM, the forms below take a context and read both versions.
 *   _decode_ctx : points to the host (h_points NULL: only the count)
 *   _decode_dev : points left in HBM (int32 [n,3], Morton order, origin added);
 *                 h_level_n (16 entries, nullable) = nodes per octree level */
/* Blob version 3 (round 4; csrc/octree_host.cpp gives the layout): the frame's
 * leaves in K = min(8, n / 4096) parts (at least 2), each a complete version-1
 * blob under the frame's root cube, cut between grandparent cells — coded by K
 * workgroups + K host coders and decoded by K host decoders side by side (the
 * serial decoder of a latent-sized slot stood at the head of every decode with
 * the GPU waiting for it).  pcc_octree_encode writes it for sets of
 * PCC_OCTREE_V3_MIN_LEAVES leaves and more, up to PCC_OCTREE_V2_MIN_LEAVES;
 * every host decoder above reads it (pcc_octree_unpack_levels: levels depth-1
 * and depth-2 are the frame's, the levels above count a node once per part). */
#define PCC_OCTREE_V2_MIN_LEAVES 65536
#define PCC_OCTREE_V3_MIN_LEAVES 8192
int pcc_octree_encode_version(pcc_ctx* ctx, const uint64_t* d_keys, int64_t n,
                              int key_shift, int version, uint8_t* h_out,
                              int64_t cap, int64_t* h_len);
int pcc_octree_blob_version(const uint8_t* h_in, int64_t len);
int pcc_octree_decode_ctx(pcc_ctx* ctx, const uint8_t* h_in, int64_t len,
                          int32_t* h_points, int64_t cap_points,
                          int64_t* h_n_points);
int pcc_octree_decode_dev(pcc_ctx* ctx, const uint8_t* h_in, int64_t len,
                          int32_t* d_points, int64_t cap_points,
                          int64_t* h_n_points, int64_t* h_level_n);

/* ---- whole-GOP entry points ------------------------------ */

/* replaces: CompressionPipeline.compress() (sender/encoder/codec_pipeline.py:
 * 196-236, called from sender/encoder/encoder.py:139) and
 * DecompressionPipeline.decompress() (receiver/decoder/codec_parallel.py:141-171,
 * called from receiver/decoder/decoder.py:62) for one GOP, in native code: the
 * op-level entry points above driven in the reference's stage order.  The
 * containers are byte-identical to those of the Python mirror of the pipelines
 * (codec_pipeline.py / codec_parallel.py of this package), which drives the same
 * entry points one by one.
 *
 * A codec = model weights in HBM + one ctx (stream, scratch) + a device pool
 * for the tensors of one call + pinned staging: one codec per in-flight call
 * (the reference runs up to 3, sender/encoder/encoder.py:50).  Results (output
 * containers, reconstructed cloud) are owned by the codec and stay valid until
 * the next call on it.
 *
 * h_ckpt: "PCCW" blob = u32 count, then per tensor { u16 name_len, name, u8 dtype
 * (0 float32, 1 int32), u8 ndim, u32 dims[], u64 nbytes, pad to 8, data, pad to
 * 8 }, little-endian — the tensors of assets/demo_small.npz (DESIGN.md MODEL);
 * native.pack_checkpoint() writes it. */
typedef struct pcc_codec pcc_codec;
typedef struct pcc_buf {
  const uint8_t* data;
  int64_t len;
} pcc_buf;
typedef struct pcc_cloud_info {
  int64_t n_points;
  int32_t n_frames;         /* frames the container announces                      */
  int32_t n_offsets;        /* entries of h_offsets = frames with points, plus one */
  const int64_t* h_offsets; /* row range of every frame in d_coords / d_colors     */
  const int32_t* d_coords;  /* [n_points,4] rows (b,x,y,z), Morton order per frame */
  const float* d_colors;    /* [n_points,3] as the colour head leaves them         */
  double q_g, q_a;          /* quality setting read from the container             */
} pcc_cloud_info;

pcc_codec* pcc_codec_create(const void* h_ckpt, size_t n, int device, void* stream);
void pcc_codec_destroy(pcc_codec* codec);
pcc_ctx* pcc_codec_ctx(pcc_codec* codec); /* the codec's ctx (stream, profiler) */

/* Container version the encoder entry points of this codec write (default 0).
 *   PCC_CONTAINER_V0  the reference's layout, byte for byte (make_bitstream_batched,
 *                     codec_pipeline.py:464-517): y and z strings are single rANS
 *                     streams, coded on the host (CompressAI's format)
 *   PCC_CONTAINER_V1  the same fields, the top byte of the first word (num_frames,
 *                     at most 65535) set to 1, and the y / z strings in the GPU
 *                     coder's wave-interleaved form (pcc_rans_encode_dev): no serial
 *                     host coder on the path.  The reference's decoder cannot read
 *                     it; pcc_decode_gop reads both versions. */
#define PCC_CONTAINER_V0 0
#define PCC_CONTAINER_V1 1
int pcc_codec_set_container_version(pcc_codec* codec, int version);
/* Seek points of the host-coded y strings (round 4).  pieces > 1: a version-0 container written by this codec carries,
 * BEHIND its last frame record, the trailer "PCSK" | int32 count | count x (int32 symbol index | uint64 coder state |
 * int32 32-bit words consumed) — where a decoder of the y string stands at `pieces` - 1 cuts of the symbol array
 * (multiples of 64 near k n / pieces; none for strings under 65536 symbols).  Everything in front of the trailer is
 * the reference's container byte for byte, and the reference's reader never reaches the trailer: it reads exactly
 * num_frames frame records (receiver/decoder/codec_parallel.py:200-213).  pcc_decode_gop* decodes the pieces between
 * the points on as many host threads, checks every piece's end against the next point and falls back to the serial
 * decode when a trailer does not parse or check out (a container without one is decoded serially as before).
 * 0 (default): no trailer. */
int pcc_codec_set_seek_points(pcc_codec* codec, int pieces);

/* d_coords int32 [n,4] rows (b,x,y,z), b in [0,n_frames); d_feats float32 [n,4]
 * = (1,r,g,b) (codec_pipeline.py:258); h_q [n_q,2] = (q_g,q_a) per quality
 * (shared/config.yaml:12-15).  h_out receives n_q containers; h_k (nullable)
 * int64 [3,n_frames] = k[scale][frame]; h_stage_s (nullable) double[7] seconds =
 * analysis, hyper_analysis, factorized_model, hyper_synthesis,
 * geometry_compression, gaussian_model, bitstream_writing.
 * Errors: PCC_E_RANGE / PCC_E_DUP for bad coordinates, like the SparseTensor
 * constructor of the Python mirror. */
int pcc_encode_gop(pcc_codec* codec, const int32_t* d_coords,
                   const float* d_feats, int64_t n, int n_frames,
                   const double* h_q, int n_q, pcc_buf* h_out, int64_t* h_k,
                   double* h_stage_s);
/* The same call on the frames as the capture stage produces them
 * (sender/capturer/capturer.py:111-126; consumed by unpack_batch,
 * codec_pipeline.py:243-262, and utils.stack_tensors, shared/utils.py:10-42):
 * per frame f, h_d_points[f] = device array [h_n[f],3] of int16 (points_i16 != 0)
 * or int32 voxel coordinates, h_d_colors[f] = device array [h_n[f],3] of float64
 * (colors_f64 != 0) or float32 colours in [0,1].  The batch column, the casts and
 * the (1,r,g,b) rows are formed inside the key and gather kernels; at most 32
 * frames per call (more: concatenate and use pcc_encode_gop).  The h_* tables are
 * host arrays of n_frames entries.  Same outputs and errors as pcc_encode_gop. */
int pcc_encode_gop_frames(pcc_codec* codec, const void* const* h_d_points,
                          int points_i16, const void* const* h_d_colors,
                          int colors_f64, const int64_t* h_n, int n_frames,
                          const double* h_q, int n_q, pcc_buf* h_out,
                          int64_t* h_k, double* h_stage_s);
/* The same with the frame arrays still in HOST memory, as compress(gop) receives
 * them from the capturer over ZeroMQ (codec_pipeline.py:243-262): h_points[f] /
 * h_colors[f] are host arrays.  The library uploads them itself and overlaps the
 * two PCIe legs with the start of the path: the points go up first, the Morton
 * keys and their sort run while the (4x larger, float64) colours follow. */
int pcc_encode_gop_host_frames(pcc_codec* codec, const void* const* h_points,
                               int points_i16, const void* const* h_colors,
                               int colors_f64, const int64_t* h_n, int n_frames,
                               const double* h_q, int n_q, pcc_buf* h_out,
                               int64_t* h_k, double* h_stage_s);
/* h_stage_s (nullable) double[6] seconds = bitstream_reading,
 * geometry_decompression, factorized_model, hyper_synthesis, guassian_model,
 * synthesis_transform.  Truncated / inconsistent containers: PCC_E_STREAM. */
int pcc_decode_gop(pcc_codec* codec, const uint8_t* h_in, int64_t len,
                   pcc_cloud_info* h_info, double* h_stage_s);
/* copy the cloud of the last pcc_decode_gop into caller-owned device buffers
 * (int32 [n_points,4], float32 [n_points,3]; either may be NULL); returns when
 * the copies are complete */
int pcc_decode_fetch(pcc_codec* codec, int32_t* d_coords, float* d_colors);
/* the same cloud as pack_batches returns it (codec_parallel.py:474-502):
 * points int32 [n_points,3] (no batch column; frame i = rows h_offsets[i] ..
 * h_offsets[i+1]), colours float32 [n_points,3] with NaN -> 0 and
 * clip(c * 255, 0, 255) / 255 applied on the device.  The destinations may be
 * device or host buffers (hipMemcpyDefault); returns when they are filled. */
int pcc_decode_fetch_packed(pcc_codec* codec, int32_t* points, float* colors);
/* decompress() into the caller's arrays in one call (codec_parallel.py:141-171 +
 * pack_batches 474-502): pcc_decode_gop with pcc_decode_fetch_packed's kernel
 * and transfers queued behind the last layer — no synchronisation and no trip
 * through the caller between the two.  points / colors hold cap_points rows
 * (host or device memory); pcc_container_points gives the number of points a
 * container announces (sum over frames of its finest k, an upper bound of what
 * is decoded; host-only parse, nothing beyond the slot bounds is validated).
 * More points decoded than cap_points: PCC_E_ARG, nothing is written.  Any
 * other error (PCC_E_STREAM from a malformed version-1 stream is only known when
 * the GPU coder's status word comes back, after the transfers were queued):
 * the contents of points / colors are undefined. */
int pcc_container_points(const uint8_t* h_in, int64_t len, int64_t* h_n_points,
                         int32_t* h_n_frames);
int pcc_decode_gop_packed(pcc_codec* codec, const uint8_t* h_in, int64_t len,
                          int32_t* points, float* colors, int64_t cap_points,
                          pcc_cloud_info* h_info, double* h_stage_s);

/* ---- capture pre-step ------------------------------- */

/* replaces: the voxelisation the capturer does per camera frame with numpy +
 * Open3D (sender/capturer/capturer.py:88-126).  Input: ZED XYZRGBA float32
 * [m,4] (colour packed in the 4th float, r = bits 0-7, g = 8-15, b = 16-23).
 *  pcc_vox_valid : valid[i] = finite && norm <= depth_clip; returns the
 *                  per-axis minimum of the valid points and their number
 *  pcc_vox_keys  : Open3D voxel index of every valid point packed as
 *                  ix<<42|iy<<21|iz (all ones for invalid points); d_flag is
 *                  OR-ed with 1 if an index does not fit 21 bits
 *  pcc_vox_mean  : on keys sorted with pcc_sort_pairs (d_perm = its permutation,
 *                  first n_valid entries): per voxel, mean position / colour in
 *                  double (input order) -> integer voxel rint(mean/voxel_size)
 *                  as rows (0,x,y,z) and float64 colours in [0,1]
 *  pcc_unique_rows: indices of the first row of every run of equal rows in an
 *                  array sorted so that duplicates are adjacent */
int pcc_vox_valid(pcc_ctx* ctx, const float* d_xyzrgba, int64_t m,
                  float depth_clip, uint8_t* d_valid, float* h_min_bound,
                  int64_t* h_n_valid);
int pcc_vox_keys(pcc_ctx* ctx, const float* d_xyzrgba, const uint8_t* d_valid,
                 int64_t m, const double* h_voxel_min_bound, double voxel_size,
                 uint64_t* d_keys, int32_t* d_flag);
int pcc_vox_mean(pcc_ctx* ctx, const float* d_xyzrgba,
                 const uint64_t* d_sorted_keys, const uint32_t* d_perm,
                 int64_t n_valid, double voxel_size, int32_t* d_out_coords,
                 double* d_out_colors, int64_t cap, int64_t* h_n_voxels);
int pcc_unique_rows(pcc_ctx* ctx, const int32_t* d_sorted_coords, int64_t n,
                    uint32_t* d_rows, int64_t* h_n_unique);

#ifdef __cplusplus
}
#endif
#endif /* PCC_H */
