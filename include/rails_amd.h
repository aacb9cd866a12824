/* rails_amd — C ABI of the MI355X-native Mixture-of-Logits (MoL) retrieval path.
 *
 * The reference (bailuding/rails) is pure Python/PyTorch and has no FFI of its own; the "interface
 * each entry point replaces" is therefore a Python call site of the reference, cited per function as
 * `path:line` inside the reference repository.  INTEGRATION.md shows the ctypes binding a maintainer
 * adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM of the current HIP device) unless stated otherwise;
 *     matrices are dense row-major exactly as the reference's state_dict() / tensors hold them;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every call only enqueues work on it: no allocation, no synchronisation, no host copies;
 *   - return value 0 on success, a negative RAILS_E* code otherwise; rails_last_error() returns
 *     a thread-local human-readable message for the last failure;
 *   - `run_if` (where an entry point has it) is the LAUNCH PREDICATE: NULL, or an int32 in device memory that every kernel of the
 *     call reads when it starts, in stream order -- the call is a no-op unless it is non-zero.  It is how a caller enqueues a
 *     fallback behind a device-side verdict without reading the verdict on the host (rails_rescore_verdict and rails_range_flag_i32
 *     write such flags).  No counterpart in the reference;
 *   - packed buffers (`gate pack`, `item index`, `query pack`) are opaque fp32 blobs whose sizes
 *     come from the *_floats() helpers; they are only valid for the shape they were built for.
 */
#ifndef RAILS_AMD_H_
#define RAILS_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this interface: bumped whenever a struct gains a field or an entry point changes meaning (round 2 -> 3: 3, the
 * structs of round 2 carried no version; 5: the launch predicate became the explicit `run_if` argument of the entry points that
 * honour it and the per-thread rails_set_run_predicate is gone -- the library keeps no state between calls but the last error;
 * 6: rails_mol_coarse_topk gained its out_of_range output and its optional int8 pre-filter (rails_mol_coarse_prefilter_*),
 * rails_topk_candidates is new, rails_mol_score_indexed takes any n_cand; 7: rails_rescore_verdict gained its guard arguments and state[7], the rails_*_probe_* entry points are new, and rails_mol_score_topk / _survivors / rails_select_survivors -- the selection fused into the scoring kernels, 0.9 % slower than the dense kernels + rails_topk wherever it was measured -- are gone;
 * 8: rails_mol_score_dense_upper[_supported] and rails_mol_index_rows_* / rails_mol_score_indexed_rows are new, rails_rescore_select gained one_sided;
 * 9: rails_candidates_* -- the threshold selection and the fused finish of the proved exact top-k -- and rails_merge_candidates_verdict
 * are new, rails_mol_score_indexed_rows gained cand_counts, the component table became item-group-major (rails_mol_component_build
 * gained n_total / first_item, rails_mol_component_topk its out_of_range flag, rails_mol_component_topk_capacity is new);
 * 10: rails_topk_candidates_filtered, rails_rerank_topk_filtered / rails_rerank_workspace_bytes and rails_mol_coarse_topk_capacity are new).  A binding checks rails_abi_version() == RAILS_ABI_VERSION at load time: callers built
 * against an older header pass shorter structs, and the library would read the new fields from whatever follows them. */
#define RAILS_ABI_VERSION 10
int rails_abi_version(void);

#define RAILS_OK 0
#define RAILS_EINVAL (-22)       /* bad argument (null pointer, non-positive size, k > n ...) */
#define RAILS_ENOTSUP (-95)      /* shape / option outside what the HIP kernels implement */
#define RAILS_ENOMEM (-12)       /* caller-provided workspace too small */
#define RAILS_ELAUNCH (-5)       /* HIP launch failure (message carries hipGetErrorString) */

#define RAILS_GEGLU 0
#define RAILS_SWIGLU 1

/* rails_mol_shape.precision -- the arithmetic of the fused scoring pass and, with it, the FORMAT of the three packed
 * buffers (gate pack, query pack, item index); every buffer must be built and used with the same value.
 *   RAILS_PRECISION_FP32   exact fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 fragments.  The parity path and the default.
 *   RAILS_PRECISION_F16X3  every operand of the three contractions is stored as f16 hi + f16 lo (same bytes as the fp32
 *                          fragment it replaces) and every product block costs three f16 MFMAs (lo*hi, hi*lo, hi*hi),
 *                          fp32 accumulate: ~22 significant bits per product, same 1e-4 logit bar, ~3-4x the speed.
 *                          The item index then holds Ex to 22 bits (rails_mol_index_unpack returns hi + lo);
 *                          rails_mol_coarse_build / rails_mol_component_build take an fp32-format index only.
 *   RAILS_PRECISION_F16X1  the F16X3 buffers (same format, interchangeable) scored with the hi * hi product only: plain f16
 *                          operands, fp32 accumulate, 1.6x the speed of F16X3, logits ~1e-2 off (NOT within the 1e-4 bar).
 *                          It exists as the first pass of a speculate-then-verify top-k whose candidates are re-scored in
 *                          fp32 (rails_rescore_select; rails_amd precision "f16-exact"), not as a result-producing mode. */
#define RAILS_PRECISION_FP32 0
#define RAILS_PRECISION_F16X3 1
#define RAILS_PRECISION_F16X1 2

/* rails_mol_shape.gating_combination: how the three gate parts become mixture weights (MoLGatingFn.forward).
 *   GLU_SILU  g = gq * gi + gqi;  w = g * sigmoid(g)      (every shipped config; needs all three parts)
 *   NONE      w = [gq] + [gi] + gqi                        (absent parts contribute nothing); exact-fp32 precision only */
#define RAILS_COMBINE_GLU_SILU 0
#define RAILS_COMBINE_NONE 1
#define RAILS_MAX_UID_TABLES 4

/* Hyper-parameters of one MoL module; field names follow create_mol_interaction_module
 * (modeling/similarity_utils.py:42-70). */
typedef struct rails_mol_shape {
  int32_t query_embedding_dim;
  int32_t item_embedding_dim;
  int32_t dot_product_dimension;      /* d */
  int32_t query_dot_product_groups;   /* P_Q */
  int32_t item_dot_product_groups;    /* P_X */
  int32_t query_hidden_dim;           /* GLU width of the query projection (512); <= 0: plain Linear (similarity_utils.py:108-116) */
  int32_t gating_query_hidden_dim;    /* 128 */
  int32_t gating_item_hidden_dim;     /* 128 */
  int32_t gating_qi_hidden_dim;       /* 128 */
  int32_t query_nonlinearity;         /* RAILS_GEGLU | RAILS_SWIGLU */
  int32_t num_uid_tables;             /* len(uid_embedding_hash_sizes) */
  int32_t dot_product_l2_norm;        /* 0 | 1 */
  float temperature;                  /* 0.05 */
  float eps;                          /* 1e-6 */
  int32_t precision;                  /* RAILS_PRECISION_FP32 | RAILS_PRECISION_F16X3 | RAILS_PRECISION_F16X1 */
  int32_t item_hidden_dim;            /* > 0: the item projection has a GLU hidden layer of this width (similarity_utils.py:127-143) */
  int32_t item_nonlinearity;          /* RAILS_GEGLU | RAILS_SWIGLU, used when item_hidden_dim > 0 */
  int32_t gating_combination;         /* RAILS_COMBINE_GLU_SILU | RAILS_COMBINE_NONE (similarity_fn.py:175-197) */
  int32_t gating_has_query;           /* 0: no query-only gate part (gating_query_fn = False); only with RAILS_COMBINE_NONE */
  int32_t gating_has_item;            /* 0: no item-only gate part  (gating_item_fn = False);  only with RAILS_COMBINE_NONE */
} rails_mol_shape;

/* Raw module weights, one pointer per state_dict() tensor (SURVEY.md section 8b lists the keys). */
typedef struct rails_mol_weights {
  const float* q_glu_w;   /* _query_embeddings_fn._query_emb_proj_module.1._w   (D_q, 2*query_hidden); NULL if query_hidden_dim <= 0 */
  const float* q_glu_b;   /* ...1._b                                            (2*query_hidden)      */
  const float* q_proj_w;  /* ...2.weight  (d*(P_Q-u), query_hidden);  query_hidden_dim <= 0: ...1.weight (d*(P_Q-u), D_q) */
  const float* q_proj_b;  /* ...2.bias / ...1.bias                                                    */
  const float* uid_table[RAILS_MAX_UID_TABLES];  /* _uid_embeddings_{i}.weight   (hash_i + 1, d)       */
  int64_t uid_hash_size[RAILS_MAX_UID_TABLES];
  const float* i_proj_w;  /* _item_embeddings_fn._item_emb_proj_module.1.weight  (P_X*d, D_i)         */
  const float* i_proj_b;
  const float* gq_w1;     /* _gating_fn._query_only_partial_module.0.weight      (H_q, D_q)           */
  const float* gq_b1;
  const float* gq_w2;     /* ...2.weight                                          (L, H_q)            */
  const float* gi_w1;     /* _gating_fn._item_only_partial_module.1.weight       (H_i, D_i)           */
  const float* gi_b1;
  const float* gi_w2;     /* ...3.weight                                          (L, H_i)            */
  const float* gqi_w1;    /* _gating_fn._qi_partial_module.1.weight              (H, L); gating_qi_hidden_dim <= 0 (pair gate without hidden
                           * layer, similarity_utils.py:199-206): the single Linear's (L, L) weight, gqi_b1 its (L) bias, gqi_w2 / gqi_b2 NULL */
  const float* gqi_b1;
  const float* gqi_w2;    /* ...3.weight                                          (L, H)              */
  const float* gqi_b2;
  const float* i_glu_w;   /* item_hidden_dim > 0: _item_embeddings_fn._item_emb_proj_module.1._w (D_i, 2*item_hidden); then      */
  const float* i_glu_b;   /* ...1._b, and i_proj_w / i_proj_b are ...2.weight (P_X*d, item_hidden) / ...2.bias; else NULL  */
} rails_mol_weights;

/* Thread-local message of the last failed call ("" if none). */
const char* rails_last_error(void);

/* Library / device probe: number of compute units of the current device (256 on MI355X), or < 0. */
int rails_device_compute_units(void);

/* 1 if the fused scoring kernel is compiled for this shape, else 0 (rails_last_error() says why). */
int rails_mol_shape_supported(const rails_mol_shape* shape);

/* ---- index build -----------------------------------------------------------------------------
 * Replaces what MoLTopKModule.__init__ (rails/indexing/mol_top_k.py:29-81) keeps per corpus, plus the
 * item-side work the reference redoes on every forward: RecoMoLItemEmbeddingsFn.forward
 * (rails/similarities/mol/item_embeddings_fns.py:149-183) and the item-only gate
 * (rails/similarities/mol/similarity_fn.py:170-171). */

/* fp32 count of the packed pair-gate weights. */
size_t rails_mol_gate_pack_floats(const rails_mol_shape* shape);
/* Permute the pair-gate MLP (gqi_*) into MFMA fragment order. */
int rails_mol_pack_gate_weights(const rails_mol_shape* shape, const rails_mol_weights* w, float* gate_pack,
                                void* stream);

/* fp32 count of the item index for n_items items (tiles of 32 items, last tile zero padded). */
size_t rails_mol_index_floats(const rails_mol_shape* shape, int64_t n_items);
/* items: (n_items, D_i) row-major.  Writes component embeddings Ex (l2-normalised) and the item gate gi
 * of every item into `index` in tile/fragment order. */
int rails_mol_index_build(const rails_mol_shape* shape, const rails_mol_weights* w, const float* items,
                          int64_t n_items, float* index, void* stream);
/* Inverse view for accessors/tests: plain Ex (n_items, P_X, d) and/or gi (n_items, L); either may be NULL.
 * Mirrors MoLSimilarity.get_item_component_embeddings (similarity_fn.py:294-339). */
int rails_mol_index_unpack(const rails_mol_shape* shape, const float* index, int64_t n_items, float* ex_out,
                           float* gi_out, void* stream);
/* Gather rows of an index into a new tile-packed index: out tile layout over `n_rows * n_cand` items
 * taken at cand_idx[r * n_cand + j] (positions in `index`, int64).  n_cand must be a multiple of 32.
 * Replaces self._item_embeddings[idx] (mol_top_k.py:361-363) for the rerank pass. */
int rails_mol_index_gather(const rails_mol_shape* shape, const float* index, int64_t n_items,
                           const int64_t* cand_idx, int64_t n_rows, int64_t n_cand, float* out_index,
                           void* stream);

/* ---- query side ------------------------------------------------------------------------------
 * Replaces RecoMoLQueryEmbeddingsFn.forward (rails/similarities/mol/query_embeddings_fns.py:175-254,
 * GLU at rails/similarities/layers.py:19-74) and the query-only gate (similarity_fn.py:166-169). */
/* Floats of the query pack: Eq fragments of ceil(batch / (32/P_Q)) query groups, gq fragments of `batch` rows, and the
 * prologue's own scratch (GLU output, first gate layer, raw projection and raw gate of ceil(batch/32)*32 rows). */
size_t rails_mol_query_pack_floats(const rails_mol_shape* shape, int32_t batch);
/* queries: (batch, D_q); user_ids: (batch) int64 or NULL when num_uid_tables == 0.
 * eq_out (batch, P_Q, d) and gq_out (batch, L) are optional plain copies (NULL to skip). */
int rails_mol_query_prologue(const rails_mol_shape* shape, const rails_mol_weights* w, const float* queries,
                             const int64_t* user_ids, int32_t batch, float* query_pack, float* eq_out,
                             float* gq_out, void* stream);
/* The same launch(es), writing the pack TWICE: query_pack in shape->precision's format and query_pack_other in the other
 * one (fp32 fragments <-> f16 hi/lo fragments; same values, rails_mol_query_pack_floats floats each).  For callers that score
 * one batch in two precisions (the verified fast modes: f16 first pass, fp32 re-scoring of the candidates). */
int rails_mol_query_prologue_both(const rails_mol_shape* shape, const rails_mol_weights* w, const float* queries,
                                  const int64_t* user_ids, int32_t batch, float* query_pack, float* query_pack_other,
                                  void* stream);

/* ---- scoring ---------------------------------------------------------------------------------
 * Replaces MoLSimilarity.forward for the shared-corpus case (similarity_fn.py:341-413, B' == 1):
 * logits[b * ld + x] for b < batch, x < n_items. */
int rails_mol_score_dense(const rails_mol_shape* shape, const float* gate_pack, const float* query_pack,
                          int32_t batch, const float* index, int64_t n_items, float* logits, int64_t ld,
                          const int32_t* run_if, void* stream);
/* The first pass of the proved exact top-k with a PER-PAIR error bound (no counterpart in the reference, which scores every item in one
 * precision: rails/indexing/mol_top_k.py:99-130): precision F16X3 only; logits[b * ld + x] = s + (ub2 c + ub1) c + ub0 with s the f16x3 logit
 * rails_mol_score_dense writes and c = max_l |cl_l| over the pair's P_Q * P_X cross logits as the kernel computed them.  With the
 * coefficients of rails_amd/f16x3_bound.py upper_bound_poly (>= 0, finite) the value is an UPPER BOUND of the pair's fp32-kernel logit: the
 * bound on |f16x3 - fp32| is quadratic in the magnitude of the cross logits, and the a-priori |cl| <= 1/tau it is otherwise evaluated at is
 * 2-3 x what the pairs of a corpus reach (config 4: one eps = 3.0 for every pair needs > 60 000 candidates per query at 12.5 M items, the
 * per-pair bound 730-900).  Built for the f16x3 kernels of the BASELINE shapes (the 256-logit team kernel and the 8-wave register-resident units);
 * rails_mol_score_dense_upper_supported says whether a shape has it. */
int rails_mol_score_dense_upper_supported(const rails_mol_shape* shape);
int rails_mol_score_dense_upper(const rails_mol_shape* shape, const float* gate_pack, const float* query_pack, int32_t batch, const float* index,
                                int64_t n_items, float ub2, float ub1, float ub0, float* logits, int64_t ld, const int32_t* run_if, void* stream);
/* Per-row candidates (B' == B branch, similarity_fn.py:397-402): `cand_index` was produced by
 * rails_mol_index_gather with n_rows == batch; logits[b * ld + j] for j < n_cand. */
int rails_mol_score_candidates(const rails_mol_shape* shape, const float* gate_pack, const float* query_pack,
                               int32_t batch, const float* cand_index, int64_t n_cand, float* logits,
                               int64_t ld, void* stream);
/* Per-row candidates scored IN PLACE: logits[b][j] = MoL(query b, item positions[b][j]) with the item operands read straight from the
 * shared index -- rails_mol_index_gather + rails_mol_score_candidates without the gathered copy and its launch (same arithmetic per
 * pair, same bits).  Exact-fp32 shapes on the independent-wave shell (rails_mol_score_indexed_supported != 0); the 256-logit shape
 * and the f16 precisions gather.  positions must lie in [0, n_items) (they are clamped, not masked); any n_cand >= 1. */
int rails_mol_score_indexed_supported(const rails_mol_shape* shape, int32_t batch, int64_t n_cand);
/* The same re-scoring from a ROW-MAJOR copy of the fp32 index: rails_mol_index_rows_build writes item i's operands as rails_mol_index_rows_floats /
 * n_items consecutive floats (fragment slot s, lane half h as float4 number 2 s + h of the row), so that a candidate's bytes are fetched in
 * whole cache lines -- in the tile-packed index each 16-byte piece of a candidate lies in a line of its own (8 x the bytes; amzn-books, 32 x 1 024
 * candidates: 47 -> 24 us; 32 x 10 272: the whole step 3.67 -> 3.34 ms).  Same values in the same order: the logits are bit-identical to rails_mol_score_indexed's.  Shapes and sizes where
 * rails_mol_score_indexed_supported holds; exact-fp32 precision.  One more copy of the index in memory (optional: callers that cannot afford it
 * use rails_mol_score_indexed). */
size_t rails_mol_index_rows_floats(const rails_mol_shape* shape, int64_t n_items);
int rails_mol_index_rows_build(const rails_mol_shape* shape, const float* index, int64_t n_items, float* index_rows, void* stream);
/* cand_counts (optional, may be NULL): row b has cand_counts[b] <= n_cand candidates (rails_candidates_select's counts): only logits[b][0 .. cand_counts[b])
 * are written, the tiles past them are skipped. */
int rails_mol_score_indexed_rows(const rails_mol_shape* shape, const float* gate_pack, const float* query_pack, int32_t batch, const float* index_rows,
                                 int64_t n_items, const int64_t* positions, int64_t n_cand, float* logits, int64_t ld, const int32_t* cand_counts,
                                 void* stream);
int rails_mol_score_indexed(const rails_mol_shape* shape, const float* gate_pack, const float* query_pack, int32_t batch, const float* index,
                            int64_t n_items, const int64_t* positions, int64_t n_cand, float* logits, int64_t ld, void* stream);

/* ---- dot-product (MIPS) scoring ---------------------------------------------------------------
 * Replaces torch.mm(query_embeddings, item_embeddings_t) of MIPSBruteForceTopK.forward
 * (rails/indexing/mips_top_k.py:56-81) and of DotProductSimilarity.forward
 * (rails/similarities/dot_product_similarity_fn.py:48-54).  fp32. */
size_t rails_mips_index_floats(int32_t dim, int64_t n_items);          /* tile-packed copy of the (n, dim) table */
int rails_mips_index_build(const float* items, int64_t n_items, int32_t dim, float* index, void* stream);
size_t rails_mips_query_ws_floats(int32_t dim, int32_t batch);         /* scratch for the packed queries */
int rails_mips_score(const float* queries, int32_t batch, int32_t dim, const float* index, int64_t n_items,
                     float* query_ws, float* logits, int64_t ld, void* stream);

/* Per-row candidates: out[bq * n_cand + x] = <queries[bq], items[bq / r][x]> for queries (n_queries, dim) and items
 * (n_queries / r, n_cand, dim) -- the two bmm branches of DotProductSimilarity.forward
 * (rails/similarities/dot_product_similarity_fn.py:55-68). */
int rails_dot_rowwise(const float* queries, const float* items, int64_t n_queries, int32_t n_cand, int32_t dim, int32_t r,
                      float* out, void* stream);

/* ---- coarse pass of the two-pass approximate top-k ---------------------------------------------
 * Replaces MoLAvgTopK.__init__'s averaged bf16 table (rails/indexing/mol_top_k.py:321-325) and the bf16 `mm`
 * of MoLAvgTopK.forward / topk_ids (mol_top_k.py:351-354, :418-425).  Every value the reference rounds to bf16
 * is rounded to bf16 here; scores come back as fp32 holding bf16 values.  The caller selects the K' best with
 * rails_topk and reranks with rails_mol_index_gather + rails_mol_score_candidates. */
size_t rails_mol_coarse_table_bytes(const rails_mol_shape* shape, int64_t n_items);   /* 2*d bytes per item */
int rails_mol_coarse_build(const rails_mol_shape* shape, const float* index, int64_t n_items, void* table, void* stream);
/* eq: plain (batch, P_Q, d) fp32 from rails_mol_query_prologue's eq_out.  average_queries 0: sum over P_Q
 * (forward), 1: mean over P_Q (topk_ids).  scores[b * ld + x]. */
int rails_mol_coarse_score(const rails_mol_shape* shape, const float* eq, int32_t batch, int32_t average_queries,
                           const void* table, int64_t n_items, float* scores, int64_t ld, const int32_t* run_if, void* stream);

/* Fused coarse scoring + exact top-K' (the same scores as rails_mol_coarse_score followed by rails_topk, without
 * materialising the (batch, n_items) matrix): per-group maxima of a strided sample of the table fix a per-query threshold,
 * one streaming pass collects the items at or above it, and the K' best of those are selected with the position tie rule.
 * Four launches.  out_scores / out_positions: (batch, k_prime), descending.  out_counts[b] = candidates query b collected;
 * the result for query b is exact iff k_prime <= out_counts[b] <= rails_mol_coarse_topk_capacity(batch, n_items, k_prime)
 * (min(24576, max(4096, 8 k_prime)) rounded up to a multiple of 64 for corpora beyond 4 Mi items; 4 k_prime below, where a denser
 * sample leaves ~1.8 k_prime candidates; capacity + 1 is reported when one of the 16 internal sub-lists overflowed) -- otherwise
 * (heavy ties at the threshold) the caller falls back to rails_mol_coarse_score + rails_topk; slots of an under-filled row
 * name position 0 with score -inf.  out_of_range (may be NULL): one int32 the call sets to 1 if some query's count is
 * outside that range and to 0 otherwise -- what rails_range_flag_i32 computes from out_counts, inside the call's own launches
 * (usable as the run_if predicate of the fallback).
 * k_prime <= 4096, batch <= 128 (slice larger batches).  rails_mol_coarse_topk_workspace_bytes returns 0 when the sizes are
 * unsupported. */
size_t rails_mol_coarse_topk_workspace_bytes(const rails_mol_shape* shape, int32_t batch, int64_t n_items, int32_t k_prime);
int32_t rails_mol_coarse_topk_capacity(int32_t batch, int64_t n_items, int32_t k_prime);   /* 0: unsupported sizes (ABI 10) */
int rails_mol_coarse_topk(const rails_mol_shape* shape, const float* eq, int32_t batch, int32_t average_queries,
                          const void* table, int64_t n_items, int32_t k_prime, void* workspace, size_t workspace_bytes,
                          float* out_scores, int64_t* out_positions, int32_t* out_counts, int32_t* out_of_range,
                          void* prefilter, void* stream);
/* Optional int8 pre-filter of rails_mol_coarse_topk's streaming pass (no counterpart in the reference; it changes what the pass
 * READS, not what it returns): a copy of the coarse table as int8 with one scale (256-byte header + d bytes per item).  With it the
 * streaming pass reads the int8 copy, one int8 MFMA per 32 items, against a per-query integer bound that no item reaching the
 * query's threshold can miss (|bf16 dot - scaled int8 dot| <= s |q|_1 / 2 + s_q max|x|_1 / 2 + 3 s s_q d / 4), and scores only the
 * tiles that pass it from the bf16 table: same candidates, same counts, same output, about half the bytes.  prefilter == NULL: the
 * pass reads the bf16 table.  d in {32, 64, 128} (rails_mol_coarse_prefilter_bytes returns 0 otherwise).
 * The header keeps two running uint64 statistics the calls add to: bytes 32..39 = (32-item tile, 32-query tile) blocks that passed the
 * integer bound, bytes 40..47 = blocks tested.  A table whose single scale is set by a few outliers makes most blocks pass (the
 * output stays exact, the pass then reads both copies): a caller that reads a ratio near 1 should pass NULL from then on. */
size_t rails_mol_coarse_prefilter_bytes(const rails_mol_shape* shape, int64_t n_items);
int rails_mol_coarse_prefilter_build(const rails_mol_shape* shape, const void* table, int64_t n_items, void* prefilter, void* stream);

/* ---- per-component candidate generation (MoLNaiveTopK / MoLCombTopK) ------------------------------
 * Replaces the bf16 component table (rails/indexing/mol_top_k.py:61-73, :172-174) and the per-query-group bf16 `mm`
 * (mol_top_k.py:247-251, :498-502).  scores has batch * P_Q * P_X rows, row (b * P_Q + i) * P_X + m, n_items columns;
 * feed it to rails_topk with k = k_per_group and reshape the (rows, k) positions to (batch, P_Q * P_X * k). */
/* The table is ITEM-GROUP-major (ABI 9): table[m][x][:] = bf16(Ex[x, m, :]) for the n_total items of the corpus, so that a scan streams one group's
 * rows back to back.  rails_mol_component_build writes the rows of items [first_item, first_item + n_items) of every group from an index (chunk)
 * of n_items items -- one call with n_total = n_items, first_item = 0 for a whole index. */
size_t rails_mol_component_table_bytes(const rails_mol_shape* shape, int64_t n_items);   /* 2 * P_X * d bytes per item */
int rails_mol_component_build(const rails_mol_shape* shape, const float* index, int64_t n_items, void* table, int64_t n_total, int64_t first_item,
                              void* stream);
int rails_mol_component_score(const rails_mol_shape* shape, const float* eq, int32_t batch, const void* table,
                              int64_t n_items, float* scores, int64_t ld, const int32_t* run_if, void* stream);
/* Fused scoring + exact top-k_group of every (query group, item group) row, without the (rows, n_items) score matrix:
 * same scheme, same outputs contract and same counts check as rails_mol_coarse_topk, over batch * P_Q * P_X rows.
 * out_scores / out_positions: (batch * P_Q * P_X, k_group); out_counts: (batch * P_Q * P_X).
 * batch * P_Q <= 256 query rows per call (128 at d = 128; a zero workspace size says "unsupported": callers slice the batch or take the materialising path).
 * out_of_range (optional, an int32 in device memory, zeroed by the call's first launch): raised when some row's candidate count left
 * [k_group, rails_mol_component_topk_capacity] -- the launch predicate of the caller's redo. */
size_t rails_mol_component_topk_workspace_bytes(const rails_mol_shape* shape, int32_t batch, int64_t n_items, int32_t k_group);
int32_t rails_mol_component_topk_capacity(const rails_mol_shape* shape, int32_t batch, int64_t n_items, int32_t k_group);
int rails_mol_component_topk(const rails_mol_shape* shape, const float* eq, int32_t batch, const void* table, int64_t n_items,
                             int32_t k_group, void* workspace, size_t workspace_bytes, float* out_scores,
                             int64_t* out_positions, int32_t* out_counts, int32_t* out_of_range, void* stream);
/* torch.sort(indices, dim=1) on (rows, n) int64, n <= 16384 (mol_top_k.py:257, :515); in == out allowed. */
int rails_sort_rows_i64(const int64_t* in, int32_t rows, int32_t n, int64_t* out, void* stream);
/* scores[r][j] = fill where sorted_idx[r][j] == sorted_idx[r][j-1] (mol_top_k.py:277-284, :535-542). */
int rails_mask_sorted_duplicates(const int64_t* sorted_idx, float* scores, int64_t ld, int32_t rows, int32_t n, float fill,
                                 void* stream);

/* ---- exact top-k -----------------------------------------------------------------------------
 * Replaces torch.topk(all_logits, dim=1, k, sorted, largest=True) + the id gather
 * (rails/indexing/mol_top_k.py:123-130).  Row b reads scores[b * ld + 0..n).  Ties are broken by
 * position ascending, so results are deterministic and identical for any sharding of the corpus.
 * If `ids` is non-NULL, out_ids[b][j] = ids[ids_row_stride * b + pos] (ids_row_stride 0 = one shared
 * id row, the reference's item_ids.squeeze(0)); else out_ids holds positions.
 * `sorted` == 0 still returns the exact top-k set (in descending order; unordered is a subset of valid
 * answers). */
size_t rails_topk_workspace_bytes(int32_t rows, int64_t n, int32_t k);
int rails_topk(const float* scores, int64_t ld, int32_t rows, int64_t n, int32_t k, int32_t sorted,
               const int64_t* ids, int64_t ids_row_stride, float* out_scores, int64_t* out_ids,
               void* workspace, size_t workspace_bytes, const int32_t* run_if, void* stream);

/* The final_topk of a candidate rerank (rails/indexing/mol_top_k.py:371-382: torch.topk over the candidates' scores,
 * torch.gather of their corpus positions, item_ids lookup) in one launch: row b holds n_cand <= 16384 scores, candidate j of row b
 * is corpus position positions[b * n_cand + j]; out_ids[b][j'] = ids[position] (ids: one shared id row, or NULL for the position
 * itself).  Same order and tie rule (candidate column ascending) as rails_topk on the same rows. */
int rails_topk_candidates(const float* scores, int64_t ld, int32_t rows, int32_t n_cand, int32_t k, const int64_t* positions,
                          const int64_t* ids, float* out_scores, int64_t* out_ids, void* stream);

/* The same selection with the seen-id filter of rails_filter_seen_ids applied to the k' winners inside the launch: what
 * CandidateIndex.get_top_k_outputs keeps of a candidate rerank (indexing/candidate_index.py:149-175 after rails/indexing/mol_top_k.py:260-293 /
 * :518-551, whose modules return ALL their candidates sorted: the first k unseen ones of that list are the first k unseen ones of its
 * top k + width, since a masked duplicate never outranks a scored candidate and at most `width` of the scored ones are seen) ->
 * (out_ids, out_scores) of k per row, the same bits as rails_topk_candidates(k') followed by rails_filter_seen_ids.
 * Sizes: 1024 < n_cand <= 8192, k <= k' <= 512, k' <= n_cand, width <= 256; RAILS_ENOTSUP otherwise (the caller composes the two calls). */
int rails_topk_candidates_filtered(const float* scores, int64_t ld, int32_t rows, int32_t n_cand, int32_t k_prime, const int64_t* positions,
                                   const int64_t* ids, const int64_t* invalid_ids, int32_t width, int32_t k, int64_t* out_ids,
                                   float* out_scores, void* stream);

/* The same result from candidates in ANY order, duplicates included (the union a Naive / Comb rerank collects, before the integer sort of
 * rails/indexing/mol_top_k.py:262 / :520 that makes duplicates neighbours): row b holds the scores of its n_cand candidate positions; of every
 * position the first copy counts, ranked by (score desc, position asc) -- the order of the sorted form -- then the top k' with the seen-id filter as
 * above.  Two launches (an LDS hash set of the row's positions writes the keys; the selection of rails_topk_candidates_filtered on them), no
 * sort.  *out_of_range (int32, device-visible, zeroed by the caller; may be pinned host memory) is set to 1 when a row holds fewer than k'
 * distinct positions: the outputs are then undefined and the caller takes the sorted form (rails_sort_rows_i64 -> score ->
 * rails_mask_sorted_duplicates -> rails_topk_candidates_filtered), which ranks masked duplicates as the reference does.  positions < 2^32 - 1.
 * Sizes as rails_topk_candidates_filtered; workspace: rails_rerank_workspace_bytes(rows, n_cand), 256-byte aligned. */
size_t rails_rerank_workspace_bytes(int32_t rows, int32_t n_cand);
int rails_rerank_topk_filtered(const float* scores, int64_t ld, int32_t rows, int32_t n_cand, int32_t k_prime, const int64_t* positions,
                               const int64_t* ids, const int64_t* invalid_ids, int32_t width, int32_t k, void* workspace, size_t workspace_bytes,
                               int64_t* out_ids, float* out_scores, int32_t* out_of_range, void* stream);

/* CandidateIndex.get_top_k_outputs' selection in ONE chain (reference indexing/candidate_index.py:149-175 after
 * rails/indexing/mol_top_k.py:123-130): exact top-k' of every row with the id map, then the seen-id filter of
 * rails_filter_seen_ids applied to the k' winners INSIDE the final selection launch -> (out_ids, out_scores) of k per row, the
 * same bits as rails_topk followed by rails_filter_seen_ids.  Sizes: rails_topk_filter_fusable(n, k', width, k) != 0
 * (n > 1024, k' <= 512, width <= 256); RAILS_ENOTSUP otherwise -- call the two entry points then. */
int rails_topk_filter_fusable(int64_t n, int32_t k_prime, int32_t width, int32_t k);
int rails_topk_filtered(const float* scores, int64_t ld, int32_t rows, int64_t n, int32_t k_prime, const int64_t* ids, int64_t ids_row_stride,
                        const int64_t* invalid_ids, int32_t width, int32_t k, int64_t* out_ids, float* out_scores, void* workspace,
                        size_t workspace_bytes, const int32_t* run_if, void* stream);

/* ---- item-sharded top-k (no counterpart in the reference, whose eval is single-GPU: eval_from_checkpoint.py:554-555) ----
 * Each rank turns its local top-k into one message row of 2k int64 (k score words: fp32 bits in the low half | k ids;
 * rows with k_local < k are padded with (-inf, -1)); the caller all-gathers the messages in rank order; every rank
 * merges them.  Contiguous item-id shards + rank order keep the tie rule, so the result equals the unsharded top-k. */
int rails_pack_candidates(const float* scores, const int64_t* ids, int32_t rows, int32_t k_local, int32_t k, int64_t* msg,
                          void* stream);
/* gathered: (n_ranks, rows, 2k) int64.  n_ranks * k <= 16384.  k_out <= n_ranks * k. */
int rails_merge_candidates(const int64_t* gathered, int32_t n_ranks, int32_t rows, int32_t k, int32_t k_out,
                           float* out_scores, int64_t* out_ids, void* stream);
/* rails_merge_candidates followed by rails_filter_seen_ids in ONE launch (the sharded counterpart of rails_topk_filtered): the k_prime
 * merged winners of every row go through the seen-id filter inside the merge kernel and k_out (ids, scores) per row are written --
 * the same bits as the two calls.  k_prime <= 512, width <= 256 (RAILS_ENOTSUP otherwise). */
int rails_merge_candidates_filtered(const int64_t* gathered, int32_t n_ranks, int32_t rows, int32_t k, int32_t k_prime,
                                    const int64_t* invalid_ids, int32_t width, int32_t k_out, int64_t* out_ids, float* out_scores,
                                    void* stream);

/* rails_merge_candidates[_filtered] over the 2k + 2 wide messages of rails_candidates_finish's sharded form ([k score words | k ids | m | err] per
 * row and rank) WITH the global verdict of the item-sharded proved top-k in the same launch: row b is proved iff no rank reported a bad row
 * (err = inf), every guard value is within guard_limit, and  merged k_out-th score - max over ranks of m > eps  (eps as in rails_candidates_finish).
 * The last workgroup folds the rows into `state` (rails_rescore_verdict's layout) and mirrors it into state_host (optional, pinned host memory;
 * state_host[5], the call counter, is written last).  call_ws: 8 + 4 rows uint32 in device memory (16-byte aligned), zeroed once by the caller (every call leaves its first word, the arrival counter, zero).
 * Every rank computes the same verdict from the same gathered bytes, so the ranks agree on a redo without another exchange.
 * invalid_ids == NULL: (out_scores, out_ids) are (rows, k_out); else the seen-id filter runs inside the launch and they are (rows, f_k).
 * No counterpart in the reference (single-GPU eval: eval_from_checkpoint.py:554-555). */
int rails_merge_candidates_verdict(const int64_t* gathered, int32_t n_ranks, int32_t rows, int32_t k, int32_t k_out, float default_eps, float safety,
                                   const float* guard_values, int32_t guard_per_row, float guard_limit, float* state, float* state_host,
                                   void* call_ws, const int64_t* invalid_ids, int32_t width, int32_t f_k, int64_t* out_ids, float* out_scores,
                                   void* stream);

/* Finish of a speculate-then-verify brute-force top-k (precision "f16x3-exact"; no counterpart in the reference, whose
 * MoLBruteForceTopK scores everything in one precision, mol_top_k.py:84-130).  Per row: n_cand entries with their exact fp32
 * logits (row stride ld) and corpus positions (< n_items <= 2^32).  The first n_ranked are the candidates: the top n_ranked items
 * by the approximate logits approx_scores (rows, n_ranked), same order.  The rest are probes: arbitrary items whose approximate
 * logit is read from the dense matrix approx_dense (row stride ld_dense) at their position.
 * Writes the candidates' top k by (exact score desc, position asc) -- the dense path's total order -- as (scores, ids[position]
 * or the position when ids is NULL), and row_ok[row] = 1 iff  k-th exact score > min candidate approx + margin_eps  and
 * |exact - approx| <= check_eps on every candidate and probe  (then no item outside the candidates can belong to the row's
 * top k, given |approx - exact| <= margin_eps everywhere; the probes watch that bound outside the candidates).  n_cand <= 16384.
 * row_stats (rows x 2 floats, optional): [largest |exact - approx| over the row's candidates and probes (inf for a NaN), k-th exact
 * score - min candidate approx], for callers that calibrate the bound from what they observe; row_ok or row_stats may be NULL.
 * one_sided != 0: approx_scores are UPPER BOUNDS of the exact scores (rails_mol_score_dense_upper): the monitored error is
 * max(0, exact - approx) -- zero while the bound holds -- and margin_eps = 0 is the proof (every item outside the candidates has
 * exact <= approx <= min candidate approx < k-th exact score). */
/* *flag |= 1 if any of the n int32 values lies outside [lo, hi]: the validity check of the fused scans' candidate counts
 * (rails_mol_coarse_topk / rails_mol_component_topk) on the device, feeding the launch predicate (run_if) of their materialising
 * redo (rails_mol_coarse_score / rails_mol_component_score + rails_topk).  The caller zeroes *flag. */
int rails_range_flag_i32(const int32_t* values, int32_t n, int32_t lo, int32_t hi, int32_t* flag, void* stream);

/* Verdict of a speculate-then-verify call, on the device: from the row_stats of rails_rescore_select (rows x 2 floats) and the
 * caller's calibration state (8 floats in device memory, zero-initialised once):
 *   state[0]  largest |first pass - fp32| ever observed (updated here; never decreases)
 *   state[1]  REDO flag, an int32 (written here): 1 iff some row's margin <= eps or a NaN was seen, eps = max(default_eps, safety * state[0])
 *   state[2]  eps   state[3] / state[4]  this call's largest error / smallest margin   state[5] / state[6]  calls / redone calls so far
 *   state[7]  largest |guard value| ever observed
 * guard_values (optional, may be NULL): guard_count floats whose magnitudes must not exceed guard_limit for the caller's bound on
 * |first pass - fp32| to hold -- the proved exact top-k passes the batch's prescaled query-gate rows gq' (the one data-dependent magnitude of
 * its a-priori bound, rails_amd/f16x3_bound.py); a larger value or a NaN raises REDO like a failed margin.
 * No counterpart in the reference (rails/indexing/mol_top_k.py:99-130 scores every item in one precision). */
int rails_rescore_verdict(const float* row_stats, int32_t rows, float default_eps, float safety, const float* guard_values, int64_t guard_count,
                          float guard_limit, float* state, void* stream);

/* row_stats for a GLOBAL verdict of the item-sharded proved top-k (rails_amd/sharded.py; no counterpart in the reference, whose eval is
 * single-GPU: eval_from_checkpoint.py:554-555): row_stats[row] = [err_max[0], kth_scores[row * ld + col] - m_max[row]], where kth_scores
 * holds the merged top-k' (col = k' - 1), m_max the all-reduced maximum over ranks of the best first-pass score each rank left outside its
 * candidates, err_max the all-reduced largest |first pass - fp32| seen on any candidate.  Feed it to rails_rescore_verdict. */
int rails_margin_stats(const float* kth_scores, int64_t ld, int32_t col, const float* m_max, const float* err_max, int32_t rows, float* row_stats, void* stream);

int rails_rescore_select(const float* exact_scores, int64_t ld, const float* approx_scores, const float* approx_dense, int64_t ld_dense,
                         const int64_t* positions, const int64_t* ids, int64_t n_items, int32_t rows, int32_t n_ranked, int32_t n_cand,
                         int32_t k, float margin_eps, float check_eps, int32_t one_sided, float* out_scores, int64_t* out_ids, int32_t* row_ok,
                         float* row_stats, void* stream);

/* ---- candidates of the proved exact top-k: threshold selection and fused finish (round 6) -------------------------------------------
 * No counterpart in the reference (its MoLBruteForceTopK scores every item in one precision and calls torch.topk,
 * rails/indexing/mol_top_k.py:99-130; the seen-id filter fused below is indexing/candidate_index.py:149-175).  The speculate-then-verify
 * flow needs, per row, a candidate SET C and a value m such that every item outside C has a first-pass score <= m -- not the exact kc
 * best.  rails_candidates_select takes a threshold: with b(s) = the bin of score s among 4 096 equal bins of [lo, hi] (scores outside are
 * clamped; b is monotone), C = {x : b(s_x) >= b_t} for the LOWEST bin b_t that leaves at most `cap` candidates, and m = min over C of s.
 * One histogram launch + one compaction launch (rows of <= 65 536 scores: one launch), against the ten launches of an exact rails_topk
 * with k = kc.
 *   workspace: rails_candidates_workspace_bytes(rows) bytes, ZEROED ONCE by the caller; every select + finish pair leaves it zeroed.  Its
 *              first `rows` int32 are the rows' candidate counts between the two calls (rails_mol_score_indexed_rows' cand_counts).
 *   out_positions (rows, cand_ld) int64 / out_approx (rows, cand_ld) float: the candidates (in no particular order) and their first-pass
 *              scores; cand_ld >= cap; cap <= 16384; n < 2^32.  A NaN score raises the row's flag (the finish then fails the row).
 * rails_candidates_finish, one workgroup per row: the row's candidates sorted by (exact fp32 score desc, position asc) -- the dense
 * path's total order -- and the best k written as (out_scores, out_ids = ids[position] or the position when ids is NULL); the row FAILS
 * unless it has >= k candidates, no NaN, every guard value within guard_limit, and  k-th exact score - m > eps  with eps = max(default_eps,
 * safety * largest |exact - approx| seen so far (state[0]) or on this row)  -- then no item outside the candidates can belong to the row's
 * top k given |approx - exact| <= eps everywhere (one_sided != 0: approx are UPPER bounds of the exact scores, the monitored error is
 * max(0, exact - approx), default_eps = 0 is the proof).  A row whose candidates are all n_items items passes on that alone.
 * The last workgroup folds the rows into `state` (rails_rescore_verdict's layout; state[1] = the REDO flag, the launch predicate of the
 * caller's fallback) and mirrors it into state_host (optional; PINNED HOST memory mapped into the device: the words first, state_host[5],
 * the call counter the host polls, last -- no copy launch).  guard: rows x guard_per_row floats (the batch's prescaled query-gate rows).
 * invalid_ids != NULL: the seen-id filter of the candidate index over the k winners inside the same launch -> f_out_ids / f_out_scores
 * (rows, f_k); k <= 512, width <= 256.
 * msg != NULL (item-sharded form, rails_amd/sharded.py): instead of outputs and verdict, row b of msg (rows, 2k + 2) int64 receives
 * [k score words | k ids | m | largest error (inf: the row is bad)] -- short rows padded with (-inf, -1) -- for ONE all-gather; the verdict
 * is rails_merge_candidates_verdict's, after the merge. */
size_t rails_candidates_workspace_bytes(int32_t rows);
int rails_candidates_select(const float* scores, int64_t ld, int32_t rows, int64_t n, int32_t cap, float lo, float hi, void* workspace,
                            int64_t* out_positions, float* out_approx, int64_t cand_ld, void* stream);
int rails_candidates_finish(const float* exact_scores, int64_t ld, const float* approx, const int64_t* positions, int64_t cand_ld, int32_t cap,
                            void* workspace, const int64_t* ids, int64_t n_items, int32_t rows, int32_t k, float default_eps, float safety,
                            int32_t one_sided, const float* guard_values, int32_t guard_per_row, float guard_limit, float* out_scores,
                            int64_t* out_ids, const int64_t* invalid_ids, int32_t width, int32_t f_k, int64_t* f_out_ids, float* f_out_scores,
                            float* state, float* state_host, int64_t* msg, void* stream);

/* ---- arithmetic-model probes (test infrastructure of the proved exact top-k; no counterpart in the reference) -------------------------
 * The a-priori bound on |first pass - fp32 logit| (rails_amd/f16x3_bound.py) models the two matrix instructions and the two
 * transcendentals the scoring kernels are made of; these entry points run ONE such instruction per element on caller-supplied operands so
 * that tests can measure the model on the part (tests/test_gpu_parity.py::test_f16_mfma_accumulation_model and neighbours).
 *   rails_mfma_probe_f16: n independent D = C + A B with v_mfma_f32_32x32x16_f16; A (32 x 16) and B (16 x 32) row-major f16 bit patterns,
 *                         C and D (32 x 32) row-major fp32, n of each back to back.
 *   rails_mfma_probe_f32: the same with v_mfma_f32_32x32x2_f32; A (32 x 2), B (2 x 32) fp32.
 *   rails_scalar_probe_f32: out[0..n) = v_exp_f32(x), out[n..2n) = v_rcp_f32(x), out[2n..3n) = x / (1 + 2^x) as the kernels compute it. */
int rails_mfma_probe_f16(const uint16_t* a, const uint16_t* b, const float* c, float* d, int64_t n, void* stream);
int rails_mfma_probe_f32(const float* a, const float* b, const float* c, float* d, int64_t n, void* stream);
int rails_scalar_probe_f32(const float* x, int64_t n, float* out, void* stream);

/* ---- synthetic corpora (measurement and test infrastructure; no counterpart in the reference, whose item tables are trained) ----
 * out[(i - first_item) * dim + c] for items first_item <= i < first_item + n_items: a counter-based hash of (seed, i, c) --
 *   h = splitmix64((i * dim + c) ^ (seed * 0xD1B54A32D192ED03));  value = float(sum of h's four 16-bit lanes - 131070) * scale
 * -- so any shard or sub-range of a corpus is reproducible without materialising the table, on the device and (bit for bit) on the
 * host (oracle/mol_oracle.py hash_item_table, scale = float32(sigma * sqrt(3) / 65536): Irwin-Hall(4), std sigma). */
int rails_hash_item_table(uint64_t seed, int64_t first_item, int64_t n_items, int32_t dim, float scale, float* out, void* stream);

/* ---- seen-id filter --------------------------------------------------------------------------
 * Replaces the row-wise masking of CandidateIndex.get_top_k_outputs (indexing/candidate_index.py:154-178):
 * keep the first k ids of each row of (rows, k_prime) that do not occur in invalid_ids (rows, width),
 * back-filling from the filtered-out ones (in order) when fewer than k survive. */
int rails_filter_seen_ids(const int64_t* top_ids, const float* top_scores, int32_t rows, int32_t k_prime,
                          const int64_t* invalid_ids, int32_t width, int32_t k, int64_t* out_ids,
                          float* out_scores, void* stream);

/* ---- HSTU query encoder, eval path -----------------------------------
 * The step upstream of the retrieval path: modeling/sequential/hstu.py (HSTU.encode with no cache / delta path), without
 * fbgemm-gpu.  All tensors are fp32 and padded: (batch, seq_len, ...) with `lengths[b]` valid positions per row; rows at
 * positions >= lengths[b] are held at zero, which is what the reference's jagged layout amounts to.
 * rails_amd/hstu.py chains these per layer; a non-Python host would do the same. */
/* x = [ids != 0 and n < lengths[b]] * (embeddings * scale + pos_emb[n])      input_features_preprocessors.py:75-92 */
int rails_hstu_preprocess(const float* embeddings, const int64_t* ids, const int64_t* lengths, const float* pos_emb,
                          int32_t batch, int32_t seq_len, int32_t dim, float scale, float* out, void* stream);
/* out[r] = F.layer_norm(x[r], eps) (no affine) [* mul[r]]; ld* are row strides in floats.  hstu.py:268-276, :419-424 */
int rails_rows_layer_norm(const float* x, int64_t ldx, int64_t rows, int32_t dim, float eps, const float* mul, int64_t ldm,
                          float* out, int64_t ldo, void* stream);
/* C = act(A W + bias) + residual.  w_is_nk 0: W is (K, N) row-major (the `_uvqk` parameter); 1: W is (N, K), a torch Linear
 * weight (`_o.weight`).  act 0 none / 1 silu.  With lengths != NULL rows r = b * seq_len + n, n >= lengths[b], are written
 * as zeros.  hstu.py:374-378 (torch.mm + silu), :426-434 (the output Linear + residual). */
int rails_gemm_f32(const float* a, int64_t lda, const float* w, int32_t w_is_nk, const float* bias, const float* residual,
                   int64_t ldr, int64_t m, int32_t n, int32_t k, int32_t act, const int64_t* lengths, int32_t seq_len, float* c,
                   int64_t ldc, void* stream);
/* MoLGatingFn.forward's combination + SoftmaxDropoutCombiner.forward as a stand-alone unit (reference
 * rails/similarities/mol/similarity_fn.py:148-201 and :31-46/:66-96, eval mode), for callers that use the modules on their own --
 * inside the scoring path all of this is fused into the scoring kernels.  Row r = b * items_per_query + x:
 *   g = query_part[b] * item_part[x or r] + pair_part[r]   (RAILS_COMBINE_GLU_SILU: w = g * sigmoid(g))
 *   g = the sum of the parts that are not NULL             (RAILS_COMBINE_NONE: w = g)
 *   pi = softmax(w) [/ clamp(sum pi, eps) if renormalise: the combiner does that whenever its dropout rate is > 0]
 *   out[r] = sum_l pi[l] * logits[r][l];  probs_out (rows, num_logits), optional, receives pi.
 * The combiner alone: pair_part = the gating weights, the other parts NULL, RAILS_COMBINE_NONE.  num_logits <= 1024. */
int rails_mol_gate_combine(const float* logits, int64_t ld_logits, const float* pair_part, int64_t ld_pair, const float* query_part,
                           const float* item_part, int64_t rows, int32_t items_per_query, int32_t num_logits, int32_t item_part_per_row,
                           int32_t combination, int32_t renormalise, float eps, float* out, float* probs_out, void* stream);

/* out = act(l) * g with [l | g] = x W + b: GeGLU (act = erf gelu) / SwiGLU (act = silu) as stand-alone layers
 * (reference rails/similarities/layers.py:19-74; `kind` RAILS_GEGLU | RAILS_SWIGLU).  w is the `_w` parameter, (in_features,
 * 2 * out_features) row-major; b the `_b` parameter (2 * out_features) or NULL; scratch holds rows * 2 * out_features floats
 * (the pre-activations); out is (rows, out_features) dense.  Inside the MoL path the same unit is fused into the query prologue
 * and the index build; this entry point serves callers that use the layer on its own. */
int rails_glu_f32(const float* x, int64_t ldx, const float* w, const float* b, int64_t rows, int32_t in_features, int32_t out_features,
                  int32_t kind, float* scratch, float* out, void* stream);
/* buckets[b][j][i] = #{t : thresholds[t] <= |ts[b, min(i+1, seq_len-1)] - ts[b, j]|} (uint8, key-major): the time bucket of
 * (query i, key j) of RelativeBucketedTimeAndPositionBasedBias (hstu.py:107-138); thresholds (num_buckets int64, ascending)
 * are the bucket edges of floor(log(max(|dt|, 1)) / 0.301) evaluated in float32 exactly as torch does.  Once per batch:
 * it depends on neither the layer nor the head. */
int rails_hstu_time_buckets(const int64_t* timestamps, int32_t batch, int32_t seq_len, const int64_t* thresholds,
                            int32_t num_buckets, uint8_t* out, void* stream);
/* uvqk: (batch * seq_len, ld) rows [u | v | q | k] (u, v: heads * dv wide; q, k: heads * dqk wide), the SiLU'd GEMM output.
 * out[b, i, h, :] = sum_{j <= i} silu(q_i . k_j + pos_w[seq_len - 1 + j - i] + ts_w[buckets[b, j, i]]) / seq_len * v_j ;
 * no bias at all when buckets == NULL (the reference without timestamps).  hstu.py:144-213.  dv <= 32, dqk <= 32. */
int rails_hstu_attention(const float* uvqk, int64_t ld, int32_t batch, int32_t seq_len, int32_t heads, int32_t dqk, int32_t dv,
                         const int64_t* lengths, const uint8_t* buckets, const float* ts_w, const float* pos_w,
                         int32_t num_buckets, float* out, void* stream);
/* The whole eval-path encoder in ONE launch for short sequences (seq_len <= 64 and the LDS bound checked by
 * rails_hstu_fused_supported): one workgroup per sequence keeps the residual stream, the uvqk activations and the bias tables
 * in LDS and runs all blocks back to back; writes the postprocessed embedding at position lengths[b] - 1 (HSTU.encode,
 * hstu.py:741-803).  `layers`: DEVICE array of n_blocks rails_hstu_layer (device pointers to each block's parameters; ts_w /
 * pos_w are only read when buckets != NULL).  buckets as for rails_hstu_attention.  RAILS_ENOTSUP for other geometries: the
 * caller then chains the per-layer entry points above. */
typedef struct rails_hstu_layer {
  const float* uvqk;   /* (dim, 2 * heads * (dv + dqk)) */
  const float* o_w;    /* (dim, heads * dv), a torch Linear weight */
  const float* o_b;    /* (dim) */
  const float* ts_w;   /* (num_buckets + 1) */
  const float* pos_w;  /* (2 * seq_len - 1) */
} rails_hstu_layer;
int rails_hstu_fused_supported(int32_t seq_len, int32_t dim, int32_t heads, int32_t dqk, int32_t dv, int32_t num_buckets);
int rails_hstu_encode_fused(const float* embeddings, const int64_t* ids, const int64_t* lengths, const uint8_t* buckets,
                            const float* pos_emb, const rails_hstu_layer* layers, int32_t n_blocks, int32_t batch, int32_t seq_len,
                            int32_t dim, int32_t heads, int32_t dqk, int32_t dv, int32_t num_buckets, int32_t postproc_mode, float eps,
                            float* out, void* stream);
/* out[r] = normalise(x[row_index ? row_index[r] : r]); mode 0 LayerNorm (no affine), 1 x / max(||x||, eps).
 * output_postprocessors.py:38-85 + get_current_embeddings (modeling/sequential/utils.py:74-90). */
int rails_rows_normalize(const float* x, int64_t ldx, const int64_t* row_index, int64_t rows, int32_t dim, int32_t mode, float eps,
                         float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAILS_AMD_H_ */
