// Internal declarations shared by the HIP translation units behind include/rails_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/rails_amd.h"

namespace mol {

using Shape = rails_mol_shape;
using Weights = rails_mol_weights;

enum : int { kOk = RAILS_OK, kErrInvalid = RAILS_EINVAL, kErrUnsupported = RAILS_ENOTSUP,
             kErrNoMem = RAILS_ENOMEM, kErrLaunch = RAILS_ELAUNCH };

constexpr int kScoreThreads = 512;  // 8 waves: two per SIMD, so one wave's VALU phases (silu, softmax)
constexpr int kScoreWaves = 8;      // run under its partner's MFMAs

inline int num_logits(const Shape& s) { return s.query_dot_product_groups * s.item_dot_product_groups; }
inline int queries_per_group(const Shape& s) { return 32 / s.query_dot_product_groups; }
inline int64_t num_tiles(int64_t n_items) { return (n_items + 31) / 32; }
inline int64_t tile_floats(const Shape& s) {
  return 32LL * (s.item_dot_product_groups * s.dot_product_dimension + num_logits(s));
}

struct ScoreArgs {
  const float* wpack;   // packed pair-gate weights (fragment order), Geo::kWpackFloats
  const float* eqfrag;  // [n_groups][32 * d]
  const float* gqfrag;  // [B][L] as [hi][e]
  const float* ipack;   // item tiles
  float* logits;        // [B][ld]
  int64_t ld;
  int64_t n_items;      // shared corpus: items in the index; candidates: items per row
  int64_t n_tiles;      // shared corpus: tiles in the index; candidates: tiles per row
  int B;
  int n_groups;
  int per_row;          // 1: tile t of row b lives at ipack tile (b * n_tiles + t)
  float temperature;
  float rcp_temperature;
  int combine_none;     // 1: gating_combination "none": w = gq + gi + gqi (absent parts are stored as zeros), no silu
  int single;           // 1 (with split): precision F16X1 -- the one-product kernels, which ignore the lo fragments
  int split;            // 1: precision mode f16x3 -- gate pack, query pack and item index hold f16 hi/lo fragments (mol_layout.h)
  const int32_t* run_if;   // device flag (the entry point's run_if argument): the launch is a no-op when *run_if == 0; NULL = unconditional
  // per-row candidates addressed IN the shared index (rails_mol_score_indexed): candidate j of row b is item cand_pos[b * n_items + j]
  // of an index of index_items items; NULL = per-row candidates come as their own gathered tiles (rails_mol_score_candidates)
  const int64_t* cand_pos;
  int64_t index_items;
  const float* irows;             // with cand_pos: the ROW-MAJOR copy of the index (rails_mol_index_rows_build) the candidates are read from instead of ipack
  const int32_t* cand_count;      // with irows (optional): row b has cand_count[b] <= n_items candidates -- tiles beyond them are skipped (rails_candidates_select's counts)
  int dry_run;                    // 1: validate the dispatch (shape, shell) without launching
  // rails_mol_score_dense_upper: logits[b][x] = s + (ub2 c + ub1) c + ub0, c = max_l |cl_l| of the pair (mol_score_wsplit.h UPPER)
  int upper;
  float ub2, ub1, ub0;
};

int pack_gate_weights_split(const Shape& s, const Weights& w, float* wpack, hipStream_t stream);
inline bool is_split(const Shape& s) { return s.precision == RAILS_PRECISION_F16X3 || s.precision == RAILS_PRECISION_F16X1; }

void set_error(const char* fmt, ...);
// Launch predicate (the run_if argument of rails_mol_score_dense / rails_topk / ...): the kernels of a call return at once unless
// the device flag is non-zero -- a device-side conditional (the verified modes' fallback) without a host round trip.
#define MOL_RUN_IF(flag) do { if ((flag) != nullptr && *(flag) == 0) return; } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute and the Python layer drives several devices from
// one process: remember the opt-in per (call site, device).  Two threads racing on the first call both set it (idempotent).
struct DynLdsOnce { std::atomic<unsigned long long> mask{0}; };
inline int ensure_dyn_lds(DynLdsOnce& once, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return kErrLaunch;
  const unsigned long long bit = 1ull << (dev & 63);
  if (once.mask.load(std::memory_order_acquire) & bit) return kOk;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return kErrLaunch;
  once.mask.fetch_or(bit, std::memory_order_release);
  return kOk;
}

// Shapes of the small-unit kernel (mol_score_small.h: v_mfma_f32_16x16x4_f32, 2 queries x 16 items per wave, exact fp32): the three
// real-dataset shapes.  For them the gate pack carries a second copy of the pair-gate weights in that kernel's fragment order.
inline bool score_small_shape(const Shape& s) {
  if (s.precision != RAILS_PRECISION_FP32 || s.query_dot_product_groups != 8 || s.gating_qi_hidden_dim != 128) return false;
  const int px = s.item_dot_product_groups, dd = s.dot_product_dimension;
  return (px == 4 && dd == 64) || (px == 4 && dd == 128) || (px == 8 && dd == 32);
}
inline size_t gate_pack32_floats(const Shape& s) {
  const size_t H = s.gating_qi_hidden_dim > 0 ? (size_t)s.gating_qi_hidden_dim : 0, L = (size_t)num_logits(s);
  return H > 0 ? 2 * H * L + H + L : L * L + L;   // no hidden layer: one (L, L) matrix + bias
}
int score_launch_small(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream);
int score_launch(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream);
int score_launch_f16(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream);
int score_launch_f16x1(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream);
bool score_extra_shape(const Shape& s);   // mol_score_extra_shapes.h
int score_launch_extra(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream);
int score_launch_f16_extra(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream);
int score_launch_f16x1_extra(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream);
bool score_supported(const Shape& s);

int pack_gate_weights(const Shape& s, const Weights& w, float* wpack, hipStream_t stream);
int index_build(const Shape& s, const Weights& w, const float* items, int64_t n, float* ipack, hipStream_t stream);
int index_unpack(const Shape& s, const float* ipack, int64_t n, float* ex, float* gi, hipStream_t stream);
int index_rows_build(const Shape& s, const float* ipack, int64_t n, float* rows, hipStream_t stream);
// fp32 Ex fragments of a freshly built index -> f16 hi/lo fragments, in place (precision f16x3)
int index_split_inplace(const Shape& s, float* ipack, int64_t n, hipStream_t stream);
int index_gather(const Shape& s, const float* ipack, int64_t n, const int64_t* idx, int64_t rows, int64_t n_cand,
                 float* out, hipStream_t stream);
size_t query_scratch_floats(const Shape& s, int B);
int query_prologue(const Shape& s, const Weights& w, const float* q, const int64_t* user_ids, int B, float* qpack,
                   float* eq_out, float* gq_out, hipStream_t stream, float* qpack_other = nullptr);

int coarse_build(const Shape& s, const float* ipack, int64_t n, void* table, hipStream_t stream);
size_t coarse_topk_workspace_bytes(const Shape& s, int B, int64_t n, int k_prime);
int coarse_topk_capacity(int B, int64_t n, int k_prime);
int coarse_topk(const Shape& s, const float* eq, int B, int avg, const void* table, int64_t n, int k_prime, void* ws,
                size_t ws_bytes, float* out_scores, int64_t* out_pos, int32_t* out_counts, int32_t* out_flag, void* prefilter, int n_cu,
                hipStream_t stream);
size_t coarse_prefilter_bytes(const Shape& s, int64_t n);
int coarse_prefilter_build(const Shape& s, const void* table, int64_t n, void* prefilter, hipStream_t stream);
size_t component_topk_workspace_bytes(const Shape& s, int B, int64_t n, int k_group);
int component_topk(const Shape& s, const float* eq, int B, const void* table, int64_t n, int k_group, void* ws, size_t ws_bytes,
                   float* out_scores, int64_t* out_pos, int32_t* out_counts, int32_t* out_flag, int n_cu, hipStream_t stream);
int component_topk_capacity(const Shape& s, int B, int64_t n, int k_group);
// ---- HSTU query encoder, eval path (hstu.hip) ----
int hstu_preprocess(const float* emb, const int64_t* ids, const int64_t* lengths, const float* pos_emb, int B, int N, int D,
                    float scale, float* out, hipStream_t stream);
int rows_layer_norm(const float* x, int64_t ldx, int64_t rows, int D, float eps, const float* mul, int64_t ldm, float* out, int64_t ldo,
                    hipStream_t stream);
int rows_normalize(const float* x, int64_t ldx, const int64_t* row_index, int64_t rows, int D, int mode, float eps, float* out,
                   hipStream_t stream);
int gemm_f32(const float* A, int64_t lda, const float* W, int w_is_nk, const float* bias, const float* residual, int64_t ldr, int64_t M,
             int N, int K, int act, const int64_t* lengths, int seq_len, float* C, int64_t ldc, hipStream_t stream);
int gate_combine(const float* y, int64_t ldy, const float* gqi, int64_t ldq, const float* gq, const float* gi, int64_t rows, int X, int L,
                 int gi_per_row, int glu_silu, int renorm, float eps, float* out, float* pi_out, hipStream_t stream);
int glu_gate(const float* t, int64_t ldt, int64_t rows, int F, int kind, float* out, hipStream_t stream);
int hstu_time_buckets(const int64_t* timestamps, int B, int N, const int64_t* thresholds, int num_buckets, unsigned char* out,
                      hipStream_t stream);
int hstu_attention(const float* uvqk, int64_t ld, int B, int N, int H, int dqk, int dv, const int64_t* lengths,
                   const unsigned char* buckets, const float* ts_w, const float* pos_w, int num_buckets, float* out,
                   hipStream_t stream);
bool hstu_fused_supported(int N, int D, int H, int dqk, int dv, int num_buckets);
int hstu_encode_fused(const float* emb, const int64_t* ids, const int64_t* lengths, const unsigned char* buckets, const float* pos_emb,
                      const void* layers, int n_blocks, int B, int N, int D, int H, int dqk, int dv, int num_buckets, int mode,
                      float eps, float* out, hipStream_t stream);
int select_keys(const unsigned long long* keys, int rows, int keys_per_row, int k, float* out_scores, int64_t* out_pos,
                hipStream_t stream);
int bf16_rows_kth(const unsigned short* rows16, int64_t ld, int n_rows, int n, int r, float* thr, hipStream_t stream);
int select_sublists(const unsigned long long* keys, const unsigned int* counts, int rows, int cap, int n_sub, int k, float* out_scores,
                    int64_t* out_pos, int32_t* out_counts, int32_t* out_flag, hipStream_t stream);
int coarse_score(const Shape& s, const float* eq, int B, int avg, const void* table, int64_t n, float* scores, int64_t ld,
                 hipStream_t stream, const int32_t* run_if = nullptr);

int component_build(const Shape& s, const float* ipack, int64_t n, void* table, int64_t n_total, int64_t first, hipStream_t stream);
int component_score(const Shape& s, const float* eq, int B, const void* table, int64_t n, float* scores, int64_t ld,
                    hipStream_t stream, const int32_t* run_if = nullptr);
int sort_rows_i64(const int64_t* in, int rows, int n, int64_t* out, hipStream_t stream);
int mask_sorted_duplicates(const int64_t* idx, float* scores, int64_t ld, int rows, int n, float fill, hipStream_t stream);
size_t rerank_workspace_bytes(int rows, int n_cand);
int rerank_topk_filtered(const float* scores, int64_t ld, int rows, int n_cand, int k_prime, const int64_t* positions, const int64_t* ids,
                         const int64_t* invalid, int width, int k, void* ws, size_t ws_bytes, int64_t* out_ids, float* out_scores, int32_t* flag,
                         hipStream_t stream);

int hash_item_table(unsigned long long seed, int64_t first_item, int64_t n_items, int dim, float scale, float* out, hipStream_t stream);
int mips_pack_items(const float* items, int64_t n, int D, float* out, hipStream_t stream);
int mips_score(const float* q, int B, int D, const float* ifrag, int64_t n, float* qfrag_ws, float* logits, int64_t ld,
               int n_cu, hipStream_t stream);

int dot_rowwise(const float* q, const float* items, int64_t Bq, int X, int D, int r, float* out, hipStream_t stream);

size_t topk_workspace_bytes(int rows, int64_t n, int k);
// f_invalid != NULL: the seen-id filter of filter_seen fused into the final selection launch (out_* then hold f_k per row)
bool topk_can_fuse_filter(int64_t n, int k, int width, int k_out);
int topk(const float* scores, int64_t ld, int rows, int64_t n, int k, const int64_t* ids, int64_t ids_row_stride,
         float* out_scores, int64_t* out_ids, void* ws, size_t ws_bytes, int n_cu, hipStream_t stream,
         const int64_t* f_invalid = nullptr, int f_width = 0, int f_k = 0, const unsigned short* scores16 = nullptr,
         const int32_t* run_if = nullptr, const int64_t* ids_index = nullptr, int64_t ids_index_ld = 0);   // ids_index: see map_id (topk.hip)
// scores16 != NULL: the rows are bf16 bit patterns (ld, n in elements); only where topk_bf16_source_ok says so
bool topk_bf16_source_ok(int rows, int64_t n, int k);
int pack_candidates(const float* scores, const int64_t* ids, int rows, int k_local, int k, int64_t* msg, hipStream_t stream);
int range_flag(const int32_t* v, int n, int lo, int hi, int32_t* flag, hipStream_t stream);
int rescore_verdict(const float* row_stats, int rows, float default_eps, float safety, const float* guard, int64_t guard_count, float guard_limit,
                    float* state, hipStream_t stream);
int margin_stats(const float* kth, int64_t ld, int col, const float* m_max, const float* err_max, int rows, float* row_stats, hipStream_t stream);
// arithmetic-model probes (arith_model.hip)
int mfma_probe_f16(const unsigned short* a, const unsigned short* b, const float* c, float* d, int64_t n, hipStream_t stream);
int mfma_probe_f32(const float* a, const float* b, const float* c, float* d, int64_t n, hipStream_t stream);
int scalar_probe(const float* x, int64_t n, float* out, hipStream_t stream);
int rescore_select(const float* exact, int64_t ld, const float* approx, const float* approx_dense, int64_t ld_dense, const int64_t* positions,
                   const int64_t* ids, int rows, int n_ranked, int kc, int k, float margin_eps, float check_eps, int one_sided, float* out_scores,
                   int64_t* out_ids, int* ok, float* stats, hipStream_t stream);
size_t candidates_workspace_bytes(int rows);
int candidates_select(const float* scores, int64_t ld, int rows, int64_t n, int cap, float lo, float hi, void* ws, int64_t* out_pos, float* out_approx,
                      int64_t cand_ld, int n_cu, hipStream_t stream);
int candidates_finish(const float* exact, int64_t ld, const float* approx, const int64_t* pos, int64_t cand_ld, int cap, void* ws, const int64_t* ids,
                      int64_t n_items, int rows, int k, float default_eps, float safety, int one_sided, const float* guard, int guard_per_row,
                      float guard_limit, float* out_scores, int64_t* out_ids, const int64_t* f_invalid, int f_width, int f_k, int64_t* f_out_ids,
                      float* f_out_scores, float* state, float* state_host, int64_t* msg, hipStream_t stream);
struct MergeVerdict {   // rails_merge_candidates_verdict: the global verdict of the item-sharded proved top-k, inside the merge launch
  float default_eps, safety; const float* guard; int guard_per_row; float guard_limit; float* state; float* state_host; unsigned int* call;
};
int merge_candidates(const int64_t* gathered, int R, int rows, int k, int k_out, float* out_scores, int64_t* out_ids,
                     hipStream_t stream, const int64_t* f_invalid = nullptr, int f_width = 0, int f_k = 0, const MergeVerdict* verdict = nullptr);
int filter_seen(const int64_t* top_ids, const float* top_scores, int rows, int k_prime, const int64_t* invalid,
                int width, int k, int64_t* out_ids, float* out_scores, hipStream_t stream);

}  // namespace mol
