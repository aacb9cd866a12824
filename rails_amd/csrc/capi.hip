// extern "C" entry points declared in include/rails_amd.h: argument validation + dispatch.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "mol_kernels.h"
#include "mol_layout.h"

namespace mol {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int fail(int code, const char* what) {
  if (code == kOk) return code;
  if (code == kErrLaunch) {
    const hipError_t e = hipGetLastError();
    set_error("%s: HIP launch failed (%s)", what, hipGetErrorString(e));
  } else if (g_err[0] == '\0' || code == kErrInvalid) {
    if (g_err[0] == '\0') set_error("%s: error %d", what, code);
  }
  return code;
}

static bool shape_ok(const Shape* s) {
  if (!s) { set_error("shape is NULL"); return false; }
  if (s->query_embedding_dim <= 0 || s->item_embedding_dim <= 0 || s->dot_product_dimension <= 0 ||
      s->query_dot_product_groups <= 0 || s->item_dot_product_groups <= 0) {
    set_error("shape has a non-positive dimension");
    return false;
  }
  if (s->num_uid_tables < 0 || s->num_uid_tables > RAILS_MAX_UID_TABLES ||
      s->num_uid_tables >= s->query_dot_product_groups) {
    set_error("num_uid_tables = %d out of range", s->num_uid_tables);
    return false;
  }
  if (!(s->temperature > 0.0f)) { set_error("temperature must be > 0"); return false; }
  return true;
}

static bool shape_supported(const Shape* s) {
  if (!shape_ok(s)) return false;
  if (s->gating_qi_hidden_dim <= 0 && (s->precision != RAILS_PRECISION_FP32 || !score_extra_shape(*s))) {
    set_error("a pair gate without hidden layer (gating_qi_hidden_dim <= 0) is built in precision fp32 for P_Q x P_X x d = 8x8x32, 8x4x64, 16x4x32 "
              "(mol_score_extra_shapes.h)");
    return false;
  }
  if (s->gating_combination != RAILS_COMBINE_GLU_SILU && s->gating_combination != RAILS_COMBINE_NONE) {
    set_error("gating_combination must be RAILS_COMBINE_GLU_SILU or RAILS_COMBINE_NONE, got %d", s->gating_combination);
    return false;
  }
  if (s->gating_combination == RAILS_COMBINE_GLU_SILU && (!s->gating_has_query || !s->gating_has_item)) {
    set_error("gating_combination glu_silu needs the query-only and the item-only gate part (the reference multiplies them)");
    return false;
  }
  if ((s->gating_has_query && s->gating_query_hidden_dim <= 0) || (s->gating_has_item && s->gating_item_hidden_dim <= 0)) {
    set_error("gating hidden dims must be > 0 for the gate parts that exist");
    return false;
  }
  if (s->precision != RAILS_PRECISION_FP32 && s->precision != RAILS_PRECISION_F16X3 && s->precision != RAILS_PRECISION_F16X1) {
    set_error("precision must be RAILS_PRECISION_FP32, _F16X3 or _F16X1, got %d", s->precision);
    return false;
  }
  if (is_split(*s) && !s->dot_product_l2_norm) {
    set_error("precision f16x3 needs dot_product_l2_norm = 1 (cross logits bounded by 1/temperature keep the f16 operands in range)");
    return false;
  }
  if (!score_supported(*s)) {
    set_error("no fused scoring kernel for P_Q x P_X x d = %dx%dx%d with gating_qi_hidden_dim = %d "
              "(built: 8x4x64, 8x4x128, 8x8x32, 16x16x64 with 128)",
              s->query_dot_product_groups, s->item_dot_product_groups, s->dot_product_dimension,
              s->gating_qi_hidden_dim);
    return false;
  }
  return true;
}

static int compute_units() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

}  // namespace mol

using namespace mol;

extern "C" {

const char* rails_last_error(void) { return g_err; }

int rails_device_compute_units(void) {
  const int n = compute_units();
  if (n < 0) { set_error("no HIP device"); return RAILS_ELAUNCH; }
  return n;
}

int rails_mol_shape_supported(const rails_mol_shape* shape) { return shape_supported(shape) ? 1 : 0; }

size_t rails_mol_gate_pack_floats(const rails_mol_shape* s) {
  if (!shape_ok(s)) return 0;
  // the small-unit kernel's shapes carry the pair-gate weights twice: 32x32x2 fragment order, then 16x16x4 fragment order
  return gate_pack32_floats(*s) * (score_small_shape(*s) ? 2 : 1);
}

int rails_mol_pack_gate_weights(const rails_mol_shape* s, const rails_mol_weights* w, float* gate_pack, void* stream) {
  g_err[0] = '\0';
  if (!shape_supported(s)) return RAILS_ENOTSUP;
  if (!w || !gate_pack || !w->gqi_w1 || !w->gqi_b1 || (s->gating_qi_hidden_dim > 0 && (!w->gqi_w2 || !w->gqi_b2))) {
    set_error("pack_gate_weights: NULL pointer");
    return RAILS_EINVAL;
  }
  if (is_split(*s)) return fail(pack_gate_weights_split(*s, *w, gate_pack, (hipStream_t)stream), "pack_gate_weights");
  return fail(pack_gate_weights(*s, *w, gate_pack, (hipStream_t)stream), "pack_gate_weights");
}

size_t rails_mol_index_floats(const rails_mol_shape* s, int64_t n_items) {
  if (!shape_ok(s) || n_items < 0) return 0;
  return (size_t)(num_tiles(n_items) * tile_floats(*s));
}

int rails_mol_index_build(const rails_mol_shape* s, const rails_mol_weights* w, const float* items, int64_t n_items,
                          float* index, void* stream) {
  g_err[0] = '\0';
  if (!shape_supported(s)) return RAILS_ENOTSUP;
  if (n_items < 0) { set_error("index_build: n_items < 0"); return RAILS_EINVAL; }
  if (n_items == 0) return RAILS_OK;
  if (!w || !items || !index || !w->i_proj_w || !w->i_proj_b || (s->gating_has_item && (!w->gi_w1 || !w->gi_b1 || !w->gi_w2)) ||
      (s->item_hidden_dim > 0 && (!w->i_glu_w || !w->i_glu_b))) {
    set_error("index_build: NULL pointer");
    return RAILS_EINVAL;
  }
  int r = index_build(*s, *w, items, n_items, index, (hipStream_t)stream);
  if (r == kOk && is_split(*s)) r = index_split_inplace(*s, index, n_items, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "index_build");
}

int rails_mol_index_unpack(const rails_mol_shape* s, const float* index, int64_t n_items, float* ex_out, float* gi_out,
                           void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (n_items <= 0 || (!ex_out && !gi_out)) return RAILS_OK;
  if (!index) { set_error("index_unpack: NULL index"); return RAILS_EINVAL; }
  return fail(index_unpack(*s, index, n_items, ex_out, gi_out, (hipStream_t)stream), "index_unpack");
}

int rails_mol_index_gather(const rails_mol_shape* s, const float* index, int64_t n_items, const int64_t* cand_idx,
                           int64_t n_rows, int64_t n_cand, float* out_index, void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (n_rows < 0 || n_cand < 0) { set_error("index_gather: negative size"); return RAILS_EINVAL; }
  if (n_rows == 0 || n_cand == 0) return RAILS_OK;
  if (!index || !cand_idx || !out_index) { set_error("index_gather: NULL pointer"); return RAILS_EINVAL; }
  const int r = index_gather(*s, index, n_items, cand_idx, n_rows, n_cand, out_index, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "index_gather");
}

size_t rails_mol_query_pack_floats(const rails_mol_shape* s, int32_t batch) {
  if (!shape_ok(s) || batch < 0) return 0;
  const int QT = queries_per_group(*s);
  const size_t groups = (size_t)((batch + QT - 1) / QT);
  return groups * 32 * (size_t)s->dot_product_dimension + (size_t)batch * (size_t)num_logits(*s) +
         query_scratch_floats(*s, batch);   // + the prologue's own scratch rows (GLU output, first gate layer)
}

static int query_prologue_checked(const rails_mol_shape* s, const rails_mol_weights* w, const float* queries, const int64_t* user_ids,
                                  int32_t batch, float* query_pack, float* query_pack_other, float* eq_out, float* gq_out, void* stream);

int rails_mol_query_prologue(const rails_mol_shape* s, const rails_mol_weights* w, const float* queries,
                             const int64_t* user_ids, int32_t batch, float* query_pack, float* eq_out, float* gq_out,
                             void* stream) {
  return query_prologue_checked(s, w, queries, user_ids, batch, query_pack, nullptr, eq_out, gq_out, stream);
}

int rails_mol_query_prologue_both(const rails_mol_shape* s, const rails_mol_weights* w, const float* queries,
                                  const int64_t* user_ids, int32_t batch, float* query_pack, float* query_pack_other,
                                  void* stream) {
  if (!query_pack_other) { g_err[0] = '\0'; set_error("query_prologue_both: NULL pointer"); return RAILS_EINVAL; }
  return query_prologue_checked(s, w, queries, user_ids, batch, query_pack, query_pack_other, nullptr, nullptr, stream);
}

static int query_prologue_checked(const rails_mol_shape* s, const rails_mol_weights* w, const float* queries, const int64_t* user_ids,
                                  int32_t batch, float* query_pack, float* query_pack_other, float* eq_out, float* gq_out, void* stream) {
  g_err[0] = '\0';
  if (!shape_supported(s)) return RAILS_ENOTSUP;
  if (batch < 0) { set_error("query_prologue: batch < 0"); return RAILS_EINVAL; }
  if (batch == 0) return RAILS_OK;
  if (!w || !queries || !query_pack || !w->q_proj_w || !w->q_proj_b || (s->query_hidden_dim > 0 && (!w->q_glu_w || !w->q_glu_b)) ||
      (s->gating_has_query && (!w->gq_w1 || !w->gq_b1 || !w->gq_w2))) {
    set_error("query_prologue: NULL pointer");
    return RAILS_EINVAL;
  }
  if (s->num_uid_tables > 0) {
    if (!user_ids) { set_error("query_prologue: user_ids is required when num_uid_tables > 0"); return RAILS_EINVAL; }
    for (int t = 0; t < s->num_uid_tables; ++t)
      if (!w->uid_table[t] || w->uid_hash_size[t] <= 0) { set_error("query_prologue: uid table %d missing", t); return RAILS_EINVAL; }
  }
  return fail(query_prologue(*s, *w, queries, user_ids, batch, query_pack, eq_out, gq_out, (hipStream_t)stream, query_pack_other),
              "query_prologue");
}

// Launch arguments of a scoring pass over a shared corpus (per_row = 0) or per-row candidates (per_row = 1).
static int fill_score_args(const rails_mol_shape* s, const float* gate_pack, const float* query_pack, int32_t batch, const float* index,
                           int64_t n_items, float* logits, int64_t ld, int per_row, ScoreArgs* out, const int32_t* run_if = nullptr) {
  ScoreArgs a{};
  const int QT = queries_per_group(*s);
  a.n_groups = (batch + QT - 1) / QT;
  a.wpack = gate_pack;
  a.eqfrag = query_pack;
  a.gqfrag = query_pack + (int64_t)a.n_groups * 32 * s->dot_product_dimension;
  a.ipack = index;
  a.logits = logits;
  a.ld = ld;
  a.n_items = n_items;
  a.n_tiles = num_tiles(n_items);
  a.B = batch;
  a.per_row = per_row;
  a.temperature = s->temperature;
  a.rcp_temperature = 1.0f / s->temperature;
  a.split = is_split(*s) ? 1 : 0;
  a.single = s->precision == RAILS_PRECISION_F16X1 ? 1 : 0;
  a.combine_none = s->gating_combination == RAILS_COMBINE_NONE ? 1 : 0;
  a.run_if = run_if;
  *out = a;
  return kOk;
}

static int score_common(const rails_mol_shape* s, const float* gate_pack, const float* query_pack, int32_t batch,
                        const float* index, int64_t n_items, float* logits, int64_t ld, int per_row, void* stream,
                        const char* what, const int32_t* run_if = nullptr) {
  g_err[0] = '\0';
  if (!shape_supported(s)) return RAILS_ENOTSUP;
  if (batch < 0 || n_items < 0) { set_error("%s: negative size", what); return RAILS_EINVAL; }
  if (batch == 0 || n_items == 0) return RAILS_OK;
  if (!gate_pack || !query_pack || !index || !logits) { set_error("%s: NULL pointer", what); return RAILS_EINVAL; }
  if (ld < n_items) { set_error("%s: ld (%lld) < n_items (%lld)", what, (long long)ld, (long long)n_items); return RAILS_EINVAL; }
  if (per_row && n_items % 32 != 0) { set_error("%s: n_cand must be a multiple of 32", what); return RAILS_EINVAL; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("%s: no HIP device", what); return RAILS_ELAUNCH; }
  ScoreArgs a;
  fill_score_args(s, gate_pack, query_pack, batch, index, n_items, logits, ld, per_row, &a, run_if);
  const int r = score_launch(*s, a, cu, (hipStream_t)stream);
  return r == kOk ? r : fail(r, what);
}

int rails_mol_score_dense(const rails_mol_shape* s, const float* gate_pack, const float* query_pack, int32_t batch,
                          const float* index, int64_t n_items, float* logits, int64_t ld, const int32_t* run_if, void* stream) {
  return score_common(s, gate_pack, query_pack, batch, index, n_items, logits, ld, 0, stream, "score_dense", run_if);
}

int rails_mol_score_dense_upper_supported(const rails_mol_shape* s) {
  if (!s || !shape_supported(s) || s->precision != RAILS_PRECISION_F16X3) return 0;
  const int cu = compute_units();
  if (cu <= 0) return 0;
  ScoreArgs a;
  fill_score_args(s, nullptr, nullptr, 32, nullptr, 32, nullptr, 32, 0, &a);
  a.upper = 1;
  a.dry_run = 1;
  const int r = score_launch(*s, a, cu, nullptr);
  g_err[0] = '\0';
  return r == kOk ? 1 : 0;
}

int rails_mol_score_dense_upper(const rails_mol_shape* s, const float* gate_pack, const float* query_pack, int32_t batch, const float* index,
                                int64_t n_items, float ub2, float ub1, float ub0, float* logits, int64_t ld, const int32_t* run_if, void* stream) {
  g_err[0] = '\0';
  if (!shape_supported(s)) return RAILS_ENOTSUP;
  if (batch < 0 || n_items < 0) { set_error("score_dense_upper: negative size"); return RAILS_EINVAL; }
  if (!(ub2 >= 0.0f && ub1 >= 0.0f && ub0 >= 0.0f) || !(ub2 + ub1 + ub0 < INFINITY)) {
    set_error("score_dense_upper: the bound polynomial needs three finite coefficients >= 0");
    return RAILS_EINVAL;
  }
  if (batch == 0 || n_items == 0) return RAILS_OK;
  if (!gate_pack || !query_pack || !index || !logits) { set_error("score_dense_upper: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n_items) { set_error("score_dense_upper: ld (%lld) < n_items (%lld)", (long long)ld, (long long)n_items); return RAILS_EINVAL; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("score_dense_upper: no HIP device"); return RAILS_ELAUNCH; }
  ScoreArgs a;
  fill_score_args(s, gate_pack, query_pack, batch, index, n_items, logits, ld, 0, &a, run_if);
  a.upper = 1;
  a.ub2 = ub2;
  a.ub1 = ub1;
  a.ub0 = ub0;
  const int r = score_launch(*s, a, cu, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "score_dense_upper");
}

int rails_mol_score_indexed_supported(const rails_mol_shape* s, int32_t batch, int64_t n_cand) {
  if (!s || !shape_supported(s) || is_split(*s) || batch <= 0 || n_cand <= 0) return 0;
  const int cu = compute_units();
  if (cu <= 0) return 0;
  ScoreArgs a;
  fill_score_args(s, nullptr, nullptr, batch, nullptr, n_cand, nullptr, n_cand, 1, &a);
  a.cand_pos = reinterpret_cast<const int64_t*>(8);   // never dereferenced: dry run
  a.dry_run = 1;
  const int r = score_launch(*s, a, cu, nullptr);
  g_err[0] = '\0';
  return r == kOk ? 1 : 0;
}

int rails_mol_score_indexed(const rails_mol_shape* s, const float* gate_pack, const float* query_pack, int32_t batch, const float* index,
                            int64_t n_items, const int64_t* positions, int64_t n_cand, float* logits, int64_t ld, void* stream) {
  g_err[0] = '\0';
  if (!shape_supported(s)) return RAILS_ENOTSUP;
  if (batch < 0 || n_items <= 0 || n_cand < 0) { set_error("score_indexed: bad size"); return RAILS_EINVAL; }
  if (batch == 0 || n_cand == 0) return RAILS_OK;
  if (!gate_pack || !query_pack || !index || !positions || !logits) { set_error("score_indexed: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n_cand) { set_error("score_indexed: ld < n_cand"); return RAILS_EINVAL; }
  if (is_split(*s)) { set_error("score_indexed: exact-fp32 precision only"); return RAILS_ENOTSUP; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("score_indexed: no HIP device"); return RAILS_ELAUNCH; }
  ScoreArgs a;
  fill_score_args(s, gate_pack, query_pack, batch, index, n_cand, logits, ld, 1, &a);
  a.cand_pos = positions;
  a.index_items = n_items;
  const int r = score_launch(*s, a, cu, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "score_indexed");
}

size_t rails_mol_index_rows_floats(const rails_mol_shape* s, int64_t n_items) {
  if (!shape_ok(s) || n_items < 0 || is_split(*s)) return 0;
  return (size_t)n_items * (size_t)(tile_floats(*s) / 32);
}

int rails_mol_index_rows_build(const rails_mol_shape* s, const float* index, int64_t n_items, float* rows, void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (is_split(*s)) { set_error("index_rows_build: exact-fp32 index only"); return RAILS_ENOTSUP; }
  if (n_items < 0) { set_error("index_rows_build: n_items < 0"); return RAILS_EINVAL; }
  if (n_items == 0) return RAILS_OK;
  if (!index || !rows) { set_error("index_rows_build: NULL pointer"); return RAILS_EINVAL; }
  return fail(index_rows_build(*s, index, n_items, rows, (hipStream_t)stream), "index_rows_build");
}

int rails_mol_score_indexed_rows(const rails_mol_shape* s, const float* gate_pack, const float* query_pack, int32_t batch, const float* index_rows,
                                 int64_t n_items, const int64_t* positions, int64_t n_cand, float* logits, int64_t ld, const int32_t* cand_counts,
                                 void* stream) {
  g_err[0] = '\0';
  if (!shape_supported(s)) return RAILS_ENOTSUP;
  if (batch < 0 || n_items <= 0 || n_cand < 0) { set_error("score_indexed_rows: bad size"); return RAILS_EINVAL; }
  if (batch == 0 || n_cand == 0) return RAILS_OK;
  if (!gate_pack || !query_pack || !index_rows || !positions || !logits) { set_error("score_indexed_rows: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n_cand) { set_error("score_indexed_rows: ld < n_cand"); return RAILS_EINVAL; }
  if (is_split(*s)) { set_error("score_indexed_rows: exact-fp32 precision only"); return RAILS_ENOTSUP; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("score_indexed_rows: no HIP device"); return RAILS_ELAUNCH; }
  ScoreArgs a;
  fill_score_args(s, gate_pack, query_pack, batch, index_rows, n_cand, logits, ld, 1, &a);
  a.cand_pos = positions;
  a.index_items = n_items;
  a.irows = index_rows;
  a.cand_count = cand_counts;
  const int r = score_launch(*s, a, cu, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "score_indexed_rows");
}

int rails_mol_score_candidates(const rails_mol_shape* s, const float* gate_pack, const float* query_pack, int32_t batch,
                               const float* cand_index, int64_t n_cand, float* logits, int64_t ld, void* stream) {
  return score_common(s, gate_pack, query_pack, batch, cand_index, n_cand, logits, ld, 1, stream, "score_candidates");
}

size_t rails_mips_index_floats(int32_t dim, int64_t n_items) {
  if (dim <= 0 || n_items < 0) return 0;
  return (size_t)num_tiles(n_items) * 32 * (size_t)((dim + 7) / 8 * 8);
}

int rails_mips_index_build(const float* items, int64_t n_items, int32_t dim, float* index, void* stream) {
  g_err[0] = '\0';
  if (dim <= 0 || n_items < 0) { set_error("mips_index_build: bad size"); return RAILS_EINVAL; }
  if (n_items == 0) return RAILS_OK;
  if (!items || !index) { set_error("mips_index_build: NULL pointer"); return RAILS_EINVAL; }
  return fail(mips_pack_items(items, n_items, dim, index, (hipStream_t)stream), "mips_index_build");
}

size_t rails_mips_query_ws_floats(int32_t dim, int32_t batch) {
  if (dim <= 0 || batch < 0) return 0;
  return (size_t)((batch + 31) / 32) * 32 * (size_t)((dim + 7) / 8 * 8);
}

int rails_mips_score(const float* queries, int32_t batch, int32_t dim, const float* index, int64_t n_items, float* query_ws,
                     float* logits, int64_t ld, void* stream) {
  g_err[0] = '\0';
  if (dim <= 0 || batch < 0 || n_items < 0) { set_error("mips_score: bad size"); return RAILS_EINVAL; }
  if (batch == 0 || n_items == 0) return RAILS_OK;
  if (!queries || !index || !query_ws || !logits) { set_error("mips_score: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n_items) { set_error("mips_score: ld < n_items"); return RAILS_EINVAL; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("mips_score: no HIP device"); return RAILS_ELAUNCH; }
  return fail(mips_score(queries, batch, dim, index, n_items, query_ws, logits, ld, cu, (hipStream_t)stream), "mips_score");
}

int rails_dot_rowwise(const float* queries, const float* items, int64_t n_queries, int32_t n_cand, int32_t dim, int32_t r,
                      float* out, void* stream) {
  g_err[0] = '\0';
  if (n_queries < 0 || n_cand < 0 || dim <= 0 || r <= 0 || n_queries % r != 0) { set_error("dot_rowwise: bad size"); return RAILS_EINVAL; }
  if (n_queries == 0 || n_cand == 0) return RAILS_OK;
  if (!queries || !items || !out) { set_error("dot_rowwise: NULL pointer"); return RAILS_EINVAL; }
  return fail(dot_rowwise(queries, items, n_queries, n_cand, dim, r, out, (hipStream_t)stream), "dot_rowwise");
}

size_t rails_mol_coarse_table_bytes(const rails_mol_shape* s, int64_t n_items) {
  if (!shape_ok(s) || n_items < 0) return 0;
  return (size_t)n_items * (size_t)s->dot_product_dimension * 2;
}

int rails_mol_coarse_build(const rails_mol_shape* s, const float* index, int64_t n_items, void* table, void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (s->dot_product_dimension % 8 != 0) { set_error("coarse_build: d must be a multiple of 8"); return RAILS_ENOTSUP; }
  if (n_items < 0) { set_error("coarse_build: n_items < 0"); return RAILS_EINVAL; }
  if (n_items == 0) return RAILS_OK;
  if (!index || !table) { set_error("coarse_build: NULL pointer"); return RAILS_EINVAL; }
  if (is_split(*s)) { set_error("coarse_build: needs an fp32-format item index (build one with precision = RAILS_PRECISION_FP32)"); return RAILS_ENOTSUP; }
  return fail(coarse_build(*s, index, n_items, table, (hipStream_t)stream), "coarse_build");
}

size_t rails_mol_coarse_prefilter_bytes(const rails_mol_shape* s, int64_t n_items) {
  if (!shape_ok(s)) return 0;
  return coarse_prefilter_bytes(*s, n_items);
}

int rails_mol_coarse_prefilter_build(const rails_mol_shape* s, const void* table, int64_t n_items, void* prefilter, void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (n_items <= 0) { set_error("coarse_prefilter_build: n_items must be positive"); return RAILS_EINVAL; }
  if (!table || !prefilter) { set_error("coarse_prefilter_build: NULL pointer"); return RAILS_EINVAL; }
  return fail(coarse_prefilter_build(*s, table, n_items, prefilter, (hipStream_t)stream), "coarse_prefilter_build");
}

size_t rails_mol_coarse_topk_workspace_bytes(const rails_mol_shape* s, int32_t batch, int64_t n_items, int32_t k_prime) {
  if (!shape_ok(s) || batch <= 0) return 0;
  return coarse_topk_workspace_bytes(*s, batch, n_items, k_prime);
}

int32_t rails_mol_coarse_topk_capacity(int32_t batch, int64_t n_items, int32_t k_prime) {
  return batch > 0 ? coarse_topk_capacity(batch, n_items, k_prime) : 0;
}

int rails_mol_coarse_topk(const rails_mol_shape* s, const float* eq, int32_t batch, int32_t average_queries, const void* table,
                          int64_t n_items, int32_t k_prime, void* workspace, size_t workspace_bytes, float* out_scores,
                          int64_t* out_positions, int32_t* out_counts, int32_t* out_of_range, void* prefilter, void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (batch < 0 || n_items < 0 || k_prime < 0) { set_error("coarse_topk: negative size"); return RAILS_EINVAL; }
  if (k_prime > n_items) { set_error("coarse_topk: selected index k out of range (k = %d > n = %lld)", k_prime, (long long)n_items); return RAILS_EINVAL; }
  if (batch == 0 || k_prime == 0) return RAILS_OK;
  if (!eq || !table || !workspace || !out_scores || !out_positions || !out_counts) { set_error("coarse_topk: NULL pointer"); return RAILS_EINVAL; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("coarse_topk: no HIP device"); return RAILS_ELAUNCH; }
  const int r = coarse_topk(*s, eq, batch, average_queries ? 1 : 0, table, n_items, k_prime, workspace, workspace_bytes,
                            out_scores, out_positions, out_counts, out_of_range, prefilter, cu, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "coarse_topk");
}

int rails_mol_coarse_score(const rails_mol_shape* s, const float* eq, int32_t batch, int32_t average_queries,
                           const void* table, int64_t n_items, float* scores, int64_t ld, const int32_t* run_if, void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (batch < 0 || n_items < 0) { set_error("coarse_score: negative size"); return RAILS_EINVAL; }
  if (batch == 0 || n_items == 0) return RAILS_OK;
  if (!eq || !table || !scores) { set_error("coarse_score: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n_items) { set_error("coarse_score: ld < n_items"); return RAILS_EINVAL; }
  const int r = coarse_score(*s, eq, batch, average_queries ? 1 : 0, table, n_items, scores, ld, (hipStream_t)stream, run_if);
  return r == kOk ? r : fail(r, "coarse_score");
}

size_t rails_mol_component_table_bytes(const rails_mol_shape* s, int64_t n_items) {
  if (!shape_ok(s) || n_items < 0) return 0;
  return (size_t)n_items * (size_t)s->item_dot_product_groups * (size_t)s->dot_product_dimension * 2;
}

int rails_mol_component_build(const rails_mol_shape* s, const float* index, int64_t n_items, void* table, int64_t n_total, int64_t first_item,
                              void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (s->dot_product_dimension % 8 != 0) { set_error("component_build: d must be a multiple of 8"); return RAILS_ENOTSUP; }
  if (n_items < 0 || first_item < 0 || first_item + n_items > n_total) { set_error("component_build: items [%lld, %lld) outside a table of %lld", (long long)first_item, (long long)(first_item + n_items), (long long)n_total); return RAILS_EINVAL; }
  if (n_items == 0) return RAILS_OK;
  if (!index || !table) { set_error("component_build: NULL pointer"); return RAILS_EINVAL; }
  if (is_split(*s)) { set_error("component_build: needs an fp32-format item index (build one with precision = RAILS_PRECISION_FP32)"); return RAILS_ENOTSUP; }
  return fail(component_build(*s, index, n_items, table, n_total, first_item, (hipStream_t)stream), "component_build");
}

size_t rails_mol_component_topk_workspace_bytes(const rails_mol_shape* s, int32_t batch, int64_t n_items, int32_t k_group) {
  if (!shape_ok(s) || batch <= 0) return 0;
  return component_topk_workspace_bytes(*s, batch, n_items, k_group);
}

int32_t rails_mol_component_topk_capacity(const rails_mol_shape* s, int32_t batch, int64_t n_items, int32_t k_group) {
  if (!shape_ok(s) || batch <= 0) return 0;
  return component_topk_capacity(*s, batch, n_items, k_group);
}

int rails_mol_component_topk(const rails_mol_shape* s, const float* eq, int32_t batch, const void* table, int64_t n_items,
                             int32_t k_group, void* workspace, size_t workspace_bytes, float* out_scores,
                             int64_t* out_positions, int32_t* out_counts, int32_t* out_of_range, void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (batch < 0 || n_items < 0 || k_group < 0) { set_error("component_topk: negative size"); return RAILS_EINVAL; }
  if (k_group > n_items) { set_error("component_topk: selected index k out of range (k = %d > n = %lld)", k_group, (long long)n_items); return RAILS_EINVAL; }
  if (batch == 0 || k_group == 0) return RAILS_OK;
  if (!eq || !table || !workspace || !out_scores || !out_positions || !out_counts) { set_error("component_topk: NULL pointer"); return RAILS_EINVAL; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("component_topk: no HIP device"); return RAILS_ELAUNCH; }
  const int r = component_topk(*s, eq, batch, table, n_items, k_group, workspace, workspace_bytes, out_scores, out_positions,
                               out_counts, out_of_range, cu, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "component_topk");
}

int rails_mol_component_score(const rails_mol_shape* s, const float* eq, int32_t batch, const void* table, int64_t n_items,
                              float* scores, int64_t ld, const int32_t* run_if, void* stream) {
  g_err[0] = '\0';
  if (!shape_ok(s)) return RAILS_EINVAL;
  if (batch < 0 || n_items < 0) { set_error("component_score: negative size"); return RAILS_EINVAL; }
  if (batch == 0 || n_items == 0) return RAILS_OK;
  if (!eq || !table || !scores) { set_error("component_score: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n_items) { set_error("component_score: ld < n_items"); return RAILS_EINVAL; }
  const int r = component_score(*s, eq, batch, table, n_items, scores, ld, (hipStream_t)stream, run_if);
  return r == kOk ? r : fail(r, "component_score");
}

int rails_sort_rows_i64(const int64_t* in, int32_t rows, int32_t n, int64_t* out, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || n < 0) { set_error("sort_rows_i64: negative size"); return RAILS_EINVAL; }
  if (rows == 0 || n == 0) return RAILS_OK;
  if (!in || !out) { set_error("sort_rows_i64: NULL pointer"); return RAILS_EINVAL; }
  const int r = sort_rows_i64(in, rows, n, out, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "sort_rows_i64");
}

int rails_mask_sorted_duplicates(const int64_t* sorted_idx, float* scores, int64_t ld, int32_t rows, int32_t n, float fill,
                                 void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || n < 0) { set_error("mask_sorted_duplicates: negative size"); return RAILS_EINVAL; }
  if (rows == 0 || n == 0) return RAILS_OK;
  if (!sorted_idx || !scores) { set_error("mask_sorted_duplicates: NULL pointer"); return RAILS_EINVAL; }
  return fail(mask_sorted_duplicates(sorted_idx, scores, ld, rows, n, fill, (hipStream_t)stream), "mask_sorted_duplicates");
}

size_t rails_topk_workspace_bytes(int32_t rows, int64_t n, int32_t k) {
  if (rows <= 0 || n <= 0 || k <= 0) return 256;
  return topk_workspace_bytes(rows, n, k);
}

int rails_topk(const float* scores, int64_t ld, int32_t rows, int64_t n, int32_t k, int32_t sorted, const int64_t* ids,
               int64_t ids_row_stride, float* out_scores, int64_t* out_ids, void* workspace, size_t workspace_bytes,
               const int32_t* run_if, void* stream) {
  (void)sorted;  // the descending order returned is also a valid unsorted answer
  g_err[0] = '\0';
  if (rows < 0 || n < 0 || k < 0) { set_error("topk: negative size"); return RAILS_EINVAL; }
  if (k > n) { set_error("topk: selected index k out of range (k = %d > n = %lld)", k, (long long)n); return RAILS_EINVAL; }
  if (rows == 0 || k == 0) return RAILS_OK;
  if (!scores || !out_scores || !out_ids) { set_error("topk: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n) { set_error("topk: ld < n"); return RAILS_EINVAL; }
  if (n > 16384 && !workspace) { set_error("topk: workspace is required for n > 16384"); return RAILS_ENOMEM; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("topk: no HIP device"); return RAILS_ELAUNCH; }
  const int r = topk(scores, ld, rows, n, k, ids, ids_row_stride, out_scores, out_ids, workspace, workspace_bytes, cu,
                     (hipStream_t)stream, nullptr, 0, 0, nullptr, run_if);
  return r == kOk ? r : fail(r, "topk");
}

int rails_topk_candidates(const float* scores, int64_t ld, int32_t rows, int32_t n_cand, int32_t k, const int64_t* positions,
                          const int64_t* ids, float* out_scores, int64_t* out_ids, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || n_cand < 0 || k < 0) { set_error("topk_candidates: negative size"); return RAILS_EINVAL; }
  if (k > n_cand) { set_error("topk_candidates: selected index k out of range (k = %d > n = %d)", k, n_cand); return RAILS_EINVAL; }
  if (rows == 0 || k == 0) return RAILS_OK;
  if (!scores || !positions || !out_scores || !out_ids) { set_error("topk_candidates: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n_cand) { set_error("topk_candidates: ld < n_cand"); return RAILS_EINVAL; }
  if (n_cand > 16384) { set_error("topk_candidates: n_cand = %d exceeds 16384", n_cand); return RAILS_ENOTSUP; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("topk_candidates: no HIP device"); return RAILS_ELAUNCH; }
  const int r = topk(scores, ld, rows, n_cand, k, ids, 0, out_scores, out_ids, nullptr, 0, cu, (hipStream_t)stream, nullptr, 0, 0, nullptr,
                     nullptr, positions, n_cand);
  return r == kOk ? r : fail(r, "topk_candidates");
}

int rails_topk_candidates_filtered(const float* scores, int64_t ld, int32_t rows, int32_t n_cand, int32_t k_prime, const int64_t* positions,
                                   const int64_t* ids, const int64_t* invalid_ids, int32_t width, int32_t k, int64_t* out_ids,
                                   float* out_scores, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || n_cand < 0 || k_prime < 0 || k < 0 || width < 0) { set_error("topk_candidates_filtered: negative size"); return RAILS_EINVAL; }
  if (k_prime > n_cand || k > k_prime) { set_error("topk_candidates_filtered: need k <= k' <= n_cand (k = %d, k' = %d, n = %d)", k, k_prime, n_cand); return RAILS_EINVAL; }
  if (rows == 0 || k == 0) return RAILS_OK;
  if (!scores || !positions || !out_scores || !out_ids || (width > 0 && !invalid_ids)) { set_error("topk_candidates_filtered: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n_cand) { set_error("topk_candidates_filtered: ld < n_cand"); return RAILS_EINVAL; }
  if (n_cand <= 1024 || n_cand > 8192 || !topk_can_fuse_filter(n_cand, k_prime, width, k)) {
    set_error("topk_candidates_filtered: unsupported size (n_cand = %d, k' = %d, width = %d, k = %d)", n_cand, k_prime, width, k);
    return RAILS_ENOTSUP;
  }
  const int cu = compute_units();
  if (cu <= 0) { set_error("topk_candidates_filtered: no HIP device"); return RAILS_ELAUNCH; }
  static const int64_t no_seen = 0;     // width 0: the filter still runs (it copies the first k), its list is never read
  const int r = topk(scores, ld, rows, n_cand, k_prime, ids, 0, out_scores, out_ids, nullptr, 0, cu, (hipStream_t)stream,
                     width > 0 ? invalid_ids : &no_seen, width, k, nullptr, nullptr, positions, n_cand);
  return r == kOk ? r : fail(r, "topk_candidates_filtered");
}

size_t rails_rerank_workspace_bytes(int32_t rows, int32_t n_cand) { return rows > 0 && n_cand > 0 ? rerank_workspace_bytes(rows, n_cand) : 0; }

int rails_rerank_topk_filtered(const float* scores, int64_t ld, int32_t rows, int32_t n_cand, int32_t k_prime, const int64_t* positions,
                               const int64_t* ids, const int64_t* invalid_ids, int32_t width, int32_t k, void* workspace, size_t workspace_bytes,
                               int64_t* out_ids, float* out_scores, int32_t* out_of_range, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || n_cand < 0 || k_prime < 0 || k < 0 || width < 0) { set_error("rerank_topk_filtered: negative size"); return RAILS_EINVAL; }
  if (k_prime > n_cand || k > k_prime) { set_error("rerank_topk_filtered: need k <= k' <= n_cand (k = %d, k' = %d, n = %d)", k, k_prime, n_cand); return RAILS_EINVAL; }
  if (rows == 0 || k == 0) return RAILS_OK;
  if (!scores || !positions || !out_scores || !out_ids || !out_of_range || !workspace || (width > 0 && !invalid_ids)) { set_error("rerank_topk_filtered: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n_cand) { set_error("rerank_topk_filtered: ld < n_cand"); return RAILS_EINVAL; }
  static const int64_t no_seen = 0;
  const int r = rerank_topk_filtered(scores, ld, rows, n_cand, k_prime, positions, ids, width > 0 ? invalid_ids : &no_seen, width, k, workspace,
                                     workspace_bytes, out_ids, out_scores, out_of_range, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "rerank_topk_filtered");
}

int rails_topk_filter_fusable(int64_t n, int32_t k_prime, int32_t width, int32_t k) { return topk_can_fuse_filter(n, k_prime, width, k) ? 1 : 0; }

int rails_topk_filtered(const float* scores, int64_t ld, int32_t rows, int64_t n, int32_t k_prime, const int64_t* ids, int64_t ids_row_stride,
                        const int64_t* invalid_ids, int32_t width, int32_t k, int64_t* out_ids, float* out_scores, void* workspace,
                        size_t workspace_bytes, const int32_t* run_if, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || n < 0 || k_prime < 0 || k < 0 || width < 0) { set_error("topk_filtered: negative size"); return RAILS_EINVAL; }
  if (k_prime > n) { set_error("topk_filtered: selected index k out of range (k' = %d > n = %lld)", k_prime, (long long)n); return RAILS_EINVAL; }
  if (k > k_prime) { set_error("topk_filtered: k = %d > k' = %d", k, k_prime); return RAILS_EINVAL; }
  if (rows == 0 || k == 0) return RAILS_OK;
  if (!scores || !out_scores || !out_ids || !invalid_ids) { set_error("topk_filtered: NULL pointer"); return RAILS_EINVAL; }
  if (ld < n) { set_error("topk_filtered: ld < n"); return RAILS_EINVAL; }
  if (!topk_can_fuse_filter(n, k_prime, width, k)) { set_error("topk_filtered: sizes outside the fused path (rails_topk_filter_fusable)"); return RAILS_ENOTSUP; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("topk_filtered: no HIP device"); return RAILS_ELAUNCH; }
  const int r = topk(scores, ld, rows, n, k_prime, ids, ids_row_stride, out_scores, out_ids, workspace, workspace_bytes, cu, (hipStream_t)stream,
                     invalid_ids, width, k, nullptr, run_if);
  return r == kOk ? r : fail(r, "topk_filtered");
}

int rails_pack_candidates(const float* scores, const int64_t* ids, int32_t rows, int32_t k_local, int32_t k, int64_t* msg,
                          void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || k_local < 0 || k < 0 || k_local > k) { set_error("pack_candidates: bad size"); return RAILS_EINVAL; }
  if (rows == 0 || k == 0) return RAILS_OK;
  if (!msg || (k_local > 0 && (!scores || !ids))) { set_error("pack_candidates: NULL pointer"); return RAILS_EINVAL; }
  return fail(pack_candidates(scores, ids, rows, k_local, k, msg, (hipStream_t)stream), "pack_candidates");
}

int rails_merge_candidates(const int64_t* gathered, int32_t n_ranks, int32_t rows, int32_t k, int32_t k_out, float* out_scores,
                           int64_t* out_ids, void* stream) {
  g_err[0] = '\0';
  if (n_ranks <= 0 || rows < 0 || k <= 0 || k_out < 0 || (int64_t)k_out > (int64_t)n_ranks * k) { set_error("merge_candidates: bad size"); return RAILS_EINVAL; }
  if (rows == 0 || k_out == 0) return RAILS_OK;
  if (!gathered || !out_scores || !out_ids) { set_error("merge_candidates: NULL pointer"); return RAILS_EINVAL; }
  const int r = merge_candidates(gathered, n_ranks, rows, k, k_out, out_scores, out_ids, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "merge_candidates");
}

int rails_merge_candidates_filtered(const int64_t* gathered, int32_t n_ranks, int32_t rows, int32_t k, int32_t k_prime, const int64_t* invalid_ids,
                                    int32_t width, int32_t k_out, int64_t* out_ids, float* out_scores, void* stream) {
  g_err[0] = '\0';
  if (n_ranks <= 0 || rows < 0 || k <= 0 || k_prime <= 0 || (int64_t)k_prime > (int64_t)n_ranks * k || width < 0 || k_out <= 0 || k_out > k_prime) { set_error("merge_candidates_filtered: bad size"); return RAILS_EINVAL; }
  if (rows == 0) return RAILS_OK;
  if (!gathered || !out_scores || !out_ids || !invalid_ids) { set_error("merge_candidates_filtered: NULL pointer"); return RAILS_EINVAL; }
  const int r = merge_candidates(gathered, n_ranks, rows, k, k_prime, out_scores, out_ids, (hipStream_t)stream, invalid_ids, width, k_out);
  return r == kOk ? r : fail(r, "merge_candidates_filtered");
}

int rails_merge_candidates_verdict(const int64_t* gathered, int32_t n_ranks, int32_t rows, int32_t k, int32_t k_out, float default_eps, float safety,
                                   const float* guard_values, int32_t guard_per_row, float guard_limit, float* state, float* state_host,
                                   void* call_ws, const int64_t* invalid_ids, int32_t width, int32_t f_k, int64_t* out_ids, float* out_scores,
                                   void* stream) {
  g_err[0] = '\0';
  if (n_ranks <= 0 || rows < 0 || k <= 0 || k_out <= 0 || (int64_t)k_out > (int64_t)n_ranks * k || width < 0 || !(default_eps >= 0.0f) || !(safety >= 0.0f)) {
    set_error("merge_candidates_verdict: bad size");
    return RAILS_EINVAL;
  }
  if (rows == 0) return RAILS_OK;
  if (!gathered || !out_scores || !out_ids || !state || !call_ws || (guard_values && (guard_per_row <= 0 || !(guard_limit >= 0.0f))) ||
      (invalid_ids && (f_k <= 0 || f_k > k_out))) {
    set_error("merge_candidates_verdict: bad argument");
    return RAILS_EINVAL;
  }
  MergeVerdict v{default_eps, safety, guard_values, guard_per_row, guard_limit, state, state_host, static_cast<unsigned int*>(call_ws)};
  const int r = merge_candidates(gathered, n_ranks, rows, k, k_out, out_scores, out_ids, (hipStream_t)stream, invalid_ids, width, f_k, &v);
  return r == kOk ? r : fail(r, "merge_candidates_verdict");
}

int rails_abi_version(void) { return RAILS_ABI_VERSION; }

int rails_hash_item_table(uint64_t seed, int64_t first_item, int64_t n_items, int32_t dim, float scale, float* out, void* stream) {
  g_err[0] = '\0';
  if (first_item < 0 || n_items < 0 || dim <= 0) { set_error("hash_item_table: bad size"); return RAILS_EINVAL; }
  if (n_items == 0) return RAILS_OK;
  if (!out) { set_error("hash_item_table: NULL pointer"); return RAILS_EINVAL; }
  return fail(hash_item_table(seed, first_item, n_items, dim, scale, out, (hipStream_t)stream), "hash_item_table");
}

int rails_range_flag_i32(const int32_t* values, int32_t n, int32_t lo, int32_t hi, int32_t* flag, void* stream) {
  g_err[0] = '\0';
  if (n < 0 || (n > 0 && (!values || !flag))) { set_error("range_flag: bad argument"); return RAILS_EINVAL; }
  return fail(range_flag(values, n, lo, hi, flag, (hipStream_t)stream), "range_flag");
}

int rails_rescore_verdict(const float* row_stats, int32_t rows, float default_eps, float safety, const float* guard_values, int64_t guard_count,
                          float guard_limit, float* state, void* stream) {
  g_err[0] = '\0';
  if (rows <= 0 || !row_stats || !state || !(default_eps >= 0.0f) || !(safety >= 0.0f) || guard_count < 0 || (guard_values && !(guard_limit >= 0.0f))) {
    set_error("rescore_verdict: bad argument");
    return RAILS_EINVAL;
  }
  const int r = rescore_verdict(row_stats, rows, default_eps, safety, guard_values, guard_count, guard_limit, state, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "rescore_verdict");
}

int rails_margin_stats(const float* kth_scores, int64_t ld, int32_t col, const float* m_max, const float* err_max, int32_t rows, float* row_stats, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || col < 0 || ld <= col || (rows > 0 && (!kth_scores || !m_max || !err_max || !row_stats))) { set_error("margin_stats: bad argument"); return RAILS_EINVAL; }
  return fail(margin_stats(kth_scores, ld, col, m_max, err_max, rows, row_stats, (hipStream_t)stream), "margin_stats");
}

int rails_mfma_probe_f16(const uint16_t* a, const uint16_t* b, const float* c, float* d, int64_t n, void* stream) {
  g_err[0] = '\0';
  if (n < 0 || (n > 0 && (!a || !b || !c || !d))) { set_error("mfma_probe_f16: bad argument"); return RAILS_EINVAL; }
  return fail(mfma_probe_f16(a, b, c, d, n, (hipStream_t)stream), "mfma_probe_f16");
}

int rails_mfma_probe_f32(const float* a, const float* b, const float* c, float* d, int64_t n, void* stream) {
  g_err[0] = '\0';
  if (n < 0 || (n > 0 && (!a || !b || !c || !d))) { set_error("mfma_probe_f32: bad argument"); return RAILS_EINVAL; }
  return fail(mfma_probe_f32(a, b, c, d, n, (hipStream_t)stream), "mfma_probe_f32");
}

int rails_scalar_probe_f32(const float* x, int64_t n, float* out, void* stream) {
  g_err[0] = '\0';
  if (n < 0 || (n > 0 && (!x || !out))) { set_error("scalar_probe_f32: bad argument"); return RAILS_EINVAL; }
  return fail(scalar_probe(x, n, out, (hipStream_t)stream), "scalar_probe_f32");
}

int rails_rescore_select(const float* exact_scores, int64_t ld, const float* approx_scores, const float* approx_dense, int64_t ld_dense,
                         const int64_t* positions, const int64_t* ids, int64_t n_items, int32_t rows, int32_t n_ranked, int32_t n_cand,
                         int32_t k, float margin_eps, float check_eps, int32_t one_sided, float* out_scores, int64_t* out_ids, int32_t* row_ok,
                         float* row_stats, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || n_ranked <= 0 || n_cand < n_ranked || k <= 0 || k > n_ranked || ld < n_cand) { set_error("rescore_select: bad size"); return RAILS_EINVAL; }
  if (n_items <= 0 || n_items > 0xFFFFFFFFll) { set_error("rescore_select: positions must fit 32 bits (n_items = %lld)", (long long)n_items); return RAILS_EINVAL; }
  if (rows == 0) return RAILS_OK;
  if (!exact_scores || !approx_scores || !positions || !out_scores || !out_ids || (!row_ok && !row_stats) || (n_cand > n_ranked && (!approx_dense || ld_dense < n_items))) {
    set_error("rescore_select: NULL pointer or short stride");
    return RAILS_EINVAL;
  }
  const int r = rescore_select(exact_scores, ld, approx_scores, approx_dense, ld_dense, positions, ids, rows, n_ranked, n_cand, k, margin_eps,
                               check_eps, one_sided != 0, out_scores, out_ids, row_ok, row_stats, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "rescore_select");
}

size_t rails_candidates_workspace_bytes(int32_t rows) { return candidates_workspace_bytes(rows); }

int rails_candidates_select(const float* scores, int64_t ld, int32_t rows, int64_t n, int32_t cap, float lo, float hi, void* workspace,
                            int64_t* out_positions, float* out_approx, int64_t cand_ld, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || n < 0 || cap <= 0 || ld < n) { set_error("candidates_select: bad size"); return RAILS_EINVAL; }
  if (rows == 0 || n == 0) return RAILS_OK;
  if (!scores || !workspace || !out_positions || !out_approx) { set_error("candidates_select: NULL pointer"); return RAILS_EINVAL; }
  const int cu = compute_units();
  if (cu <= 0) { set_error("candidates_select: no HIP device"); return RAILS_ELAUNCH; }
  const int r = candidates_select(scores, ld, rows, n, cap, lo, hi, workspace, out_positions, out_approx, cand_ld, cu, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "candidates_select");
}

int rails_candidates_finish(const float* exact_scores, int64_t ld, const float* approx, const int64_t* positions, int64_t cand_ld, int32_t cap,
                            void* workspace, const int64_t* ids, int64_t n_items, int32_t rows, int32_t k, float default_eps, float safety,
                            int32_t one_sided, const float* guard_values, int32_t guard_per_row, float guard_limit, float* out_scores,
                            int64_t* out_ids, const int64_t* invalid_ids, int32_t width, int32_t f_k, int64_t* f_out_ids, float* f_out_scores,
                            float* state, float* state_host, int64_t* msg, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || cap <= 0 || k <= 0 || ld < cap || cand_ld < cap || n_items <= 0 || n_items > 0xFFFFFFFFll) { set_error("candidates_finish: bad size"); return RAILS_EINVAL; }
  if (rows == 0) return RAILS_OK;
  if (!exact_scores || !approx || !positions || !workspace || (!msg && (!out_scores || !out_ids || !state)) || !(default_eps >= 0.0f) || !(safety >= 0.0f) ||
      (guard_values && (guard_per_row <= 0 || !(guard_limit >= 0.0f))) || (invalid_ids && (msg || !f_out_ids || !f_out_scores))) {
    set_error("candidates_finish: bad argument");
    return RAILS_EINVAL;
  }
  const int r = candidates_finish(exact_scores, ld, approx, positions, cand_ld, cap, workspace, ids, n_items, rows, k, default_eps, safety, one_sided != 0,
                                  guard_values, guard_per_row, guard_limit, out_scores, out_ids, invalid_ids, width, f_k, f_out_ids, f_out_scores, state,
                                  state_host, msg, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "candidates_finish");
}

int rails_filter_seen_ids(const int64_t* top_ids, const float* top_scores, int32_t rows, int32_t k_prime,
                          const int64_t* invalid_ids, int32_t width, int32_t k, int64_t* out_ids, float* out_scores,
                          void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || k_prime < 0 || width < 0 || k < 0) { set_error("filter_seen_ids: negative size"); return RAILS_EINVAL; }
  if (rows == 0 || k == 0) return RAILS_OK;
  if (!top_ids || !top_scores || !out_ids || !out_scores || (width > 0 && !invalid_ids)) {
    set_error("filter_seen_ids: NULL pointer");
    return RAILS_EINVAL;
  }
  const int r = filter_seen(top_ids, top_scores, rows, k_prime, invalid_ids, width, k, out_ids, out_scores,
                            (hipStream_t)stream);
  return r == kOk ? r : fail(r, "filter_seen_ids");
}

// ---- HSTU query encoder, eval path ----
int rails_hstu_preprocess(const float* embeddings, const int64_t* ids, const int64_t* lengths, const float* pos_emb, int32_t batch,
                          int32_t seq_len, int32_t dim, float scale, float* out, void* stream) {
  g_err[0] = '\0';
  if (batch < 0 || seq_len < 0 || dim < 0) { set_error("hstu_preprocess: negative size"); return RAILS_EINVAL; }
  if (batch == 0 || seq_len == 0 || dim == 0) return RAILS_OK;
  if (!embeddings || !ids || !lengths || !pos_emb || !out) { set_error("hstu_preprocess: NULL pointer"); return RAILS_EINVAL; }
  return fail(hstu_preprocess(embeddings, ids, lengths, pos_emb, batch, seq_len, dim, scale, out, (hipStream_t)stream), "hstu_preprocess");
}

int rails_rows_layer_norm(const float* x, int64_t ldx, int64_t rows, int32_t dim, float eps, const float* mul, int64_t ldm, float* out,
                          int64_t ldo, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || dim <= 0) { set_error("rows_layer_norm: bad size"); return RAILS_EINVAL; }
  if (rows == 0) return RAILS_OK;
  if (!x || !out || ldx < dim || ldo < dim || (mul && ldm < dim)) { set_error("rows_layer_norm: NULL pointer or short stride"); return RAILS_EINVAL; }
  return fail(rows_layer_norm(x, ldx, rows, dim, eps, mul, ldm, out, ldo, (hipStream_t)stream), "rows_layer_norm");
}

int rails_gemm_f32(const float* a, int64_t lda, const float* w, int32_t w_is_nk, const float* bias, const float* residual, int64_t ldr,
                   int64_t m, int32_t n, int32_t k, int32_t act, const int64_t* lengths, int32_t seq_len, float* c, int64_t ldc,
                   void* stream) {
  g_err[0] = '\0';
  if (m < 0 || n < 0 || k <= 0) { set_error("gemm_f32: bad size"); return RAILS_EINVAL; }
  if (m == 0 || n == 0) return RAILS_OK;
  if (!a || !w || !c || lda < k || ldc < n || (residual && ldr < n)) { set_error("gemm_f32: NULL pointer or short stride"); return RAILS_EINVAL; }
  if (act != 0 && act != 1) { set_error("gemm_f32: unknown activation %d", act); return RAILS_EINVAL; }
  if (lengths && (seq_len <= 0 || m % seq_len != 0)) { set_error("gemm_f32: rows are not batch * seq_len"); return RAILS_EINVAL; }
  return fail(gemm_f32(a, lda, w, w_is_nk ? 1 : 0, bias, residual, ldr, m, n, k, act, lengths, seq_len, c, ldc, (hipStream_t)stream), "gemm_f32");
}

int rails_mol_gate_combine(const float* logits, int64_t ld_logits, const float* pair_part, int64_t ld_pair, const float* query_part,
                           const float* item_part, int64_t rows, int32_t items_per_query, int32_t num_logits, int32_t item_part_per_row,
                           int32_t combination, int32_t renormalise, float eps, float* out, float* probs_out, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || items_per_query <= 0 || num_logits <= 0) { set_error("gate_combine: bad size"); return RAILS_EINVAL; }
  if (rows == 0) return RAILS_OK;
  if (!logits || !out || ld_logits < num_logits || (pair_part && ld_pair < num_logits) || rows % items_per_query != 0) {
    set_error("gate_combine: NULL pointer, short stride or rows not a multiple of items_per_query");
    return RAILS_EINVAL;
  }
  if (combination != RAILS_COMBINE_GLU_SILU && combination != RAILS_COMBINE_NONE) { set_error("gate_combine: unknown combination %d", combination); return RAILS_EINVAL; }
  if (combination == RAILS_COMBINE_GLU_SILU && (!pair_part || !query_part || !item_part)) { set_error("gate_combine: glu_silu needs all three gate parts"); return RAILS_EINVAL; }
  if (!pair_part && !query_part && !item_part) { set_error("gate_combine: no gate part"); return RAILS_EINVAL; }
  return fail(gate_combine(logits, ld_logits, pair_part, ld_pair, query_part, item_part, rows, items_per_query, num_logits, item_part_per_row ? 1 : 0,
                           combination == RAILS_COMBINE_GLU_SILU ? 1 : 0, renormalise ? 1 : 0, eps, out, probs_out, (hipStream_t)stream), "gate_combine");
}

int rails_glu_f32(const float* x, int64_t ldx, const float* w, const float* b, int64_t rows, int32_t in_features, int32_t out_features,
                  int32_t kind, float* scratch, float* out, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || in_features <= 0 || out_features <= 0) { set_error("glu_f32: bad size"); return RAILS_EINVAL; }
  if (kind != RAILS_GEGLU && kind != RAILS_SWIGLU) { set_error("glu_f32: unknown kind %d", kind); return RAILS_EINVAL; }
  if (rows == 0) return RAILS_OK;
  if (!x || !w || !scratch || !out || ldx < in_features) { set_error("glu_f32: NULL pointer or short stride"); return RAILS_EINVAL; }
  const int n2 = 2 * out_features;
  const int rc = gemm_f32(x, ldx, w, 0, b, nullptr, 0, rows, n2, in_features, 0, nullptr, 0, scratch, n2, (hipStream_t)stream);
  if (rc != kOk) return fail(rc, "glu_f32 (gemm)");
  return fail(glu_gate(scratch, n2, rows, out_features, kind, out, (hipStream_t)stream), "glu_f32 (gate)");
}

int rails_hstu_time_buckets(const int64_t* timestamps, int32_t batch, int32_t seq_len, const int64_t* thresholds, int32_t num_buckets,
                            uint8_t* out, void* stream) {
  g_err[0] = '\0';
  if (batch < 0 || seq_len < 0 || num_buckets <= 0) { set_error("hstu_time_buckets: bad size"); return RAILS_EINVAL; }
  if (batch == 0 || seq_len == 0) return RAILS_OK;
  if (!timestamps || !thresholds || !out) { set_error("hstu_time_buckets: NULL pointer"); return RAILS_EINVAL; }
  const int r = hstu_time_buckets(timestamps, batch, seq_len, thresholds, num_buckets, out, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "hstu_time_buckets");
}

int rails_hstu_attention(const float* uvqk, int64_t ld, int32_t batch, int32_t seq_len, int32_t heads, int32_t dqk, int32_t dv,
                         const int64_t* lengths, const uint8_t* buckets, const float* ts_w, const float* pos_w, int32_t num_buckets,
                         float* out, void* stream) {
  g_err[0] = '\0';
  if (batch < 0 || seq_len < 0 || heads <= 0 || dqk <= 0 || dv <= 0) { set_error("hstu_attention: bad size"); return RAILS_EINVAL; }
  if (batch == 0 || seq_len == 0) return RAILS_OK;
  if (!uvqk || !lengths || !out || ld < (int64_t)2 * heads * (dqk + dv)) { set_error("hstu_attention: NULL pointer or short stride"); return RAILS_EINVAL; }
  if (buckets && (!ts_w || !pos_w || num_buckets <= 0)) { set_error("hstu_attention: buckets without bias tables"); return RAILS_EINVAL; }
  const int r = hstu_attention(uvqk, ld, batch, seq_len, heads, dqk, dv, lengths, buckets, ts_w, pos_w, num_buckets, out, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "hstu_attention");
}

int rails_hstu_fused_supported(int32_t seq_len, int32_t dim, int32_t heads, int32_t dqk, int32_t dv, int32_t num_buckets) {
  return hstu_fused_supported(seq_len, dim, heads, dqk, dv, num_buckets) ? 1 : 0;
}

int rails_hstu_encode_fused(const float* embeddings, const int64_t* ids, const int64_t* lengths, const uint8_t* buckets,
                            const float* pos_emb, const rails_hstu_layer* layers, int32_t n_blocks, int32_t batch, int32_t seq_len,
                            int32_t dim, int32_t heads, int32_t dqk, int32_t dv, int32_t num_buckets, int32_t postproc_mode, float eps,
                            float* out, void* stream) {
  g_err[0] = '\0';
  if (batch < 0 || n_blocks < 0 || (postproc_mode != 0 && postproc_mode != 1)) { set_error("hstu_encode_fused: bad argument"); return RAILS_EINVAL; }
  if (batch == 0) return RAILS_OK;
  if (!embeddings || !ids || !lengths || !pos_emb || !out || (n_blocks > 0 && !layers)) { set_error("hstu_encode_fused: NULL pointer"); return RAILS_EINVAL; }
  const int r = hstu_encode_fused(embeddings, ids, lengths, buckets, pos_emb, layers, n_blocks, batch, seq_len, dim, heads, dqk, dv,
                                  num_buckets, postproc_mode, eps, out, (hipStream_t)stream);
  return r == kOk ? r : fail(r, "hstu_encode_fused");
}

int rails_rows_normalize(const float* x, int64_t ldx, const int64_t* row_index, int64_t rows, int32_t dim, int32_t mode, float eps,
                         float* out, void* stream) {
  g_err[0] = '\0';
  if (rows < 0 || dim <= 0 || (mode != 0 && mode != 1)) { set_error("rows_normalize: bad argument"); return RAILS_EINVAL; }
  if (rows == 0) return RAILS_OK;
  if (!x || !out || ldx < dim) { set_error("rows_normalize: NULL pointer or short stride"); return RAILS_EINVAL; }
  return fail(rows_normalize(x, ldx, row_index, rows, dim, mode, eps, out, (hipStream_t)stream), "rows_normalize");
}

}  // extern "C"
