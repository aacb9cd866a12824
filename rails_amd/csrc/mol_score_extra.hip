// Exact-fp32 scoring kernels for the shapes of mol_score_extra_shapes.h (direct shell only).
#include "mol_score_extra_shapes.h"
#include "mol_score_fp32_unit.h"

namespace mol {

bool score_extra_shape(const Shape& s) {
#define X(pq, px, dd, h) \
  if (s.query_dot_product_groups == pq && s.item_dot_product_groups == px && s.dot_product_dimension == dd && s.gating_qi_hidden_dim == h) return true;
  MOL_EXTRA_SHAPES(X)
#undef X
  if (s.gating_qi_hidden_dim <= 0 && s.precision == RAILS_PRECISION_FP32) {
#define X(pq, px, dd) \
  if (s.query_dot_product_groups == pq && s.item_dot_product_groups == px && s.dot_product_dimension == dd) return true;
    MOL_NOHID_SHAPES(X)
#undef X
  }
  return false;
}

int score_launch_extra(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream) {
#define X(pq, px, dd, h)                                                                                                             \
  if (s.query_dot_product_groups == pq && s.item_dot_product_groups == px && s.dot_product_dimension == dd && s.gating_qi_hidden_dim == h) \
    return launch_kernel<Fp32Unit, pq, px, dd, h, 8, false>(a, n_cu, stream);
  MOL_EXTRA_SHAPES(X)
#undef X
  if (s.gating_qi_hidden_dim <= 0) {
#define X(pq, px, dd)                                                                                    \
  if (s.query_dot_product_groups == pq && s.item_dot_product_groups == px && s.dot_product_dimension == dd) \
    return launch_kernel<Fp32Unit, pq, px, dd, 0, 8, false>(a, n_cu, stream);
    MOL_NOHID_SHAPES(X)
#undef X
  }
  return kErrUnsupported;
}

}  // namespace mol
