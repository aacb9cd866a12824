// f16x3 scoring kernels for the shapes of mol_score_extra_shapes.h (direct shell only).
#include "mol_score_extra_shapes.h"
#include "mol_score_f16_unit.h"
#if RAILS_F16_SINGLE   // the one-product build of this file (mol_score_f16x1*.hip)
#define score_launch_f16_extra score_launch_f16x1_extra
#endif

#ifndef RAILS_F16_TIGHT_LIMIT
#define RAILS_F16_TIGHT_LIMIT 200   // accumulator registers of a unit above which the 8-wave build takes the TIGHT stream
#endif

namespace mol {

template <int PQ, int PX, int DD, int H>
static int launch_f16_extra(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using G = Geo<PQ, PX, DD, H>;
  constexpr bool tight = (PX + G::TH + G::TL) * 16 > RAILS_F16_TIGHT_LIMIT;   // as in mol_score_f16.hip
  using U = std::conditional_t<tight, F16Unit<false, true>, F16Unit<true, false>>;
  if (a.combine_none) return launch_kernel<F16Unit<false, tight, true>, PQ, PX, DD, H, 8, false>(a, n_cu, stream);
  return launch_kernel<U, PQ, PX, DD, H, 8, false>(a, n_cu, stream);
}

int score_launch_f16_extra(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream) {
#define X(pq, px, dd, h)                                                                                                             \
  if (s.query_dot_product_groups == pq && s.item_dot_product_groups == px && s.dot_product_dimension == dd && s.gating_qi_hidden_dim == h) \
    return launch_f16_extra<pq, px, dd, h>(a, n_cu, stream);
  MOL_EXTRA_SHAPES(X)
#undef X
  return kErrUnsupported;
}

}  // namespace mol
