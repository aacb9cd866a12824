// Launch side of the f16x3 scoring kernels for the shapes with tuned variants (see mol_score_f16_unit.h for the kernel).
#include "mol_score_f16_unit.h"
#include "mol_score_wsplit_f16.h"
#if RAILS_F16_SINGLE   // the one-product build of this file (mol_score_f16x1*.hip)
#define score_launch_f16 score_launch_f16x1
#endif

#ifndef RAILS_F16_TIGHT_LIMIT
#define RAILS_F16_TIGHT_LIMIT 200   // accumulator registers of a unit above which the 8-wave build takes the TIGHT stream
#endif

namespace mol {

// RAILS_F16_OVERLAP=0 disables the cross-query overlap of stage X (measurement override)
static bool f16_overlap() {
  const char* e = getenv("RAILS_F16_OVERLAP");
  return e ? atoi(e) != 0 : true;
}

template <bool OVL, int PQ, int PX, int DD, int H>
static int launch_f16(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using G = Geo<PQ, PX, DD, H>;
  // two waves per SIMD leave 256 registers per lane: when the accumulators of a unit (D1, D2, D3) take most of them, the
  // untied stream without cross-query overlap is the one that does not spill
  constexpr bool tight = (PX + G::TH + G::TL) * 16 > RAILS_F16_TIGHT_LIMIT;
  using U8 = std::conditional_t<tight, F16Unit<false, true>, F16Unit<OVL, false>>;
  using U4 = F16Unit<OVL, false>;
  if (a.combine_none && !a.upper)   // gating_combination "none": its own instantiation of the unit, direct shell only
    return launch_kernel<F16Unit<false, tight, true>, PQ, PX, DD, H, 8, false>(a, n_cu, stream);
  const int variant = choose_variant<PQ, PX, DD, H>(a, n_cu);
#if !RAILS_F16_SINGLE
  if (a.upper) {   // rails_mol_score_dense_upper: the 8-wave builds of the unit with the per-pair bound added to the logit
    using UP = std::conditional_t<tight, F16Unit<false, true, false, true>, F16Unit<true, false, false, true>>;
    if (a.combine_none || a.per_row) { set_error("the upper-bound first pass is built for the glu_silu combiner over a shared corpus"); return kErrUnsupported; }
    switch (variant) {
      case 1: return launch_kernel<UP, PQ, PX, DD, H, 8, false>(a, n_cu, stream);
      case 2: return launch_kernel<UP, PQ, PX, DD, H, 8, true>(a, n_cu, stream);
      case 5: return launch_staged1<UP, PQ, PX, DD, H, 8>(a, n_cu, stream);
      default: set_error("the upper-bound first pass has no 4-wave build (RAILS_SCORE_VARIANT %d)", variant); return kErrUnsupported;
    }
  }
#endif
  if ((variant == 2 || variant == 4 || variant == 5 || variant == 6) && a.per_row) { set_error("staged scoring kernel does not do per-row candidates"); return kErrUnsupported; }
  switch (variant) {
    case 1: return launch_kernel<U8, PQ, PX, DD, H, 8, false>(a, n_cu, stream);
    case 2: return launch_kernel<U8, PQ, PX, DD, H, 8, true>(a, n_cu, stream);
    case 3: return launch_kernel<U4, PQ, PX, DD, H, 4, false>(a, n_cu, stream);
    case 4: return launch_kernel<U4, PQ, PX, DD, H, 4, true>(a, n_cu, stream);
    case 5: return launch_staged1<U8, PQ, PX, DD, H, 8>(a, n_cu, stream);
    case 6: return launch_staged1<U4, PQ, PX, DD, H, 4>(a, n_cu, stream);
    default: set_error("unknown RAILS_SCORE_VARIANT %d", variant); return kErrInvalid;
  }
}

int score_launch_f16(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream) {
#define MOL_CASE(pq, px, dd)                                                                                 \
  if (s.query_dot_product_groups == pq && s.item_dot_product_groups == px && s.dot_product_dimension == dd)   \
    return f16_overlap() ? launch_f16<true, pq, px, dd, 128>(a, n_cu, stream) : launch_f16<false, pq, px, dd, 128>(a, n_cu, stream);
  MOL_CASE(8, 4, 64)
  MOL_CASE(8, 4, 128)
  MOL_CASE(8, 8, 32)
#undef MOL_CASE
  if (s.query_dot_product_groups == 16 && s.item_dot_product_groups == 16 && s.dot_product_dimension == 64) {
    // L = 256: tiles (160 KiB) and the gate pack (256 KiB) are beyond LDS staging -- the team kernel (mol_score_wsplit.h)
#if !RAILS_F16_SINGLE
    if (a.upper) {
      if (a.combine_none || a.per_row) { set_error("the upper-bound first pass is built for the glu_silu combiner over a shared corpus"); return kErrUnsupported; }
      return launch_wsplit<WsF16, 16, 16, 64, 128, true>(a, n_cu, stream);
    }
#endif
    return a.combine_none ? launch_wsplit<WsF16T<true>, 16, 16, 64, 128>(a, n_cu, stream) : launch_wsplit<WsF16, 16, 16, 64, 128>(a, n_cu, stream);
  }
  set_error("the f16x3 precision mode is not built for this shape");
  return kErrUnsupported;
}

#ifdef RAILS_WS_PHASES
}  // namespace mol
#if RAILS_F16_SINGLE
extern "C" int rails_debug_ws_phases_f16x1(long long* out) {
#else
extern "C" int rails_debug_ws_phases_f16x3(long long* out) {
#endif
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mol::g_ws_phase), sizeof(long long) * 16) == hipSuccess ? 0 : -1;
}
namespace mol {
#endif
#ifdef RAILS_F16_PHASES
}  // namespace mol
extern "C" int rails_debug_f16_phases(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mol::g_f16_phase), sizeof(long long) * 32) == hipSuccess ? 0 : -1;
}
namespace mol {
#endif
}  // namespace mol
