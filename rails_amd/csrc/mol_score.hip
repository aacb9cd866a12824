// Launch side of the exact-fp32 scoring kernels (the unit arithmetic is in mol_score_fp32_unit.h) and the exact-fp32
// policy of the team kernel for L = 256 (mol_score_wsplit.h).
#include "mol_score_fp32_unit.h"
#include "mol_score_wsplit.h"

namespace mol {

// ---------------------------------------------------------------------------------------------
// Exact-fp32 policy of the team kernel for L = 256 (mol_score_wsplit.h).  An operand chunk is one float4 = four K = 2
// steps of v_mfma_f32_32x32x2_f32; the gate pack and the item tiles are read in the fp32 fragment order the register-resident
// kernels use (mol_layout.h), so all buffers are shared with them.
// ---------------------------------------------------------------------------------------------
struct WsFp32 {
  static constexpr int CE = 4, OPV = 1;
  static constexpr int kW1Stream = 0;
#ifndef RAILS_WS_PD1
#define RAILS_WS_PD1 2
#endif
  static constexpr int PD1 = RAILS_WS_PD1;  // GEMM1 chunks (16 MFMAs = 1024 cycles each) requested ahead: L2 latency with the tile touched a unit earlier
  static constexpr int PD2 = 1, PD3 = 1;    // a chunk is 8 MFMAs = 512 cycles: one chunk ahead covers the LDS latency
  struct Op { float4 v; };
  __device__ __forceinline__ void init() {}
  static __device__ __forceinline__ void pin(Op& o) { asm volatile("" : "+a"(o.v.x), "+a"(o.v.y), "+a"(o.v.z), "+a"(o.v.w)); }
  static __device__ __forceinline__ Op ld(const WsBuf& b, int idx, int lane16) {
    const ws_u32x4 v = b.frag(idx, lane16);
    return Op{make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w))};
  }
  template <class G, int DD> static __device__ __forceinline__ Op eq_op(const WsBuf& b, int c, int lane16) { return ld(b, c, lane16); }
  template <class G, int DD> static __device__ __forceinline__ Op ex_op(const WsBuf& b, int m, int c, int lane16) { return ld(b, m * (DD / 8) + c, lane16); }
  template <class G> static __device__ __forceinline__ Op w1_op(const WsBuf& b, int c, int t, int lane16) { return ld(b, c * G::TH + t, lane16); }
  template <class G> static __device__ __forceinline__ Op w2_op(const WsBuf& b, int c, int v, int lane16) { return ld(b, G::kW1Floats / 256 + c * G::TL + v, lane16); }
  template <int N>
  static __device__ __forceinline__ void mma_a(f32x16 (&d)[N], const Op& a, const Op (&b)[N]) {
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma32(a.v.x, b[n].v.x, d[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma32(a.v.y, b[n].v.y, d[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma32(a.v.z, b[n].v.z, d[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma32(a.v.w, b[n].v.w, d[n]);
  }
  template <int N>
  static __device__ __forceinline__ void mma_b(f32x16 (&d)[N], const Op (&a)[N], const Op& b) {
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma32(a[n].v.x, b.v.x, d[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma32(a[n].v.y, b.v.y, d[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma32(a[n].v.z, b.v.z, d[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma32(a[n].v.w, b.v.w, d[n]);
  }
  template <int R0>
  __device__ __forceinline__ Op pack(const f32x16& acc) const { return Op{make_float4(acc[R0], acc[R0 + 1], acc[R0 + 2], acc[R0 + 3])}; }
  static __device__ __forceinline__ void st(float4* slot, int lane, const Op& o) { slot[lane] = o.v; }
  static __device__ __forceinline__ Op ldl(const float4* slot, int lane) { return Op{slot[lane]}; }
  // t / (1 + 2^t) on the -log2e-prescaled argument, in place
  static __device__ __forceinline__ void silu16(f32x16& d) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 tv = {d[r], d[r + 1]};
      const f32x2 h = tv * pk_sigmoid_arg(tv);
      d[r] = h.x;
      d[r + 1] = h.y;
    }
  }
  // Gate, softmax numerators and mixture of ONE query over this wave's EW logit K-steps, in slices the shell deals between
  // the MFMA chunks of the other query's GEMM3 (fp32 MFMA and VALU share the ALUs, but the dependency stalls of a lone VALU
  // stream at one wave per SIMD disappear):
  //   slices 0 .. EW/4-1   t2 = -log2e * (gq*gi + gqi);  u = t2 / (1 + 2^t2) = -log2e * g*sigmoid(g);  running min
  //                        gating_combination "none" (similarity_fn.py:187-197): u = gq' + gqi' - log2e * gi
  //   slice  EW/4          min over both lane halves
  //   slices EW/4+1 ..     ex = 2^(min u - u); den += ex; num += ex * cl   (cl re-read from the wave's own chunks in LDS: the exact
  //                        fp32 values it wrote there in phase 1; keeping D1w through phases 2 and 3 cost 64 registers next to 256 of weights)
  //   end                  the wave's (min u, sum ex, sum ex * cl), identical in both lane halves
  template <class G, int MW, int TLW, int EW>
  struct Epi {
    static constexpr int NS = EW / 4 + 1 + EW / 4;
    static constexpr int PF = 2;   // LDS operands (gq, gi, cl) are requested PF slices ahead of their use: a slice that waits for its own reads
                                   // pays the LDS latency every time (no second wave on the SIMD to cover it)
    float mn;
    f32x2 den2, num2;
    float4 gqr[PF + 1], gir[PF + 1], clr[PF + 1];
    __device__ __forceinline__ void begin(const float4* cl_lds, const float4* gi_lds, const float4* gq4) {
      mn = INFINITY; den2 = f32x2{0.0f, 0.0f}; num2 = f32x2{0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < PF; ++i) { gqr[i] = gq4[i]; gir[i] = gi_lds[i * 64]; }
    }
    template <int Q, int S>
    __device__ __forceinline__ void slice(f32x16 (&D3)[TLW], const f32x16 (&D1w)[MW], const float4* cl_lds /* [c * 64] */, const float4* gi_lds /* [ec * 64] */,
                                          const float4* gq4, int combine_none) {
      if constexpr (S < EW / 4) {
        constexpr int ec = S;
        if constexpr (ec + PF < EW / 4) { gqr[(ec + PF) % (PF + 1)] = gq4[ec + PF]; gir[(ec + PF) % (PF + 1)] = gi_lds[(ec + PF) * 64]; }
        else if constexpr (ec + PF - EW / 4 < PF) clr[ec + PF - EW / 4] = cl_lds[(ec + PF - EW / 4) * 64];   // the first cl chunks, for pass 2
        const float4 gq = gqr[ec % (PF + 1)], gi = gir[ec % (PF + 1)];
        if (combine_none) {
          const float giv[4] = {gi.x, gi.y, gi.z, gi.w}, gqv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = ec * 4 + j;
            const float u = __builtin_fmaf(giv[j], -kLog2e, gqv[j] + D3[e / 16][e % 16]);
            D3[e / 16][e % 16] = u;
            mn = fminf(mn, u);
          }
        } else {
          const f32x2 giv[2] = {{gi.x, gi.y}, {gi.z, gi.w}};
          const f32x2 gqv[2] = {{gq.x, gq.y}, {gq.z, gq.w}};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int e = ec * 4 + 2 * j;
            const f32x2 t2 = pk_fma(gqv[j], giv[j], f32x2{D3[e / 16][e % 16], D3[e / 16][e % 16 + 1]});
            const f32x2 uu = t2 * pk_sigmoid_arg(t2);
            D3[e / 16][e % 16] = uu.x;
            D3[e / 16][e % 16 + 1] = uu.y;
            mn = fminf(mn, fminf(uu.x, uu.y));
          }
        }
      } else if constexpr (S == EW / 4) {
        mn = fminf(mn, xor32(mn));
      } else {
        constexpr int c = S - EW / 4 - 1;
        if constexpr (c + PF < EW / 4) clr[(c + PF) % (PF + 1)] = cl_lds[(c + PF) * 64];
        const float4 cl = clr[c % (PF + 1)];
        const f32x2 clv[2] = {{cl.x, cl.y}, {cl.z, cl.w}};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int e = 4 * c + 2 * j;
          const f32x2 d = mn - f32x2{D3[e / 16][e % 16], D3[e / 16][e % 16 + 1]};
          const f32x2 ex = {__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
          den2 = den2 + ex;
          num2 = pk_fma(ex, clv[j], num2);
        }
      }
    }
    template <int Q>
    __device__ __forceinline__ void end(f32x16 (&D3)[TLW], const f32x16 (&D1w)[MW], const float4* cl_lds, float& mn_out, float& den_out, float& num_out) {
      float den = den2.x + den2.y, num = num2.x + num2.y;
      den += xor32(den);
      num += xor32(num);
      mn_out = mn;
      den_out = den;
      num_out = num;
    }
  };
};

template <int PQ, int PX, int DD, int H>
static int launch_score(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using U = Fp32Unit;
  const int variant = choose_variant<PQ, PX, DD, H>(a, n_cu);
  if ((variant == 2 || variant == 4 || variant == 5) && a.per_row) { set_error("staged scoring kernel does not do per-row candidates"); return kErrUnsupported; }
  switch (variant) {
    case 1: return launch_kernel<U, PQ, PX, DD, H, 8, false>(a, n_cu, stream);
    case 2: return launch_kernel<U, PQ, PX, DD, H, 8, true>(a, n_cu, stream);
    case 3: return launch_kernel<U, PQ, PX, DD, H, 4, false>(a, n_cu, stream);
    case 4: return launch_kernel<U, PQ, PX, DD, H, 4, true>(a, n_cu, stream);
    case 5: return launch_staged1<U, PQ, PX, DD, H, 8>(a, n_cu, stream);
    default: set_error("unknown RAILS_SCORE_VARIANT %d", variant); return kErrInvalid;
  }
}

bool score_supported(const Shape& s) {
  if (score_extra_shape(s)) return true;
  if (s.gating_qi_hidden_dim != 128) return false;
  const int pq = s.query_dot_product_groups, px = s.item_dot_product_groups, dd = s.dot_product_dimension;
  return (pq == 8 && px == 4 && dd == 64) || (pq == 8 && px == 4 && dd == 128) || (pq == 8 && px == 8 && dd == 32) ||
         (pq == 16 && px == 16 && dd == 64);
}

// Small-unit shell (mol_score_small.hip) or the 32x32x2 shells?  Same bits either way; RAILS_SCORE_VARIANT=7 forces the small
// units, any other non-zero value keeps them off.
static bool use_small_units(const Shape& s, const ScoreArgs& a, int n_cu) {
  if (!score_small_shape(s) || a.per_row || a.cand_pos || a.split) return false;
  const int variant = score_variant();
  if (variant != 0) return variant == 7;
  // Measured on MI355X (profiles/r04_small_units.txt).  The 32x32x2 shells are faster per flop once the chip is full (0.84 against
  // 0.76 of the fp32 MFMA peak on amzn-books at B = 32: half the VALU instructions per pair and a quarter of the L1 traffic per
  // flop), so the small units are for launches that cannot fill it with 32x32x2 units or that stream a large index for one or two
  // queries:
  //   (a) fewer big units than two per CU -- ML-1M at B = 4 ... 16: 26-27 us -> 17-18 us (B = 32 is 976 units: 29 us either way);
  //   (b) B <= 2 over >= 8 tiles per CU-round -- amzn-books B = 1: 0.317-0.331 -> 0.282-0.292 ms, B = 2: 0.483-0.508 -> 0.464-0.482 ms
  //       (half of GEMM1's padding rows gone, four waves per SIMD keep more of the index in flight); ML-20M's 853 tiles stay on
  //       the 32x32x2 shell (18 vs 19 us).
  const int64_t big_units = a.n_tiles * a.n_groups;
  if (big_units <= 2 * (int64_t)n_cu) return true;
  if (a.B <= 2 && a.n_tiles >= 8 * (int64_t)n_cu) return true;
  return false;
}

int score_launch(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream) {
  if (!score_supported(s)) return kErrUnsupported;
  if (a.upper && !(a.split && !a.single && !score_extra_shape(s))) {
    set_error("the upper-bound first pass is built for the f16x3 kernels of the BASELINE shapes");
    return kErrUnsupported;
  }
  if (score_extra_shape(s))
    return !a.split ? score_launch_extra(s, a, n_cu, stream) : a.single ? score_launch_f16x1_extra(s, a, n_cu, stream) : score_launch_f16_extra(s, a, n_cu, stream);
  if (a.split) return a.single ? score_launch_f16x1(s, a, n_cu, stream) : score_launch_f16(s, a, n_cu, stream);
  if (use_small_units(s, a, n_cu)) return score_launch_small(s, a, n_cu, stream);
#define MOL_CASE(pq, px, dd)                                                                             \
  if (s.query_dot_product_groups == pq && s.item_dot_product_groups == px && s.dot_product_dimension == dd) \
    return launch_score<pq, px, dd, 128>(a, n_cu, stream);
  MOL_CASE(8, 4, 64)
  MOL_CASE(8, 4, 128)
  MOL_CASE(8, 8, 32)
#undef MOL_CASE
  if (s.query_dot_product_groups == 16 && s.item_dot_product_groups == 16 && s.dot_product_dimension == 64) {
    return launch_wsplit<WsFp32, 16, 16, 64, 128>(a, n_cu, stream);
  }
  return kErrUnsupported;
}

#ifdef RAILS_WS_PHASES
}  // namespace mol
extern "C" int rails_debug_ws_phases(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mol::g_ws_phase), sizeof(long long) * 16) == hipSuccess ? 0 : -1;
}
namespace mol {
#endif
}  // namespace mol
