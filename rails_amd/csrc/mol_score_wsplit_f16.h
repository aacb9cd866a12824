// f16 policies of the team kernel for L = 256 (mol_score_wsplit.h): precision f16x3 (every operand f16 hi + f16 lo, three
// v_mfma_f32_32x32x16_f16 per product block, mol_score_f16_unit.h) and, built with RAILS_F16_SINGLE, the one-product first pass
// of "f16-exact".  An operand chunk is one K = 16 step: 8 accumulator registers of a lane -> h8 hi (+ h8 lo).  Same packed
// buffers as the register-resident f16 kernels: gate pack [W1 hi][W1 lo][W2 hi][W2 lo][b1][b2], tiles and query pack with
// hi/lo fragment pairs in the bytes of the fp32 fragments they replace (mol_layout.h).
#pragma once
#include "mol_score_f16_unit.h"
#include "mol_score_wsplit.h"

namespace mol {
#if RAILS_F16_SINGLE
inline namespace f16x1 {
#else
inline namespace f16x3 {
#endif

// NONE: gating_combination "none" as a compile-time switch (a run-time branch in every slice cost the glu_silu kernels ~7 %)
template <bool NONE>
struct WsF16T {
  static constexpr int CE = 8;
  static constexpr int OPV = RAILS_F16_SINGLE ? 1 : 2;
#ifndef RAILS_WS16_W1STREAM
#define RAILS_WS16_W1STREAM 4
#endif
  static constexpr int kW1Stream = RAILS_F16_SINGLE ? 0 : RAILS_WS16_W1STREAM;   // f16x3: 256 registers of weights; 32 of them re-read per unit
  // prefetch distances in chunks.  A chunk of f16x3 is 12 (GEMM1) / 6 MFMAs = 384 / 192 cycles, of the one-product build a third
  // of that; L2 latency ~ 800 cycles, LDS ~ 130-200.
#if RAILS_F16_SINGLE
  static constexpr int PD1 = 3, PD2 = 4, PD3 = 4;   // (PD1 = 4, all of GEMM1 a phase ahead, measured slower: 20 buffer loads cost ~1 k cycles of issue in the gate pass)
#else
#ifndef RAILS_WS16_PD1
#define RAILS_WS16_PD1 1   // 2 spills (26 registers over with 256 of weights)
#endif
  static constexpr int PD1 = RAILS_WS16_PD1, PD2 = 2, PD3 = 2;
#endif
  struct Op {
    h8 hi;
#if !RAILS_F16_SINGLE
    h8 lo;
#endif
  };
  float m1;   // -1.0, opaque to the compiler (split_pair)
  __device__ __forceinline__ void init() {
    m1 = -1.0f;
    asm volatile("" : "+v"(m1));
  }
  static __device__ __forceinline__ void pin(Op& o) {
    asm volatile("" : "+a"(o.hi));
#if !RAILS_F16_SINGLE
    asm volatile("" : "+a"(o.lo));
#endif
  }
  static __device__ __forceinline__ Op ld2(const WsBuf& b, int idx_hi, int idx_lo, int lane16) {
    Op o;
    o.hi = __builtin_bit_cast(h8, b.frag(idx_hi, lane16));
#if !RAILS_F16_SINGLE
    o.lo = __builtin_bit_cast(h8, b.frag(idx_lo, lane16));
#endif
    return o;
  }
  template <class G, int DD> static __device__ __forceinline__ Op eq_op(const WsBuf& b, int c, int lane16) { return ld2(b, 2 * c, 2 * c + 1, lane16); }
  template <class G, int DD> static __device__ __forceinline__ Op ex_op(const WsBuf& b, int m, int c, int lane16) {
    return ld2(b, m * (DD / 8) + 2 * c, m * (DD / 8) + 2 * c + 1, lane16);
  }
  // 1 KiB fragments per half (hi or lo) of a weight matrix
  template <class G> static constexpr int kN64 = G::kW1Floats / 8 / 64;
  template <class G> static __device__ __forceinline__ Op w1_op(const WsBuf& b, int c, int t, int lane16) {
    return ld2(b, c * G::TH + t, kN64<G> + c * G::TH + t, lane16);
  }
  template <class G> static __device__ __forceinline__ Op w2_op(const WsBuf& b, int c, int v, int lane16) {
    return ld2(b, 2 * kN64<G> + c * G::TL + v, 3 * kN64<G> + c * G::TL + v, lane16);
  }
  // products outermost: consecutive MFMAs go to different accumulators (dependent-accumulate latency = 2 x issue time)
  template <int N>
  static __device__ __forceinline__ void mma_a(f32x16 (&d)[N], const Op& a, const Op (&b)[N]) {
#if !RAILS_F16_SINGLE
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma16(a.lo, b[n].hi, d[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma16(a.hi, b[n].lo, d[n]);
#endif
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma16(a.hi, b[n].hi, d[n]);
  }
  template <int N>
  static __device__ __forceinline__ void mma_b(f32x16 (&d)[N], const Op (&a)[N], const Op& b) {
#if !RAILS_F16_SINGLE
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma16(a[n].lo, b.hi, d[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma16(a[n].hi, b.lo, d[n]);
#endif
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = mfma16(a[n].hi, b.hi, d[n]);
  }
  template <int R0>
  __device__ __forceinline__ Op pack(const f32x16& acc) const {
    float xs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xs[j] = acc[R0 + j];
    Op o;
    h8 lo;
    split8(xs, m1, o.hi, lo);
#if !RAILS_F16_SINGLE
    o.lo = lo;
#endif
    return o;
  }
  static __device__ __forceinline__ void st(float4* slot, int lane, const Op& o) {
    slot[lane] = __builtin_bit_cast(float4, o.hi);
#if !RAILS_F16_SINGLE
    slot[64 + lane] = __builtin_bit_cast(float4, o.lo);
#endif
  }
  static __device__ __forceinline__ Op ldl(const float4* slot, int lane) {
    Op o;
    o.hi = __builtin_bit_cast(h8, slot[lane]);
#if !RAILS_F16_SINGLE
    o.lo = __builtin_bit_cast(h8, slot[64 + lane]);
#endif
    return o;
  }
  // t / (1 + 2^t), stage by stage over groups of eight values (dependent issues are the slow ones at one wave per SIMD)
  static __device__ __forceinline__ void silu16(f32x16& d) {
#pragma unroll
    for (int g = 0; g < 16; g += 8) {
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = f_exp2(d[g + j]);
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += 1.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = f_rcp(r[j]);
#pragma unroll
      for (int j = 0; j < 8; ++j) d[g + j] *= r[j];
    }
  }
  // Gate, softmax numerators and mixture of ONE query over this wave's EW logit K-steps, scalar fp32 (packed fp32 stalls the f16
  // matrix pipe, mol_score_f16_unit.h), in EW/2 slices of one logit pair that the shell deals between the MFMAs of the other
  // query's GEMM3:  u = t2 / (1 + 2^t2), t2 = -log2e * (gq*gi + gqi);  ex = 2^(-u) WITHOUT the usual shift (u <= 0.402, so
  // ex >= 0.757: no underflow; an overflow shows on the wave's sums and that wave redoes its slice in the shifted form) -> the
  // wave's (shift, sum ex, sum ex * cl); the team fold in the shell handles mixed shifts.  Sums go to NA partial accumulators
  // (a single dependent chain of 32 adds / 64 fmas is latency-bound at one wave per SIMD).
  // cl: the one-product build keeps D1w in registers (its weights take 128 of them); f16x3 re-reads the wave's own hi/lo
  // chunks from LDS -- cl = hi + lo to 2^-22, what the gate network saw -- instead of keeping 64 registers alive next to 256.
  template <class G, int MW, int TLW, int EW>
  struct Epi {
    static constexpr int NS = EW / 4, NA = 4;
    static constexpr int PF = 2;   // gq / gi float4 (one per slice) requested PF slices ahead; f16x3: the cl hi/lo chunk (one per two slices) one ahead
    float den[NA], num[NA];
    float4 gqr[PF + 1], gir[PF + 1];
#if !RAILS_F16_SINGLE
    float4 chr[2], clr[2];
#endif
    __device__ __forceinline__ void begin(const float4* cl_lds, const float4* gi_lds, const float4* gq4) {
#pragma unroll
      for (int i = 0; i < NA; ++i) den[i] = num[i] = 0.0f;
#pragma unroll
      for (int i = 0; i < PF; ++i) { gqr[i] = gq4[i]; gir[i] = gi_lds[i * 64]; }
#if !RAILS_F16_SINGLE
      chr[0] = cl_lds[0];
      clr[0] = cl_lds[64];
#endif
    }
    // One slice = the four logits of one gi / gq float4, evaluated STAGE BY STAGE across the four (all fmas, all exps, ...): the
    // chain of one logit is seven dependent instructions, three of them transcendental, and at one wave per SIMD a dependent
    // issue costs ~9 cycles against ~5 for an independent one.
    template <int Q, int S>
    __device__ __forceinline__ void slice(f32x16 (&D3)[TLW], const f32x16 (&D1w)[MW], const float4* cl_lds, const float4* gi_lds /* [ec * 64] */,
                                          const float4* gq4, int combine_none) {
      constexpr int ec = S, e0 = 4 * S;
      if constexpr (ec + PF < EW / 4) { gqr[(ec + PF) % (PF + 1)] = gq4[ec + PF]; gir[(ec + PF) % (PF + 1)] = gi_lds[(ec + PF) * 64]; }
      const float4 gq = gqr[ec % (PF + 1)], gi = gir[ec % (PF + 1)];
      const float giv[4] = {gi.x, gi.y, gi.z, gi.w}, gqv[4] = {gq.x, gq.y, gq.z, gq.w};
#if !RAILS_F16_SINGLE
      constexpr int k = e0 / 8;
      if constexpr (e0 % 8 == 0 && k + 1 < EW / 8) { chr[(k + 1) & 1] = cl_lds[(2 * (k + 1)) * 64]; clr[(k + 1) & 1] = cl_lds[(2 * (k + 1) + 1) * 64]; }
      const h8 ch = __builtin_bit_cast(h8, chr[k & 1]), cl = __builtin_bit_cast(h8, clr[k & 1]);
#endif
      float t[4], r[4], ex[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = __builtin_fmaf(gqv[j], giv[j], D3[(e0 + j) / 16][(e0 + j) % 16]);
      if constexpr (NONE) {   // gating_combination "none": u = -log2e * (gq + gi + gqi); gq and gqi arrive prescaled, gi does not
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = __builtin_fmaf(giv[j], -kLog2e, gqv[j] + D3[(e0 + j) / 16][(e0 + j) % 16]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = f_exp2(t[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] += 1.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = f_rcp(r[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] *= r[j];   // u = t2 / (1 + 2^t2)
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) ex[j] = f_exp2(-t[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        D3[(e0 + j) / 16][(e0 + j) % 16] = t[j];   // kept for the shifted redo
        den[j] += ex[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j;
#if RAILS_F16_SINGLE
        num[j] = __builtin_fmaf(ex[j], D1w[e / G::RPQ][Q * G::RPQ + e % G::RPQ], num[j]);
#else
        num[j] = __builtin_fmaf(ex[j], (float)ch[e % 8], num[j]);   // v_fma_mix_f32
#endif
      }
#if !RAILS_F16_SINGLE
#pragma unroll
      for (int j = 0; j < 4; ++j) num[j] = __builtin_fmaf(ex[j], (float)cl[(e0 + j) % 8], num[j]);
#endif
    }
    template <int Q>
    __device__ __forceinline__ void end(f32x16 (&D3)[TLW], const f32x16 (&D1w)[MW], const float4* cl_lds, float& mn_out, float& den_out, float& num_out) {
      float dn = (den[0] + den[1]) + (den[2] + den[3]), nm = (num[0] + num[1]) + (num[2] + num[3]);
      dn += swap32(dn);
      nm += swap32(nm);
      float mn = 0.0f;
      // guard well below FLT_MAX: see epi_final (mol_score_f16_unit.h)
      // an exp got large somewhere in this wave (or, with "none", all of them tiny): the stable form
      if (__builtin_amdgcn_ballot_w64(NONE ? !(dn < 1.0e30f && dn > 1.0e-30f) : !(dn < 1.0e30f)) != 0) {
        mn = INFINITY;
#pragma unroll
        for (int e = 0; e < EW; ++e) mn = __builtin_fminf(mn, D3[e / 16][e % 16]);
        mn = __builtin_fminf(mn, swap32(mn));
        dn = 0.0f;
        nm = 0.0f;
#pragma unroll
        for (int e = 0; e < EW; ++e) {
          const float ex = f_exp2(mn - D3[e / 16][e % 16]);
          dn += ex;
#if RAILS_F16_SINGLE
          nm = __builtin_fmaf(ex, D1w[e / G::RPQ][Q * G::RPQ + e % G::RPQ], nm);
#else
          const h8 ch = __builtin_bit_cast(h8, cl_lds[(2 * (e / 8)) * 64]), cl = __builtin_bit_cast(h8, cl_lds[(2 * (e / 8) + 1) * 64]);
          nm = __builtin_fmaf(ex, (float)ch[e % 8] + (float)cl[e % 8], nm);
#endif
        }
        dn += swap32(dn);
        nm += swap32(nm);
      }
      mn_out = mn;
      den_out = dn;
      num_out = nm;
    }
  };
};
using WsF16 = WsF16T<false>;

}  // inline namespace f16x3 / f16x1
}  // namespace mol
