// Fused MoL scoring, precision mode "f16x3": the three contractions of a (query, item) pair
//
//   cl = <Eq, Ex>/tau  ->  hid = silu(W1 cl + b1)  ->  gqi = W2 hid + b2          (similarity_fn.py:389-405, :148-201)
//
// run on v_mfma_f32_32x32x16_f16 with every operand split into f16 hi + f16 lo and three MFMAs per product block
// (lo*hi, hi*lo, hi*hi), fp32 accumulate: ~22 significant bits per product at 3/16 of the fp32-MFMA time (mol_layout.h).
// Same kernel shells, same buffers and the same 1e-4 logit bar as the exact build (mol_score.hip).
//
// What shapes this file (measured on the part, profiles/r02_ubench_f16_mfma_vs_valu.txt):
//  * f16 MFMA overlaps with plain VALU work of the same or the partner wave (v_fma / v_cvt_pkrtz / v_fma_mix cost nothing
//    next to an MFMA stream until the VALU itself saturates at ~4.7 cycles per instruction), but NOT with packed-fp32
//    instructions (v_pk_*_f32: ~8 cycles each and they stall the matrix pipe) and only partly with transcendentals
//    (v_exp / v_rcp: ~9 cycles, ~4 of them not overlappable).  So this kernel is VALU-bound, and every VALU instruction
//    counts: no packed fp32 (the TU is built with -fno-slp-vectorize), no operand rescaling (f16 subnormals are kept by the
//    MFMA and by v_cvt_pkrtz, so all power-of-two scales are 1), no in-kernel conversion of Ex / Eq (both arrive pre-split,
//    written once by the index build / query prologue), v_fma_mixlo/hi avoided (transcendental-rate).
//  * one wave's stream is software-pipelined so that MFMAs always have independent VALU work next to them:
//      stage X(q)   GEMM2 of query q       ||  softmax/mixture epilogue of query q-1 (+ the operand split of cl)
//      stage Y(q)   GEMM3 K-step s         ||  silu + operand split of K-step s+1
//    Two waves per SIMD (or one with 512 registers) fill what is left.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

#include "mol_kernels.h"
#include "mol_layout.h"
#include "mol_score_shell.h"

// RAILS_F16_SINGLE = 1 (mol_score_f16x1.hip, mol_score_f16x1_extra.hip): the SAME kernels with the hi * hi product only and no lo
// halves anywhere -- plain f16 operands, fp32 accumulate: precision RAILS_PRECISION_F16X1, 1.6 x faster than f16x3 and ~1e-2
// away from the fp32 logits.  It reads the f16x3 packs (index, gate pack, query pack) and ignores their lo fragments.  Not a
// parity mode: it is the first pass of the speculate-then-verify top-k "f16-exact" (rails_amd/topk_modules.py).  Each build
// lives in its own inline namespace so that the two sets of template instantiations do not collide at link time.
#ifndef RAILS_F16_SINGLE
#define RAILS_F16_SINGLE 0
#endif
#ifndef RAILS_F16_TIGHT_PF
#define RAILS_F16_TIGHT_PF 2   // epilogue operand ring depth of the TIGHT stream
#endif
// Round-6 latency experiments on the TIGHT stream (8x8x32 at two waves per SIMD; A/B builds, tools/r06_f16_pipe_ab.sh; results: DESIGN.md 7.1):
//   RAILS_F16_TIGHT_YPIPE = 1  stage Y: silu + operand split of K-step s + 1 in the scheduling region of K-step s's MFMAs (both operand slots live)
//   RAILS_F16_TIGHT_XPIPE = 1  stage X: the cl operand split of K-step s + 1 in the region of K-step s's MFMAs (a second operand slot)
#ifndef RAILS_F16_TIGHT_YPIPE
#define RAILS_F16_TIGHT_YPIPE 0
#endif
#ifndef RAILS_F16_TIGHT_XPIPE
#define RAILS_F16_TIGHT_XPIPE 0
#endif

namespace mol {
#if RAILS_F16_SINGLE
inline namespace f16x1 {
#else
inline namespace f16x3 {
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// Scheduling directives (LLVM AMDGPU IGroupLP): inside one scheduling region, "then N instructions of class M" in program
// order of the calls.  Used to deal a region's VALU work between its MFMAs: the compiler's own schedule puts all VALU
// first and the MFMAs last, which leaves each pipe idle while the other works (PMC: MFMA busy + VALU busy = 100 %).
#define SG_VALU 0x002
#define SG_MFMA 0x008
#define SG_DS_READ 0x100
#define SG_TRANS 0x400
// N x { NV plain VALU, NT transcendental, 1 MFMA }
template <int N, int NV, int NT>
__device__ __forceinline__ void sched_interleave() {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(SG_VALU, NV, 0);
    __builtin_amdgcn_sched_group_barrier(SG_TRANS, NT, 0);
    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
  }
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a compile-time loop whose index is usable as a
// template argument (every register index of the unit is static)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// RAILS_F16_ABLATE (debug builds, tools/f16_ablation.sh): 1 = no MFMAs, 2 = no transcendentals, 3 = no VALU arithmetic at all --
// wrong results, used to price each instruction class in situ
#ifndef RAILS_F16_ABLATE
#define RAILS_F16_ABLATE 0
#endif
__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
#if RAILS_F16_ABLATE == 1
  asm volatile("" : "+v"(c) : "v"(a), "v"(b));
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ float f_exp2(float x) {
#if RAILS_F16_ABLATE >= 2
  return x * x;
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ float f_rcp(float x) {
#if RAILS_F16_ABLATE >= 2
  return x + 0.5f;
#else
  return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float swap32(float v) { return __shfl_xor(v, 32, 64); }

// two fp32 values -> packed f16 hi (round toward zero) and packed f16 lo (the fp32 remainder x - hi, exact, RTZ to f16):
// four full-rate VALU instructions per pair.  The remainder is one v_fma_mix_f32 (fma(f16 half of hi, -1.0, x) in fp32).
// `m1` is -1.0 in a VGPR the compiler cannot see through: fma(fpext(h), m1, x) then selects v_fma_mix_f32 (a visible -1.0 is
// folded into cvt + sub, two instructions), and unlike inline asm the instruction stays visible to the scheduler.
__device__ __forceinline__ void split_pair(float x0, float x1, float m1, unsigned& hi, unsigned& lo) {
#if RAILS_F16_ABLATE == 3
  hi = __builtin_bit_cast(unsigned, x0);
  lo = __builtin_bit_cast(unsigned, x1);
  return;
#endif
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  const h2v h = __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
#if RAILS_F16_SINGLE
  hi = __builtin_bit_cast(unsigned, h);
  lo = 0u;
  return;
#endif
  const float l0 = __builtin_fmaf((float)h.x, m1, x0), l1 = __builtin_fmaf((float)h.y, m1, x1);   // v_fma_mix_f32
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}
// this lane's 8 values of a K=16 step -> the step's hi and lo B operands
__device__ __forceinline__ void split8(const float (&x)[8], float m1, h8& hi, h8& lo) {
  u32x4v H, L;
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    unsigned h, l;
    split_pair(x[2 * pr], x[2 * pr + 1], m1, h, l);
    H[pr] = h;
    L[pr] = l;
  }
  hi = __builtin_bit_cast(h8, H);
  lo = __builtin_bit_cast(h8, L);
}
// t / (1 + 2^t) on the -log2e-prescaled argument (mol_layout.h): exp2, add, rcp, mul
__device__ __forceinline__ float nsilu(float t) {
#if RAILS_F16_ABLATE == 3
  return t;
#else
  const float e = f_exp2(t) + 1.0f;
  return t * f_rcp(e);
#endif
}

// Views of the split gate pack (rails_mol_pack_gate_weights with precision f16x3): [W1 hi][W1 lo][W2 hi][W2 lo][b1][b2], the whole
// pack in LDS.  (L = 256, whose pack is 256 KiB, runs the team kernel of mol_score_wsplit.h instead.)
template <class G>
struct SplitPack {
  const h8* w1hi; const h8* w1lo; const h8* w2hi; const h8* w2lo; const float* b1; const float* b2;
  float m1;   // -1.0, opaque (split_pair)
  static constexpr int N8 = G::kW1Floats / 8;   // h8 fragments per half of a weight matrix (hi or lo)
  static constexpr int kLdsFloats = G::kWpackFloats;
  __device__ __forceinline__ SplitPack(const float* smem, const float* gpack) {
    m1 = -1.0f;
    asm volatile("" : "+v"(m1));
    w1hi = reinterpret_cast<const h8*>(smem);
    w1lo = w1hi + N8;
    w2hi = w1lo + N8;
    w2lo = w2hi + N8;
    b1 = smem + G::kW1Floats + G::kW2Floats;
    b2 = b1 + G::TH * 32;
  }
  // W2 fragment f (hi or lo part) of this lane
  __device__ __forceinline__ h8 w2frag(bool hi_part, int f, int lane) const { return (hi_part ? w2hi : w2lo)[f * 64 + lane]; }
  template <int NW>
  static __device__ __forceinline__ void stage(const ScoreArgs& p, float* smem) { stage_weights<G, NW>(p, smem); }
};

// GEMM1 on pre-split operands: eq = [ks][hi|lo][lane] h8 (query pack), tEx = [m][ks][hi|lo][lane] h8 (tile, LDS or HBM).
// Item groups in chunks of <= 8 (their B fragments of a K-step are fetched together); inside a chunk the products are
// outermost so that consecutive MFMAs go to different accumulators.
// BULK: no fences between chunks -- the compiler then requests the whole tile up front (PX * d / 4 registers), which is what a
// wave that fetches its tile straight from HBM wants (one round trip per unit instead of one per chunk) and what only a
// one-wave-per-SIMD build has the registers for.
template <class G, int PX, int DD, bool BULK_ = false, bool PIPE = false>
__device__ __forceinline__ void gemm1_presplit(f32x16 (&D1)[PX], const h8* __restrict__ eq, const h8* tEx, int lane) {
  // PIPE (independent waves at two per SIMD, operands straight from memory): the one-product build reads only the hi fragments --
  // half the registers -- so it can request its whole tile up front like BULK (B = 1: 0.127 -> 0.110 ms); the three-product build cannot
  // (tile = 128 registers next to 128 of accumulators) and keeps one round trip per K-step (its hi fragments alone one step ahead
  // measured 2-6 % SLOWER at B = 1 ... 16).
  constexpr bool BULK = BULK_ || (PIPE && RAILS_F16_SINGLE != 0 && PX * (DD / 16) * 4 <= 64);
  static_assert(DD % 16 == 0, "f16x3 GEMM1 walks K in steps of 16");
  static_assert(PX <= 8, "the register-resident unit holds all item groups of a K-step at once");
  constexpr int MC = PX;
#pragma unroll
  for (int m = 0; m < PX; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) D1[m][r] = 0.0f;
  if constexpr (BULK) {
    // every fragment of the tile and of the query group requested before the first MFMA: ONE memory round trip per unit
    h8 a[DD / 8], b[PX][DD / 8];
#pragma unroll
    for (int c = 0; c < DD / 8; ++c)
      if (!RAILS_F16_SINGLE || c % 2 == 0) a[c] = eq[c * 64 + lane];
#pragma unroll
    for (int m = 0; m < PX; ++m)
#pragma unroll
      for (int c = 0; c < DD / 8; ++c)
        if (!RAILS_F16_SINGLE || c % 2 == 0) b[m][c] = tEx[(m * (DD / 8) + c) * 64 + lane];   // the one-product build never touches the lo fragments
    asm volatile("" ::: "memory");   // the requests stay above the MFMAs
#pragma unroll
    for (int ks = 0; ks < DD / 16; ++ks) {
#if !RAILS_F16_SINGLE
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a[2 * ks + 1], b[m][2 * ks], D1[m]);
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a[2 * ks], b[m][2 * ks + 1], D1[m]);
#endif
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a[2 * ks], b[m][2 * ks], D1[m]);
    }
    return;
  }
#pragma unroll
  for (int ks = 0; ks < DD / 16; ++ks) {
    const h8 ah = eq[(2 * ks) * 64 + lane], al = eq[(2 * ks + 1) * 64 + lane];
#pragma unroll
    for (int m0 = 0; m0 < PX; m0 += MC) {
      h8 bh[MC], bl[MC];
#pragma unroll
      for (int m = 0; m < MC; ++m) {
        bh[m] = tEx[((m0 + m) * (DD / 8) + 2 * ks) * 64 + lane];
        bl[m] = tEx[((m0 + m) * (DD / 8) + 2 * ks + 1) * 64 + lane];
      }
#if !RAILS_F16_SINGLE
#pragma unroll
      for (int m = 0; m < MC; ++m) D1[m0 + m] = mfma16(al, bh[m], D1[m0 + m]);
#pragma unroll
      for (int m = 0; m < MC; ++m) D1[m0 + m] = mfma16(ah, bl[m], D1[m0 + m]);
#endif
#pragma unroll
      for (int m = 0; m < MC; ++m) D1[m0 + m] = mfma16(ah, bh[m], D1[m0 + m]);
      // keep the operand fetches of later chunks below this chunk's MFMAs (register pressure)
      asm volatile("" ::: "memory");
    }
  }
}

// =================================================================================================================
// The unit's instruction stream is laid out BY HAND: a phase is a list of MFMAs and a list of VALU "slices" (a few
// instructions each, independent of the phase's MFMAs), emitted alternately with a scheduling fence after every MFMA, so
// that in program order every MFMA is followed by its share of VALU work.  (Left alone the compiler emits all VALU of a
// region, then all its MFMAs; IGroupLP's greedy solver only half fixes that and its exact solver does not terminate.)
// =================================================================================================================
template <int NM, int NS, class FM, class FS>
__device__ __forceinline__ void interleave(FM&& mf, FS&& sf) {
  static_for<NM>([&](auto ic) {
    constexpr int I = decltype(ic)::value;
    mf(ic);
    constexpr int s0 = I * NS / NM, s1 = (I + 1) * NS / NM;
    static_for<s1 - s0>([&](auto jc) { sf(std::integral_constant<int, s0 + decltype(jc)::value>{}); });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// ---- softmax / mixture epilogue of one query, in slices of two logits ------------------------------------------------
// softmax(w) is shift invariant and w = g*sigmoid(g) >= -0.2785, so the numerators are taken WITHOUT the usual maximum
// subtraction: ex = exp(w) = 2^(-u) with u = -log2e * w <= 0.402, i.e. ex >= 0.757 (no underflow, the denominator is
// >= 0.757 L) and ex overflows only for w > 88.7, a gate logit no trained model produces.  That saves a min3 per pair, a
// subtraction per logit and the cross-lane minimum; an overflow is detected on the denominator (inf / NaN) and that
// query is redone with the shifted form from the u values, which are still in registers -- same result as the
// reference's stable softmax (similarity_fn.py:31-46) in every case.
// PF = logit pairs whose gi / gq operands are requested ahead of their slice (gi may sit in HBM/L2); 4 registers each
template <class G, int PF, bool NONE = false>
struct Epi {
  static constexpr bool kNone = NONE;   // gating_combination "none" (similarity_fn.py:187-197): w = gq + gi + gqi, no silu.  A COMPILE-TIME
                                        // switch: as a run-time branch in every slice it cost the glu_silu kernels 15-25 % (2.52 -> 2.91 ms)
  static constexpr int kPF = PF;
  f32x16 D3[G::TL];   // -log2e * gqi on entry; u after pass 1
  float den, num;
  const float* gq;    // this query's -log2e * gq row, lane half's part ([hi][e] layout)
  float2 gi_r[PF], gq_r[PF];
  // gi fragment [ec = e/4][lane][4]: pair P is floats (e%4, e%4+1), e = 2P, of the lane's float4
  template <int P>
  __device__ __forceinline__ void fetch(const float* tGi, int lane) {
    constexpr int e = 2 * P;
    gi_r[P % PF] = *reinterpret_cast<const float2*>(tGi + ((e / 4) * 64 + lane) * 4 + e % 4);
    gq_r[P % PF] = *reinterpret_cast<const float2*>(gq + e);
  }
  __device__ __forceinline__ void reset(const float* gq_, const float* tGi, int lane) {
    den = 0.0f; num = 0.0f; gq = gq_;
    static_for<(PF < G::E / 2 ? PF : G::E / 2)>([&](auto pc) { fetch<decltype(pc)::value>(tGi, lane); });
  }
};
// pass 1, logits e = 2P, 2P+1 of this lane:  t2 = -log2e*(gq*gi + gqi);  u = t2/(1+2^t2) = -log2e * g*sigmoid(g)
template <class G, int P, class EP>
__device__ __forceinline__ void epi_p1(EP& s, const float* tGi, int lane) {
  constexpr int e = 2 * P, PF = EP::kPF;
  const float2 gi = s.gi_r[P % PF], gq = s.gq_r[P % PF];
  if constexpr (P + PF < G::E / 2) s.template fetch<P + PF>(tGi, lane);
  if constexpr (EP::kNone) {   // u = -log2e * (gq + gi + gqi): gq and gqi arrive prescaled, gi does not
    s.D3[e / 16][e % 16] = __builtin_fmaf(gi.x, -kLog2e, gq.x + s.D3[e / 16][e % 16]);
    s.D3[e / 16][e % 16 + 1] = __builtin_fmaf(gi.y, -kLog2e, gq.y + s.D3[e / 16][e % 16 + 1]);
  } else {
    s.D3[e / 16][e % 16] = nsilu(__builtin_fmaf(gq.x, gi.x, s.D3[e / 16][e % 16]));
    s.D3[e / 16][e % 16 + 1] = nsilu(__builtin_fmaf(gq.y, gi.y, s.D3[e / 16][e % 16 + 1]));
  }
}
// pass 2:  ex = 2^(-u) = softmax numerator;  den += ex;  num += ex * cl   (cl of this query: D1 registers R0 + ...)
template <class G, int PX, int R0, int P, class EP>
__device__ __forceinline__ void epi_p2(EP& s, const f32x16 (&D1)[PX]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    constexpr int e0 = 2 * P;
    const int e = e0 + j;
    const float ex = f_exp2(-s.D3[e / 16][e % 16]);
    s.den += ex;
    s.num = __builtin_fmaf(ex, D1[e / G::RPQ][R0 + e % G::RPQ], s.num);
  }
}
// slice S of the E slices of an epilogue: S < E/2 -> pass 1 of pair S; else pass 2 of pair S - E/2
template <class G, int PX, int R0, int S, class EP>
__device__ __forceinline__ void epi_slice(EP& s, const f32x16 (&D1)[PX], const float* tGi, int lane) {
  constexpr int HALF = G::E / 2;
  if constexpr (S < HALF) epi_p1<G, S>(s, tGi, lane);
  else epi_p2<G, PX, R0, S - HALF>(s, D1);
}
// pi = ex/den, then the eval-time renormalisation pi / clamp(sum pi, 1e-6) (similarity_fn.py:42-46): sum pi = den * (1/den)
template <class G, int PX, int R0, class EP>
__device__ __forceinline__ float epi_final(EP& s, const f32x16 (&D1)[PX]) {
  float den = s.den + swap32(s.den), num = s.num + swap32(s.num);
  // The guard sits well below FLT_MAX: with den near 1e38 the sum is still finite, but num = sum ex * cl (|cl| <= 1/tau) overflows
  // first and 1/den is a denormal that v_rcp flushes to zero -- NaNs and zeros for gate logits just under the exp overflow
  // (found with pair-gate weights x 3: 1 084 non-finite logits of 2.4 M; x 5 and x 10 overflowed den itself and were caught).
  // an exp got large somewhere in this wave (or, with "none", whose gate logits have no lower bound, all of them tiny): the stable form
  if (__builtin_amdgcn_ballot_w64(EP::kNone ? !(den < 1.0e30f && den > 1.0e-30f) : !(den < 1.0e30f)) != 0) {
    float mn = INFINITY;
#pragma unroll
    for (int e = 0; e < G::E; ++e) mn = __builtin_fminf(mn, s.D3[e / 16][e % 16]);
    mn = __builtin_fminf(mn, swap32(mn));
    den = 0.0f;
    num = 0.0f;
#pragma unroll
    for (int e = 0; e < G::E; ++e) {
      const float ex = f_exp2(mn - s.D3[e / 16][e % 16]);
      den += ex;
      num = __builtin_fmaf(ex, D1[e / G::RPQ][R0 + e % G::RPQ], num);
    }
    den += swap32(den);
    num += swap32(num);
  }
  const float rden = __builtin_amdgcn_rcpf(den);
  return (num * rden) / fmaxf(den * rden, 1e-6f);
}

// ---- the two gate GEMMs as MFMA sequences ---------------------------------------------------------------------------
// A stage has T row tiles, NK K-steps and three products per (row tile, K-step): lo*hi, hi*lo, hi*hi.  The dependent-
// accumulate latency of v_mfma_f32_32x32x16_f16 is twice its issue time (64 vs 32 cycles: back-to-back MFMAs into one
// accumulator run at half rate -- the MFMA-only ablation of the first hand-ordered version took 1.78 ms against 1.0), so
// consecutive MFMAs always alternate between two accumulators:
//   T even:  row tiles in pairs (a, b); per pair and K-step  p0a p0b p1a p1b p2a p2b
//   T == 1:  the single row tile accumulates into two partial accumulators, MFMA n -> accumulator n & 1, summed at the end
// Weight fragments sit in four register slots (lo/hi x a/b); a slot is refilled from LDS right after its last use, i.e.
// >= 4 MFMAs before the next use, with no second buffer.
template <int T, int NK>
struct Seq {
  static_assert(T == 1 || T % 2 == 0, "row tiles: one, or an even number");
  static constexpr int N = 3 * T * NK;
  static constexpr int GS = T == 1 ? 1 : 2;             // row tiles per group
  static constexpr int NGRP = T * NK / GS;              // groups of the stage, K-step major
  static constexpr int PER = 3 * GS;                    // MFMAs per group
  static constexpr int group(int I) { return I / PER; }
  static constexpr int kstep(int I) { return group(I) / (T / GS); }
  static constexpr int side(int I) { return T == 1 ? 0 : (I % PER) & 1; }              // a / b
  static constexpr int prod(int I) { return T == 1 ? I % 3 : (I % PER) / 2; }
  static constexpr int tile(int I) { return T == 1 ? 0 : GS * (group(I) % (T / GS)) + side(I); }
  static constexpr int acc(int I) { return T == 1 ? (I & 1) : tile(I); }                // accumulator index (T == 1: two partials)
  static constexpr int frag(int grp, int sd) { return T == 1 ? grp : (grp / (T / GS)) * T + GS * (grp % (T / GS)) + sd; }   // [ks][tile]
};
// R = groups in flight: 1 for fragments in LDS (refilled >= 4 MFMAs before the next use), more for fragments streamed from L2
template <int R>
struct WSlots { h8 lo[R][2], hi[R][2]; };   // [ring slot][side]

// one MFMA of a stage + the slot refills that become possible after it.  W(hi?, fragment) reads a weight fragment.
template <class S, int I, int R, int NACC, class WF>
__device__ __forceinline__ void seq_mfma(f32x16 (&acc)[NACC], WSlots<R>& ws, h8 bh, h8 bl, WF&& W) {
  constexpr int sd = S::side(I), pr = S::prod(I), grp = S::group(I), slot = grp % R;
  f32x16& d = acc[S::acc(I)];
#if RAILS_F16_SINGLE
  if constexpr (pr == 2) d = mfma16(ws.hi[slot][sd], bh, d);
  if constexpr (grp + R < S::NGRP) {
    if constexpr (pr == 2) ws.hi[slot][sd] = W(true, S::frag(grp + R, sd));
  }
#else
  if constexpr (pr == 0) d = mfma16(ws.lo[slot][sd], bh, d);
  else if constexpr (pr == 1) d = mfma16(ws.hi[slot][sd], bl, d);
  else d = mfma16(ws.hi[slot][sd], bh, d);
  if constexpr (grp + R < S::NGRP) {
    if constexpr (pr == 0) ws.lo[slot][sd] = W(false, S::frag(grp + R, sd));
    if constexpr (pr == 2) ws.hi[slot][sd] = W(true, S::frag(grp + R, sd));
  }
#endif
}
template <class S, int R, class WF>
__device__ __forceinline__ void seq_begin(WSlots<R>& ws, WF&& W) {
#pragma unroll
  for (int g = 0; g < (R < S::NGRP ? R : S::NGRP); ++g)
#pragma unroll
    for (int sd = 0; sd < S::GS; ++sd) {
#if !RAILS_F16_SINGLE
      ws.lo[g][sd] = W(false, S::frag(g, sd));
#endif
      ws.hi[g][sd] = W(true, S::frag(g, sd));
    }
}

// ---- stage X: GEMM2 of the query whose cl sit in D1 registers [R0, R0 + RPQ):  D2 = -log2e * (b1 + W1 cl) -----------
template <class G>
struct XState {
  WSlots<1> ws;
  h8 bh, bl;         // current K-step's cl operand, split right before the K-step's first MFMA
#if RAILS_F16_TIGHT_XPIPE
  h8 bh2, bl2;       // the next K-step's, split one region ahead
#endif
};
template <class G, class WP>
__device__ __forceinline__ void init_d2(f32x16 (&D2)[G::TH], const WP& w, int hi) {
#pragma unroll
  for (int t = 0; t < G::TH; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) D2[t][r] = w.b1[t * 32 + hi * 16 + r];
}
template <class G, int PX, int R0, int KS, class WP>
__device__ __forceinline__ void cl_split(const f32x16 (&D1)[PX], const WP& w, h8& bh, h8& bl) {
  float xs[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int e = 8 * KS + jj;
    xs[jj] = D1[e / G::RPQ][R0 + e % G::RPQ];
  }
  split8(xs, w.m1, bh, bl);
}
template <class G>
using XSeq = Seq<G::TH, G::E / 8>;
template <class G, class WP>
__device__ __forceinline__ void x_begin(XState<G>& st, const WP& w, int lane) {
  static_assert(G::TH % 2 == 0, "stage X keeps one accumulator per row tile: the hidden dim must be a multiple of 64");
  seq_begin<XSeq<G>>(st.ws, [&](bool hi_part, int f) { return (hi_part ? w.w1hi : w.w1lo)[f * 64 + lane]; });
}
template <class G, int PX, int R0, int I, class WP>
__device__ __forceinline__ void x_mfma(const f32x16 (&D1)[PX], f32x16 (&D2)[G::TH], XState<G>& st, const WP& w, int lane) {
  using S = XSeq<G>;
  if constexpr (I % (3 * G::TH) == 0) cl_split<G, PX, R0, S::kstep(I)>(D1, w, st.bh, st.bl);
  seq_mfma<S, I>(D2, st.ws, st.bh, st.bl, [&](bool hi_part, int f) { return (hi_part ? w.w1hi : w.w1lo)[f * 64 + lane]; });
}

// ---- stage Y: hid' = t/(1+2^t);  D3 = -log2e * (b2 + W2 hid) -----------------------------------------------------------
// The B operand of K-step s+1 (silu + split of 8 hidden values, four slices of one value pair each) is produced under the
// MFMAs of K-step s, into the other of two operand slots.
template <class G, int RY>
struct YState {
  static constexpr int NACC = G::TL == 1 ? 2 : G::TL;
  f32x16 part[G::TL == 1 ? 2 : 1];   // TL == 1 only: the two partial accumulators (Seq); otherwise D3 itself is accumulated into
  WSlots<RY> ws;
  u32x4v bh[2], bl[2];  // hid operands of K-steps s (slot s & 1), built pair by pair
};
template <class G>
using YSeq = Seq<G::TL, G::F / 8>;
template <class G, int SL, class WP, class YS>   // slice SL: value pair SL % 4 of K-step SL / 4
__device__ __forceinline__ void silu_slice(const f32x16 (&D2)[G::TH], YS& st, const WP& w) {
  constexpr int ks = SL / 4, pr = SL % 4, f = 8 * ks + 2 * pr;
  const float h0 = nsilu(D2[f / 16][f % 16]), h1 = nsilu(D2[f / 16][f % 16 + 1]);
  unsigned h, l;
  split_pair(h0, h1, w.m1, h, l);
  st.bh[ks & 1][pr] = h;
  st.bl[ks & 1][pr] = l;
}
template <class G, class WP, class YS>
__device__ __forceinline__ void y_begin(f32x16 (&D3)[G::TL], YS& st, const WP& w, int lane, int hi) {
  if constexpr (G::TL == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st.part[0][r] = w.b2[hi * G::E + r];
      st.part[1][r] = 0.0f;
    }
  } else {
#pragma unroll
    for (int v = 0; v < G::TL; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) D3[v][r] = w.b2[hi * G::E + v * 16 + r];
  }
  seq_begin<YSeq<G>>(st.ws, [&](bool hi_part, int f) { return w.w2frag(hi_part, f, lane); });
}
template <class G, int I, class WP, class YS>
__device__ __forceinline__ void y_mfma(f32x16 (&D3)[G::TL], YS& st, const WP& w, int lane) {
  using S = YSeq<G>;
  constexpr int ks = S::kstep(I);
  auto W = [&](bool hi_part, int f) { return w.w2frag(hi_part, f, lane); };
  const h8 bh = __builtin_bit_cast(h8, st.bh[ks & 1]), bl = __builtin_bit_cast(h8, st.bl[ks & 1]);
  if constexpr (G::TL == 1) seq_mfma<S, I>(st.part, st.ws, bh, bl, W);
  else seq_mfma<S, I>(D3, st.ws, bh, bl, W);
}
template <class G, class YS>
__device__ __forceinline__ void y_end(const YS& st, f32x16 (&D3)[G::TL]) {
  if constexpr (G::TL == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) D3[0][r] = st.part[0][r] + st.part[1][r];
  }
}

#ifdef RAILS_F16_PHASES   // tools/f16_phases.sh: shader-clock stamps of workgroup 0 / wave 0's units (the last one stays)
static __device__ long long g_f16_phase[32];
#define F16_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && item0 == 32 * 20 * (int64_t)gridDim.x) g_f16_phase[i] = (long long)clock64(); } while (0)
#else
#define F16_STAMP(i)
#endif

// OVERLAP: stage X of query Q+1 carries the epilogue of query Q (needs D2 and D3 of two queries live at once).
// TIGHT:   the accumulators alone fill the register budget (8x8x32 at two waves per SIMD: 224 of 256): no operand double
//          buffering in stage Y and no pinned order -- the compiler's own schedule fits without spilling, a pinned one does not.
// UPPER:   the unit writes logit + (ub2 c + ub1) c + ub0, c = max |cl| over the pair's logits as GEMM1 left them: an upper bound of the pair's
//          fp32 logit under the per-pair form of the a-priori bound (rails_mol_score_dense_upper; mol_score_wsplit.h has the team kernel's).
template <bool OVERLAP, bool TIGHT, bool NONE = false, bool UPPER = false>
struct F16Unit {
  static constexpr bool kIndexedCandidates = false;
  template <class G>
  static constexpr int kLdsWeightFloats = SplitPack<G>::kLdsFloats;
  template <class G, int NW>
  static __device__ __forceinline__ void stage(const ScoreArgs& p, float* smem) { SplitPack<G>::template stage<NW>(p, smem); }

  template <class G, int PX, int DD, bool BULK = false, int PIPE = 0>   // PIPE > 1 (the fp32 unit's register ring) means "none" here
  static __device__ __forceinline__ void gemm1(f32x16 (&D1)[PX], const float* __restrict__ eq, const float4* tEx, int lane) {
    gemm1_presplit<G, PX, DD, BULK, (PIPE == 1)>(D1, reinterpret_cast<const h8*>(eq), reinterpret_cast<const h8*>(tEx), lane);
  }

  template <class G, int PX>
  static __device__ __forceinline__ void queries(f32x16 (&D1)[PX], const ScoreArgs& p, int g, int only, int64_t item0,
                                                 const float* smem, const float4* tGi4, int lane, int hi, int x) {
    constexpr int NXM = (G::E / 8) * G::TH * 3;   // MFMAs of stage X
    constexpr int NYM = (G::F / 8) * G::TL * 3;   // MFMAs of stage Y
    constexpr int NYS = G::F / 8;                 // K-steps of stage Y
    const SplitPack<G> w(smem, p.wpack);
    const float* tGi = reinterpret_cast<const float*>(tGi4);
    const int64_t item = item0 + x;
    const bool lane_stores = hi == 0 && item < p.n_items;
    // rows past the batch end (padding of the last group) run on zero operands and the last real gate row; never stored
    auto gq_of = [&](int q) { return p.gqfrag + (int64_t)(q < p.B ? q : p.B - 1) * G::L + hi * G::E; };
    // UPPER: accumulator registers [Q RPQ, (Q + 1) RPQ) of every D1[m] are query Q's logits of item x (rows (r & 3) + 8 (r >> 2) + 4 hi); the two
    // lane halves hold the two halves of the row set.  The maximum is folded where the query's logit is stored (D1 lives that long anyway):
    // nothing extra stays live across the unit
    auto store = [&](auto qc, int q, float out) {
      if constexpr (UPPER) {
        constexpr int Q = decltype(qc)::value;
        float c = 0.0f;
#pragma unroll
        for (int m = 0; m < PX; ++m)
#pragma unroll
          for (int r = 0; r < G::RPQ; ++r) c = fmaxf(c, fabsf(D1[m][Q * G::RPQ + r]));
        c = fmaxf(c, __shfl_xor(c, 32, 64));
        out += __builtin_fmaf(__builtin_fmaf(p.ub2, c, p.ub1), c, p.ub0);
      }
      if (lane_stores && q < p.B) p.logits[(int64_t)q * p.ld + item] = out;
    };

    f32x16 D2[G::TH];
    Epi<G, (TIGHT ? RAILS_F16_TIGHT_PF : 8), NONE> ep;   // TIGHT has no registers to spare for a deeper operand ring (and its epilogue is not fenced: the compiler hoists)
    XState<G> xs;
    YState<G, 1> ys;
    auto stage_x_alone = [&](auto qc) {   // GEMM2 with nothing to hide it under but the operand splits
      constexpr int Q = decltype(qc)::value;
      init_d2<G>(D2, w, hi);
      x_begin<G>(xs, w, lane);
      if constexpr (TIGHT) {
#if RAILS_F16_TIGHT_XPIPE
        cl_split<G, PX, Q * G::RPQ, 0>(D1, w, xs.bh, xs.bl);
        __builtin_amdgcn_sched_barrier(0);
        static_for<G::E / 8>([&](auto kc) {
          constexpr int KS = decltype(kc)::value;
          if constexpr (KS + 1 < G::E / 8) cl_split<G, PX, Q * G::RPQ, KS + 1>(D1, w, xs.bh2, xs.bl2);
          static_for<3 * G::TH>([&](auto ic) {
            constexpr int I = 3 * G::TH * KS + decltype(ic)::value;
            seq_mfma<XSeq<G>, I>(D2, xs.ws, xs.bh, xs.bl, [&](bool hi_part, int f) { return (hi_part ? w.w1hi : w.w1lo)[f * 64 + lane]; });
          });
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (KS + 1 < G::E / 8) { xs.bh = xs.bh2; xs.bl = xs.bl2; }
        });
        return;
#endif
        static_for<G::E / 8>([&](auto kc) {
          static_for<3 * G::TH>([&](auto ic) { x_mfma<G, PX, Q * G::RPQ, 3 * G::TH * decltype(kc)::value + decltype(ic)::value>(D1, D2, xs, w, lane); });
          __builtin_amdgcn_sched_barrier(0);
        });
        return;
      }
      interleave<NXM, 0>([&](auto ic) { x_mfma<G, PX, Q * G::RPQ, decltype(ic)::value>(D1, D2, xs, w, lane); }, [&](auto) {});
    };
    auto stage_y = [&](auto qc) {         // silu of K-step 0 exposed, then GEMM3 || silu of the following K-steps
      y_begin<G>(ep.D3, ys, w, lane, hi);
      if constexpr (TIGHT) {
#if RAILS_F16_TIGHT_YPIPE
        static_for<4>([&](auto sc) { silu_slice<G, decltype(sc)::value>(D2, ys, w); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<NYS>([&](auto kc) {
          constexpr int KS = decltype(kc)::value;
          if constexpr (KS + 1 < NYS) static_for<4>([&](auto sc) { silu_slice<G, 4 * (KS + 1) + decltype(sc)::value>(D2, ys, w); });   // into slot (KS + 1) & 1
          static_for<3 * G::TL>([&](auto ic) { y_mfma<G, 3 * G::TL * KS + decltype(ic)::value>(ep.D3, ys, w, lane); });
          __builtin_amdgcn_sched_barrier(0);
        });
        y_end<G>(ys, ep.D3);
        return;
#endif
        static_for<NYS>([&](auto kc) {
          constexpr int KS = decltype(kc)::value;
          static_for<4>([&](auto sc) { silu_slice<G, 4 * KS + decltype(sc)::value>(D2, ys, w); });
          static_for<3 * G::TL>([&](auto ic) { y_mfma<G, 3 * G::TL * KS + decltype(ic)::value>(ep.D3, ys, w, lane); });
          __builtin_amdgcn_sched_barrier(0);   // K-steps stay in order (left alone the scheduler hoists every LDS read of the stage and spills)
        });
        y_end<G>(ys, ep.D3);
        return;
      }
      static_for<4>([&](auto sc) { silu_slice<G, decltype(sc)::value>(D2, ys, w); });
      __builtin_amdgcn_sched_barrier(0);
      interleave<NYM - 3 * G::TL, 4 * (NYS - 1)>([&](auto ic) { y_mfma<G, decltype(ic)::value>(ep.D3, ys, w, lane); },
                                                 [&](auto sc) { silu_slice<G, 4 + decltype(sc)::value>(D2, ys, w); });
      static_for<3 * G::TL>([&](auto ic) { y_mfma<G, NYM - 3 * G::TL + decltype(ic)::value>(ep.D3, ys, w, lane); });
      y_end<G>(ys, ep.D3);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto epilogue_alone = [&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      static_for<G::E>([&](auto sc) { epi_slice<G, PX, Q * G::RPQ, decltype(sc)::value>(ep, D1, tGi, lane); });
      return epi_final<G, PX, Q * G::RPQ>(ep, D1);
    };

    if (only >= 0 || !OVERLAP || (g + 1) * G::QT > p.B) {
      // per-row candidates (one query of the group), a group that reaches past the batch end (small batches: skip the padding
      // queries), or the cross-query overlap switched off: query by query
      static_for<G::QT>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        const int q = g * G::QT + Q;
        if (q < p.B && (only < 0 || q == only)) {
          F16_STAMP(4 * Q);
          stage_x_alone(qc);
          F16_STAMP(4 * Q + 1);
          ep.reset(gq_of(q), tGi, lane);
          stage_y(qc);
          F16_STAMP(4 * Q + 2);
          store(qc, q, epilogue_alone(qc));
          F16_STAMP(4 * Q + 3);
        }
      });
      return;
    }
    // shared corpus: all QT queries of the group in one straight-line stream;
    // stage X of query Q+1 carries the epilogue of query Q between its MFMAs
    F16_STAMP(0);
    stage_x_alone(std::integral_constant<int, 0>{});
    F16_STAMP(1);
    static_for<G::QT>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      const int q = g * G::QT + Q;
      ep.reset(gq_of(q), tGi, lane);
      stage_y(qc);
      F16_STAMP(2 + 2 * Q);
      if constexpr (Q + 1 < G::QT) {
        init_d2<G>(D2, w, hi);
        x_begin<G>(xs, w, lane);
        interleave<NXM, G::E>([&](auto ic) { x_mfma<G, PX, (Q + 1) * G::RPQ, decltype(ic)::value>(D1, D2, xs, w, lane); },
                              [&](auto sc) { epi_slice<G, PX, Q * G::RPQ, decltype(sc)::value>(ep, D1, tGi, lane); });
        store(qc, q, epi_final<G, PX, Q * G::RPQ>(ep, D1));
      } else {
        store(qc, q, epilogue_alone(qc));
      }
      F16_STAMP(3 + 2 * Q);
    });
  }
};

}  // inline namespace f16x3 / f16x1
}  // namespace mol
