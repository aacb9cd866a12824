// Fused Mixture-of-Logits scoring kernel for gfx950 (MI355X).
//
// Computes, for every (query b, item x) pair, steps 5-8 of the reference's eval-mode
// MoLSimilarity.forward (rails/similarities/mol/similarity_fn.py:389-413, gate :148-201,
// combiner :31-46) from the precomputed query-side (Eq, gq) and item-side (Ex, gi) operands:
//
//   cl[l]  = <Eq[b,p,:], Ex[x,m,:]> / tau                 l = p*P_X + m
//   hid    = silu(W1 cl + b1)                             (H)
//   gqi    = W2 hid + b2                                  (L)
//   g      = gq[b] * gi[x] + gqi ;  w = g * sigmoid(g)
//   pi     = softmax(w) ; pi /= clamp(sum pi, 1e-6)       (the eval-time renormalisation)
//   out    = sum_l pi[l] * cl[l]
//
// None of the (B, N, L) / (B, N, H) intermediates the reference materialises ever leaves the
// register file: one wave owns one unit = (query group of 32/P_Q queries) x (tile of 32 items),
// runs the sub-embedding contraction as 32x32x2 fp32 MFMAs with the items on the column axis,
// and feeds the accumulator registers straight back as the B operand of the two gate GEMMs
// (see mol_layout.h).  Arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32 == an fmaf chain), which is
// what lets the result sit within 1e-4 of the fp32 CPU path; the bound is the fp32 MFMA rate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "mol_kernels.h"
#include "mol_layout.h"
#include "mol_score_shell.h"

namespace mol {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): two values per VALU issue slot
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_sigmoid_arg(f32x2 t) {  // 1 / (1 + 2^t)
  f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  e = e + 1.0f;
  return f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// ---------------------------------------------------------------------------------------------
// Building blocks shared by the two kernels below.
// ---------------------------------------------------------------------------------------------

// GEMM1: D1[m][(qj,p), x] = sum_d Eq[qj,p,d] * Ex[x,m,d].  `eq` is the query group's A operand in fragment
// order ([sc][lane] float4), `tEx` the tile's B operand ([m][sc][lane] float4) -- in HBM or in LDS.
// PIPE: 0 = each K-chunk loaded where it is used; 1 = one chunk requested ahead; n > 1 = a register ring of n chunks in flight.
// SS: float4 stride between consecutive fragment slots of `tEx` as the lane sees them -- 64 in a tile (slot s of lane l at tEx[64 s + l]); 2 in
// the row-major copy of the index (rails_mol_index_rows_build: item i's slot s, half h at rows[i * RP + 2 s + h]; the rows kernel hands the
// unit the per-lane pointer row + h - lane, so that tEx[2 s + l] lands there).  Same values, same order, same bits.
template <class G, int PX, int DD, bool BULK = false, int PIPE = 0, int SS = 64>
__device__ __forceinline__ void gemm1(f32x16 (&D1)[PX], const float4* __restrict__ eq, const float4* tEx, int lane) {
#pragma unroll
  for (int m = 0; m < PX; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) D1[m][r] = 0.0f;
  if constexpr ((PIPE > 1) && (DD / 8) > PIPE) {
    // deep K (8x4x128: 16 chunks of 16 MFMAs): a chunk's A operand comes from the query pack in L1 / L2 (500-900 cycles under load)
    // against 1 024 cycles of MFMAs per chunk, and one chunk ahead leaves part of the round trip exposed; PIPE chunks are kept in
    // flight in a register ring.  Same chain order, same bits.
    constexpr int PD = PIPE, NC = DD / 8;
    float4 ra[PD], rb[PD][PX];
#pragma unroll
    for (int c = 0; c < PD; ++c) {
      ra[c] = eq[c * 64 + lane];
#pragma unroll
      for (int m = 0; m < PX; ++m) rb[c][m] = tEx[(m * NC + c) * SS + lane];
    }
#pragma unroll
    for (int sc = 0; sc < NC; ++sc) {
      const float4 a = ra[sc % PD];
      float4 b[PX];
#pragma unroll
      for (int m = 0; m < PX; ++m) b[m] = rb[sc % PD][m];
      if (sc + PD < NC) {
        ra[sc % PD] = eq[(sc + PD) * 64 + lane];
#pragma unroll
        for (int m = 0; m < PX; ++m) rb[sc % PD][m] = tEx[(m * NC + sc + PD) * SS + lane];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < PX; ++m) {
        D1[m] = mfma32(a.x, b[m].x, D1[m]);
        D1[m] = mfma32(a.y, b[m].y, D1[m]);
        D1[m] = mfma32(a.z, b[m].z, D1[m]);
        D1[m] = mfma32(a.w, b[m].w, D1[m]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if constexpr (PIPE != 0) {
    // Operands straight from memory (independent-wave shell): K-chunk sc + 1 is requested BEFORE chunk sc's MFMAs, so a chunk's
    // memory latency runs under 2 048 cycles of this wave's matrix work (and its partner's) instead of in front of them.  Costs PX
    // more float4 of registers, in the one phase of the unit that has them to spare (no D2 / D3 yet).
    float4 a_n = eq[lane], b_n[PX];
#pragma unroll
    for (int m = 0; m < PX; ++m) b_n[m] = tEx[(m * (DD / 8)) * SS + lane];
#pragma unroll
    for (int sc = 0; sc < DD / 8; ++sc) {
      const float4 a = a_n;
      float4 b[PX];
#pragma unroll
      for (int m = 0; m < PX; ++m) b[m] = b_n[m];
      if (sc + 1 < DD / 8) {
        a_n = eq[(sc + 1) * 64 + lane];
#pragma unroll
        for (int m = 0; m < PX; ++m) b_n[m] = tEx[(m * (DD / 8) + sc + 1) * SS + lane];
      }
      __builtin_amdgcn_sched_barrier(0);   // the requests stay above this chunk's MFMAs
#pragma unroll
      for (int m = 0; m < PX; ++m) {
        D1[m] = mfma32(a.x, b[m].x, D1[m]);
        D1[m] = mfma32(a.y, b[m].y, D1[m]);
        D1[m] = mfma32(a.z, b[m].z, D1[m]);
        D1[m] = mfma32(a.w, b[m].w, D1[m]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
#pragma unroll
  for (int sc = 0; sc < DD / 8; ++sc) {
    const float4 a = eq[sc * 64 + lane];
#pragma unroll
    for (int m = 0; m < PX; ++m) {
      const float4 b = tEx[(m * (DD / 8) + sc) * SS + lane];
      D1[m] = mfma32(a.x, b.x, D1[m]);
      D1[m] = mfma32(a.y, b.y, D1[m]);
      D1[m] = mfma32(a.z, b.z, D1[m]);
      D1[m] = mfma32(a.w, b.w, D1[m]);
    }
    // keep the operand fetches of later K-chunks below this chunk's MFMAs: left alone, the scheduler hoists
    // every read of the tile to the top (128 live registers) and spills.  BULK (one wave per SIMD, tile straight from HBM)
    // wants exactly that hoist: one memory round trip per unit instead of one per chunk.
    if constexpr (!BULK) asm volatile("" ::: "memory");
  }
}

// One query of the group: gate MLP (GEMM2 -> silu -> GEMM3), combine, softmax, mixture, on pre-scaled operands
// (mol_layout.h).  The query's cl values sit in accumulator registers [R0, R0 + RPQ) of every D1 tile.
template <class G, int PX, int R0, int SS = 64>
__device__ __forceinline__ float query_mlp(f32x16 (&D1)[PX], const float4* sW1, const float4* sW2, const float* sB1,
                                           const float* sB2, const float4* tGi, const float4* __restrict__ gq4,
                                           int lane, int hi, int combine_none) {
  f32x16 D3[G::TL];
  if constexpr (G::TH == 0) {
    // pair gate without hidden layer: gqi'[l, x] = -log2e * (b[l] + sum_l' W[l, l'] cl[l', x]) -- one GEMM over the logit axis, its
    // weights in the W1 slot (sW1), its bias in the b2 slot (sB2)
#pragma unroll
    for (int v = 0; v < G::TL; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) D3[v][r] = sB2[hi * G::E + v * 16 + r];
#pragma unroll
    for (int ec = 0; ec < G::E / 4; ++ec) {
#pragma unroll
      for (int v = 0; v < G::TL; ++v) {
        const float4 a = sW1[(ec * G::TL + v) * 64 + lane];
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = ec * 4 + j;
          D3[v] = mfma32(av[j], D1[e / G::RPQ][R0 + e % G::RPQ], D3[v]);
        }
      }
    }
  } else {
  // GEMM2: t[h, x] = -log2e * (b1[h] + sum_l W1[h, l] cl[l, x])
  f32x16 D2[G::TH > 0 ? G::TH : 1];
#pragma unroll
  for (int t = 0; t < G::TH; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) D2[t][r] = sB1[t * 32 + hi * 16 + r];
#pragma unroll
  for (int ec = 0; ec < G::E / 4; ++ec) {
#pragma unroll
    for (int t = 0; t < G::TH; ++t) {
      const float4 a = sW1[(ec * G::TH + t) * 64 + lane];
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = ec * 4 + j;
        D2[t] = mfma32(av[j], D1[e / G::RPQ][R0 + e % G::RPQ], D2[t]);
      }
    }
  }
  // hid' = t / (1 + 2^t) = -log2e * silu(pre): exp2, add, rcp, mul.  Fenced from the MFMAs on both sides: fp32
  // MFMA and VALU do not overlap on gfx950, so interleaving them only buys VALU->MFMA hazard nops.
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < G::TH; ++t)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 tv = {D2[t][r], D2[t][r + 1]};
      const f32x2 h = tv * pk_sigmoid_arg(tv);
      D2[t][r] = h.x;
      D2[t][r + 1] = h.y;
    }
  __builtin_amdgcn_sched_barrier(0);

  // GEMM3: gqi'[l, x] = -log2e * (b2[l] + sum_h W2[l, h] hid[h, x])
#pragma unroll
  for (int v = 0; v < G::TL; ++v)
#pragma unroll
    for (int r = 0; r < 16; ++r) D3[v][r] = sB2[hi * G::E + v * 16 + r];
#pragma unroll
  for (int fc = 0; fc < G::F / 4; ++fc) {
#pragma unroll
    for (int v = 0; v < G::TL; ++v) {
      const float4 a = sW2[(fc * G::TL + v) * 64 + lane];
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = fc * 4 + j;
        D3[v] = mfma32(av[j], D2[f / 16][f % 16], D3[v]);
      }
    }
  }

  }
  // epilogue.  t2 = -log2e * (gq*gi + gqi);  u = t2 / (1 + 2^t2) = -log2e * g*sigmoid(g);  softmax(w) = 2^(min u - u) / sum
  // gating_combination "none" (similarity_fn.py:187-197): w = gq + gi + gqi, no silu: u = gq' + gqi' - log2e * gi
  __builtin_amdgcn_sched_barrier(0);
  float mn = INFINITY;
  if (combine_none) {
#pragma unroll
    for (int ec = 0; ec < G::E / 4; ++ec) {
      const float4 gi = tGi[ec * SS + lane];
      const float4 gq = gq4[ec];
      const float giv[4] = {gi.x, gi.y, gi.z, gi.w}, gqv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = ec * 4 + j;
        const float u = __builtin_fmaf(giv[j], -kLog2e, gqv[j] + D3[e / 16][e % 16]);
        D3[e / 16][e % 16] = u;
        mn = fminf(mn, u);
      }
    }
  } else {
#pragma unroll
    for (int ec = 0; ec < G::E / 4; ++ec) {
      const float4 gi = tGi[ec * SS + lane];
      const float4 gq = gq4[ec];
      const f32x2 giv[2] = {{gi.x, gi.y}, {gi.z, gi.w}};
      const f32x2 gqv[2] = {{gq.x, gq.y}, {gq.z, gq.w}};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int e = ec * 4 + 2 * j;
        const f32x2 t2 = pk_fma(gqv[j], giv[j], f32x2{D3[e / 16][e % 16], D3[e / 16][e % 16 + 1]});
        const f32x2 u = t2 * pk_sigmoid_arg(t2);
        D3[e / 16][e % 16] = u.x;
        D3[e / 16][e % 16 + 1] = u.y;
        mn = fminf(mn, fminf(u.x, u.y));
      }
    }
  }
  mn = fminf(mn, xor32(mn));
  f32x2 den2 = {0.0f, 0.0f}, num2 = {0.0f, 0.0f};
#pragma unroll
  for (int e = 0; e < G::E; e += 2) {
    const f32x2 d = mn - f32x2{D3[e / 16][e % 16], D3[e / 16][e % 16 + 1]};
    const f32x2 ex = {__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
    den2 = den2 + ex;
    // e and e+1 are consecutive registers of one D1 tile (RPQ is even)
    num2 = pk_fma(ex, f32x2{D1[e / G::RPQ][R0 + e % G::RPQ], D1[e / G::RPQ][R0 + e % G::RPQ + 1]}, num2);
  }
  float den = den2.x + den2.y, num = num2.x + num2.y;
  den += xor32(den);
  num += xor32(num);
  // pi = ex/den, then the eval-time renormalisation pi / clamp(sum pi, 1e-6) (similarity_fn.py:42-46):
  // sum pi = den * (1/den) up to rounding
  const float rden = __builtin_amdgcn_rcpf(den);
  return (num * rden) / fmaxf(den * rden, 1e-6f);
}

// The exact-fp32 unit policy of the kernel shells (mol_score_shell.h).  SS: see gemm1.
template <int SS>
struct Fp32UnitImpl {
  static constexpr bool kIndexedCandidates = true;   // the indexed-candidate instantiation of the direct shell is built (mol_score_shell.h)
  template <class G>
  static constexpr int kLdsWeightFloats = G::kWpackFloats;
  template <class G, int NW>
  static __device__ __forceinline__ void stage(const ScoreArgs& p, float* smem) { stage_weights<G, NW>(p, smem); }
  template <class G, int PX, int DD, bool BULK = false, int PIPE = 0>
  static __device__ __forceinline__ void gemm1(f32x16 (&D1)[PX], const float* __restrict__ eq, const float4* tEx, int lane) {
    mol::gemm1<G, PX, DD, BULK, PIPE, SS>(D1, reinterpret_cast<const float4*>(eq), tEx, lane);
  }
  // All queries of one unit, each at its own static register offset (no register rotation).
  // `only` >= 0 restricts the unit to that query (per-row candidates).
  template <class G, int PX>
  static __device__ __forceinline__ void queries(f32x16 (&D1)[PX], const ScoreArgs& p, int g, int only, int64_t item0,
                                                 const float* smem, const float4* tGi, int lane, int hi, int x) {
    const float4* sW1 = reinterpret_cast<const float4*>(smem);
    const float4* sW2 = sW1 + G::kW1Floats / 4;
    const float* sB1 = smem + G::kW1Floats + G::kW2Floats;
    const float* sB2 = sB1 + G::TH * 32;
    [&]<int... Q>(std::integer_sequence<int, Q...>) {
      (
          [&] {
            const int q = g * G::QT + Q;
            if (q < p.B && (only < 0 || q == only)) {
              const float4* gq4 = reinterpret_cast<const float4*>(p.gqfrag + (int64_t)q * G::L + hi * G::E);
              const float out = query_mlp<G, PX, Q * G::RPQ, SS>(D1, sW1, sW2, sB1, sB2, tGi, gq4, lane, hi, p.combine_none);
              const int64_t item = item0 + x;
              if (hi == 0 && item < p.n_items) p.logits[(int64_t)q * p.ld + item] = out;
            }
          }(),
          ...);
    }(std::make_integer_sequence<int, G::QT>{});
  }
};
struct Fp32Unit : Fp32UnitImpl<64> {};       // operands in tile order (the index, gathered candidate tiles, LDS)
struct Fp32UnitRows : Fp32UnitImpl<2> {};    // candidates read from the row-major copy of the index (mol_score_rows_kernel)

}  // namespace mol
