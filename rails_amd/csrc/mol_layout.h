// Packed data layouts shared by the MoL kernels (gfx950, wave64, v_mfma_f32_32x32x2_f32).
//
// The scoring kernel chains three contractions per (query, item) pair
//     cl = <Eq, Ex>/tau  ->  hid = silu(W1 cl + b1)  ->  gqi = W2 hid + b2
// (reference: rails/similarities/mol/similarity_fn.py:389-405 and the Sequential built at
// modeling/similarity_utils.py:186-207) with ITEMS on the MFMA column axis (lane & 31) and the
// feature axis on the MFMA row axis.  The accumulator of one MFMA then IS the B operand of the
// next one: lane (x, hi) holds, in accumulator register r, the feature row
//     row(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi
// of item x, and a K=2 MFMA step consumes register r of both lane halves as k = {row(r,0), row(r,1)}.
// All weight / query / item operands are therefore stored pre-permuted ("fragment order") so that
// every operand fetch is one contiguous 16 B per lane and no data ever moves between lanes.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define MOL_HD __host__ __device__ __forceinline__
#else
#define MOL_HD inline
#endif

namespace mol {

// Operand pre-scaling (a contract between the packing kernels and the scoring kernel).  On gfx950 the fp32
// MFMA runs on the same ALUs as the VALU (measured: every v_fma_f32 next to it costs 4 more cycles, every
// v_exp_f32 8.5, at one or two waves per SIMD -- profiles/r01_ubench_mfma_f32_vs_valu.txt), so every VALU
// instruction removed from the scoring kernel is time saved.  Constants are therefore folded into operands:
//   EqFrag  = Eq / tau                      -> GEMM1 yields cl = <Eq,Ex>/tau directly
//   W1frag, b1frag scaled by -log2(e)       -> GEMM2 yields t = -log2e * pre, and
//                                              hid' = t * rcp(1 + exp2(t)) = -log2e * silu(pre)
//   W2frag unscaled, b2frag by -log2(e)     -> GEMM3 on hid' yields -log2e * gqi
//   gqfrag  = -log2e * gq                   -> t2 = fma(gq', gi, gqi') = -log2e * g, and
//                                              v = -t2 * rcp(1 + exp2(t2)) = log2e * g*sigmoid(g)
//   softmax(w) = exp2(v - max v) / sum
constexpr float kLog2e = 1.4426950408889634f;

// Precision mode "f16x3" (the first pass of the proved exact top-k; opt-in as a result-producing precision): ALL THREE contractions --
// GEMM1 included -- run on v_mfma_f32_32x32x16_f16 with every operand split into f16 hi + f16 lo (hi = round-toward-zero f16 of the
// value, lo = f16 of the exact fp32 remainder: RTZ in the kernels, RNE in the packs) and three MFMAs per product block (lo*hi, hi*lo,
// hi*hi; lo*lo ~ 2^-22 is dropped), accumulated in fp32: ~22 significant bits per product against fp32's 24, at 3/16 of the
// fp32-MFMA time and -- unlike fp32 MFMA -- overlapping with VALU work.  NO operand is rescaled: f16 subnormals are kept by the MFMA
// and by v_cvt_pkrtz, so every power-of-two scale is 1 (rounds 1-2 prescaled; the arithmetic model of rails_amd/f16x3_bound.py -- H2,
// the split bounds |x - hi - lo| <= 2^-20 |x| + 2^-24 -- describes THIS form).  The host refuses weights that could overflow f16
// (engine.py _check_f16_range: log2e * (||W1 row||_1 / tau + |b1|) < 60000 for |cl| <= 1/tau, dot_product_l2_norm = True).
// K = 16 per MFMA: lane half hi supplies 8 consecutive accumulator registers, so K-step s of GEMM2 covers cl registers e in [8s, 8s+8)
// and K-step s of GEMM3 covers hidden registers f in [8s, 8s+8) of both lane halves; weight fragments are stored in that order as
// [s][row tile][lane][8 x f16], hi and lo parts in separate halves of the pack; Eq and Ex arrive pre-split ([ks][hi|lo][lane] h8).

constexpr int kTileItems = 32;  // items per tile = MFMA column count

MOL_HD int acc_row(int reg, int hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * hi; }

// geometry derived from (P_Q, P_X, d, H)
template <int PQ, int PX, int DD, int H>
struct Geo {
  // a query's P_Q rows must cover both lane halves of the accumulator layout (rows 4..7 of every 8 sit in the upper half):
  // P_Q = 4 would put two queries into the same registers
  static_assert(PQ == 8 || PQ == 16 || PQ == 32, "P_Q must be 8, 16 or 32");
  static_assert(DD % 8 == 0, "dot_product_dimension must be a multiple of 8");
  static_assert(H % 32 == 0, "gate hidden dim must be a multiple of 32");
  static constexpr int L = PQ * PX;
  static_assert(L % 32 == 0, "P_Q * P_X must be a multiple of 32");
  static constexpr int QT = 32 / PQ;    // queries per query group (rows of one GEMM1 tile)
  static constexpr int RPQ = PQ / 2;    // accumulator registers per query inside one GEMM1 tile
  static constexpr int TH = H / 32;     // row tiles of the hidden layer
  static constexpr int TL = L / 32;     // row tiles of the gate output
  static constexpr int E = L / 2;       // K-steps over the logit axis (one per lane half pair)
  static constexpr int F = H / 2;       // K-steps over the hidden axis
  static constexpr int KS = DD / 2;     // K-steps of the sub-embedding contraction
  // floats
  static constexpr int kTileExFloats = kTileItems * PX * DD;
  static constexpr int kTileGiFloats = kTileItems * L;
  static constexpr int kTileFloats = kTileExFloats + kTileGiFloats;
  static constexpr int kEqGroupFloats = 32 * DD;  // one query group of Eq in fragment order
  // H == 0: a pair gate WITHOUT hidden layer (modeling/similarity_utils.py:199-206: one Linear(L, L)): its weights take the W1 slot
  // (K axis = logits, in cl's register order) with the gate-output rows in W2's row order, its bias the b2 slot; no W2, no b1
  static constexpr int kW1Floats = H > 0 ? H * L : L * L, kW2Floats = L * H;
  static constexpr int kWpackFloats = kW1Floats + kW2Floats + H + L;
};

// logit index l held by K-step e of lane half hi:  e = m * RPQ + r',  p = row(r', hi),  l = p * PX + m
MOL_HD int logit_of(int e, int hi, int PQ, int PX) {
  const int rpq = PQ / 2;
  const int m = e / rpq, r = e % rpq;
  return acc_row(r, hi) * PX + m;
}
// hidden index held by K-step f of lane half hi
MOL_HD int hidden_of(int f, int hi) { return 32 * (f / 16) + acc_row(f % 16, hi); }
// sub-embedding index held by K-step s of lane half hi
MOL_HD int kdim_of(int s, int hi, int DD) { return hi * (DD / 2) + s; }
// (register, half) that accumulator row i of a 32-row tile lands in
MOL_HD int reg_of_row(int i) { return (i & 3) + 4 * (i >> 3); }
MOL_HD int half_of_row(int i) { return (i >> 2) & 1; }

// ---- small-unit layout: v_mfma_f32_16x16x4_f32, unit = 2 queries x 16 items (mol_score_small.h; P_Q = 8 only) -----------------
// Accumulator register i (0..3) of lane group g = lane >> 4 holds row 4g + i of item column lane & 15; as the B operand of the
// next GEMM, register i is one K = 4 step with k = g.  fp32 MFMA is an fmaf chain in k order, and the small kernel must return the
// SAME BITS as the 32x32x2 kernels above (a shard of a corpus may take either shell), so every contraction visits its terms in the
// order the 32x32x2 layout does: K-step s of lane halves (0, 1) there  ==  lane groups (0, 1) and (2, 3) of half a K-step here.
// With P_Q = 8 the 32-layout visits the query groups of one item group m as p = 0, 4, 1, 5, 2, 6, 3, 7 and the hidden units of a
// block of 8 in the same pattern; so group g of K-step parity b holds
//     p16(b, g) = 2b + (g >> 1) + 4 (g & 1)
// and every packed 32-layout float4 (four K-steps of ONE lane half) carries, for lane group g, exactly its components
// (g >> 1) and (g >> 1) + 2 of the half (g & 1): the small kernel reads the item index and the query pack as they are.
MOL_HD int p16(int b, int g) { return 2 * b + (g >> 1) + 4 * (g & 1); }
// logit held by register pair index e16 = 2m + b (D1[m] register 2q + b of query q; D3 tile e16 / 4, register e16 % 4) of lane group g
MOL_HD int logit16(int e16, int g, int PX) { return p16(e16 & 1, g) * PX + (e16 >> 1); }
// hidden unit held by K-step kappa = 4t + i (D2 tile t, register i) of lane group g
MOL_HD int hidden16(int kappa, int g) { return 16 * (kappa >> 2) + 8 * ((kappa >> 1) & 1) + p16(kappa & 1, g); }

}  // namespace mol
