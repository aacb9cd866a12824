// Kernel D ("wsplit"): the fused MoL scoring pass for shapes whose logit axis is too long for one wave
// (16x16x64: L = 256 -- a unit's cross logits alone are 256 accumulator registers per lane, the pair-gate
// weights 256 KiB).  Reference arithmetic: rails/similarities/mol/similarity_fn.py:389-413 (gate :148-201,
// combiner :31-46), as in mol_score_fp32_unit.h.
//
// One workgroup = one TEAM of four waves (one per SIMD, 512 registers each) that works on one unit =
// (query group of 32/P_Q = 2 queries) x (tile of 32 items) at a time and splits the unit's LOGIT axis:
// wave w owns item groups m in [w P_X/4, (w+1) P_X/4), i.e. the K-steps e in [w E/4, (w+1) E/4) of both queries
// (mol_layout.h: e = m * RPQ + r'), hidden row tile w, and the gate-output row tiles of its own logits.
//
//   phase 1   GEMM1 of the wave's item groups (D1w, 64 registers, kept for the mixture) -> its cl values as B operands
//             into LDS                                                              -- barrier --
//   phase 2   GEMM2 of hidden row tile w over ALL logits (B operands from LDS), silu, the tile's hidden values as
//             B operands into LDS                                                   -- barrier --
//   phase 3   GEMM3 of the wave's own logit rows over all hidden units, gate, softmax numerators, and the wave's
//             partial (min, den, num) of every (query, item) into LDS; one wave folds the four partials and stores the
//             logits after the NEXT unit's first barrier (no third barrier).
//
// So every value crosses waves exactly once (all-gather of cl, all-gather of hid, 3 floats per pair), GEMM1 is
// computed once (the k-split kernel this replaces computed it twice), and no wave holds more than 64 + 64 accumulator
// registers.  What the freed registers buy: the wave's slices of W1 and W2 -- 32 hidden rows x L and its L/4 logit
// rows x H, 128 + 128 registers -- are loaded ONCE per launch and stay in registers as MFMA A operands
// (weight-stationary); per unit a wave reads only its 32 KiB slice of the tile's Ex (L2; the 16 query groups of a tile
// run on 16 workgroups of one XCD at the same time), 8 KiB of gi and the query group's 8 KiB.
//
// The arithmetic type is a policy P (exact fp32 below; the f16 hi/lo builds in mol_score_wsplit_f16.h):
//   P::CE                       accumulator registers (K-steps of two rows) covered by one operand chunk
//   P::Op                       one operand chunk of a lane; P::OPV 16-byte vectors
//   eq_op / ex_op / w1_op / w2_op   chunk loads from the query pack, the item tile, the gate pack (buffer loads)
//   mma_a<N>(D[N], A, B[N]) / mma_b<N>(D[N], A[N], B)   one chunk of N independent accumulators sharing the A / the B operand
//   pack<R0>(acc)               accumulator registers [R0, R0 + CE) -> B operand chunk
//   pin(op)                     put a stationary weight chunk into the AGPR half of the register file (MFMA A operands may be
//                               AGPRs; left to itself the allocator keeps them as VGPR "spills" there and copies at every use)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

#include "mol_kernels.h"
#include "mol_layout.h"
#include "mol_score_shell.h"

namespace mol {

typedef unsigned int ws_u32x4 __attribute__((ext_vector_type(4)));

// Buffer addressing (SGPR descriptor + scalar byte offset + ONE per-lane VGPR offset): the operands of a unit span up to
// 256 KiB per base; with flat addressing the compiler keeps a 64-bit address pair per fragment.
struct WsBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ WsBuf(const void* base, unsigned bytes)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000)) {}
  // 16-byte vector `lane` of the 1 KiB fragment `idx`
  __device__ __forceinline__ ws_u32x4 frag(int idx, int lane16) const {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, idx * 1024, 0);
  }
};

template <int N, class F>
__device__ __forceinline__ void ws_static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ float ws_xor32(float v) { return __shfl_xor(v, 32, 64); }

// Team barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding GLOBAL loads (its fence waits
// for vmcnt(0)): the next unit's operand prefetch, the gate fragments and the L2 touches would all have to land before every
// barrier -- at 6 k cycles per unit (one-product build) that is the HBM latency exposed twice per unit.
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef RAILS_WS_PHASES   // tools/wsplit_phases.sh: shader-clock stamps of workgroup 0 / wave 0, second unit
static __device__ long long g_ws_phase[16];
#define WS_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && it == 1) g_ws_phase[i] = (long long)clock64(); } while (0)
#else
#define WS_STAMP(i)
#endif

template <class P, int PQ, int PX, int DD, int H>
struct WsGeo {
  using G = Geo<PQ, PX, DD, H>;
  static constexpr int NWT = 4;
  static_assert(G::QT == 2, "the team kernel is laid out for two queries per group (P_Q = 16)");
  static_assert(G::TH == NWT, "one hidden row tile per wave: H = 128");
  static_assert(PX % NWT == 0 && G::TL % NWT == 0, "the logit axis splits four ways");
  static constexpr int MW = PX / NWT;          // item groups per wave
  static constexpr int EW = G::E / NWT;        // accumulator registers (logit K-steps) per wave and query
  static constexpr int TLW = G::TL / NWT;      // gate-output row tiles per wave
  static_assert(EW == MW * G::RPQ && EW == 16 * TLW, "a wave's logits = its item groups = its gate-output row tiles");
  static constexpr int CE = P::CE, OPV = P::OPV;
  static constexpr int NC1 = DD / (2 * CE);    // operand chunks of GEMM1 (K = d)
  static constexpr int NC2 = G::E / CE;        // of GEMM2 (K = L)
  static constexpr int NC3 = G::F / CE;        // of GEMM3 (K = H)
  static constexpr int NC2W = EW / CE;         // cl chunks a wave produces per query
  static constexpr int NC3W = 16 / CE;         // hidden chunks a wave produces per query
  static_assert(G::RPQ % CE == 0 || CE % G::RPQ == 0, "a chunk stays inside one D1 tile");
  // LDS, in floats
  static constexpr int kBiasFloats = H + G::L;
  static constexpr int kClFloats = G::QT * NC2 * OPV * 256;
  static constexpr int kHidFloats = G::QT * NC3 * OPV * 256;
  static constexpr int kGqFloats = NWT * G::QT * 2 * EW;     // [wave][query][lane half][EW]
  static constexpr int kGiFloats = NWT * (EW / 4) * 256;
  static constexpr int kPartFloats = NWT * 3 * 64;
  static constexpr int kLdsFloats = kBiasFloats + kClFloats + kHidFloats + kGqFloats + kGiFloats + kPartFloats;
};

// UPPER (f16x3 policy only; rails_mol_score_dense_upper): the kernel writes s + ub(c) instead of s, c = max_l |cl_l| of the pair and
// ub(c) = (ub2 c + ub1) c + ub0 the caller's per-pair bound on |s - fp32 logit| (rails_amd/f16x3_bound.py upper_bound_poly): an UPPER BOUND
// of the pair's fp32 logit.  Each wave folds |cl| over its own logits right after GEMM1 (32 v_max per unit) and leaves the partial in LDS,
// double-buffered by unit parity because the previous unit's partials are folded after this unit's first barrier.
template <class P, int PQ, int PX, int DD, int H, bool UPPER = false>
__global__ __launch_bounds__(256, 1) void mol_score_wsplit_kernel(ScoreArgs p) {
  using W = WsGeo<P, PQ, PX, DD, H>;
  using G = typename W::G;
  using Op = typename P::Op;
  constexpr int NWT = W::NWT, QT = G::QT, MW = W::MW, EW = W::EW, TLW = W::TLW, CE = W::CE, OPV = W::OPV;
  constexpr int NC1 = W::NC1, NC2 = W::NC2, NC3 = W::NC3, NC2W = W::NC2W, NC3W = W::NC3W;
  MOL_RUN_IF(p.run_if);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sB1 = smem;
  float* sB2 = smem + H;
  float4* sCl = reinterpret_cast<float4*>(smem + W::kBiasFloats);
  float4* sHid = sCl + W::kClFloats / 4;
  float4* sGq = sHid + W::kHidFloats / 4;
  float4* sGi = sGq + W::kGqFloats / 4;     // [wave][EW / 4][lane]: the wave's item-gate fragments of the current unit
  float* sPart = reinterpret_cast<float*>(sGi + W::kGiFloats / 4);
  [[maybe_unused]] float* sCmax = sPart + W::kPartFloats;   // UPPER: [unit parity][wave][lane]

  const int lane = threadIdx.x & 63;
  const int lane16 = lane * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  P pol;
  pol.init();

  {  // biases -> LDS (accumulator initial values)
    const float* src = p.wpack + G::kW1Floats + G::kW2Floats;
    for (int i = threadIdx.x; i < W::kBiasFloats; i += NWT * 64) smem[i] = src[i];
  }
  // this wave's weight slices -> registers, once per launch
  const WsBuf wb(p.wpack, (unsigned)(G::kWpackFloats * sizeof(float)));
  // (P::kW1Stream chunks of the W1 slice are NOT kept: a policy whose weights alone fill half the register file re-reads them
  // from L2 at the head of every phase 2, where nothing else is live, instead of letting the allocator spill them to scratch)
  constexpr int NC2R = NC2 - P::kW1Stream;
  Op w1r[NC2R], w2r[NC3][TLW];
#pragma unroll
  for (int c = 0; c < NC2R; ++c) { w1r[c] = P::template w1_op<G>(wb, c, wave, lane16); P::pin(w1r[c]); }
#pragma unroll
  for (int c = 0; c < NC3; ++c)
#pragma unroll
    for (int v = 0; v < TLW; ++v) { w2r[c][v] = P::template w2_op<G>(wb, c, wave * TLW + v, lane16); P::pin(w2r[c][v]); }
  __syncthreads();

  const int inner = p.per_row ? (int)p.n_tiles : p.n_groups;
  const unsigned n_units = (unsigned)(p.per_row ? (int64_t)p.B * p.n_tiles : p.n_tiles * p.n_groups);
  // XCD-aware numbering: hardware workgroup b runs on XCD b % 8 (each XCD has its own L2).  Consecutive units are the query
  // groups of one tile; logical id = (b % 8) * (grid / 8) + b / 8 keeps consecutive logical workgroups on one XCD, so the 16
  // groups of a tile read it from ONE L2.  Placement is a speed matter only.
  const unsigned stride = gridDim.x;
  const unsigned bx = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;

  struct Unit { int tile, g, row; };   // tile inside the row / corpus; row >= 0: per-row candidates, the only query of the unit
  auto decode = [&](unsigned u) {   // n_units < 2^31 (checked at launch): 32-bit division
    const unsigned outer = u / (unsigned)inner;
    const int innr = (int)(u - outer * (unsigned)inner);
    Unit r;
    r.tile = p.per_row ? innr : (int)outer;
    r.row = p.per_row ? (int)outer : -1;
    r.g = p.per_row ? r.row / QT : innr;
    return r;
  };
  auto tile_base = [&](const Unit& un) {
    const int64_t tile_addr = p.per_row ? (int64_t)un.row * p.n_tiles + un.tile : (int64_t)un.tile;
    return p.ipack + tile_addr * (int64_t)G::kTileFloats;
  };
  // the first PD1 operand chunks of a unit's GEMM1 are requested one phase ahead of their use
  constexpr int PD1 = P::PD1;
  static_assert(PD1 >= 1 && PD1 <= NC1, "GEMM1 prefetch distance in chunks");
  Op a_pf[PD1], b_pf[PD1][MW];
  auto prefetch_first = [&](const Unit& un) {
    const WsBuf tb(tile_base(un), (unsigned)(G::kTileFloats * sizeof(float)));
    const WsBuf eb(p.eqfrag + (int64_t)un.g * G::kEqGroupFloats, (unsigned)(G::kEqGroupFloats * sizeof(float)));
#pragma unroll
    for (int c = 0; c < PD1; ++c) {
      a_pf[c] = P::template eq_op<G, DD>(eb, c, lane16);
#pragma unroll
      for (int mm = 0; mm < MW; ++mm) b_pf[c][mm] = P::template ex_op<G, DD>(tb, wave * MW + mm, c, lane16);
    }
  };
  // L2 touch of a unit's Ex slice (this wave's item groups: MW * d / 8 KiB, contiguous) and gi slice (EW / 4 KiB), TWO units
  // ahead: one dword per 128-byte line, 8 KiB per instruction, into registers nobody reads.  Every unit works on a tile its
  // workgroup has not seen before, so without this each GEMM1 operand request is an HBM-latency miss.  (Not by LDS-DMA: the
  // compiler orders every later LDS read behind an LDS-DMA it cannot disambiguate.  Not at the head of phase 2 either: there the
  // 64-line requests slowed GEMM2's LDS operand reads by ~1.2 k cycles per unit in every precision.)
  // The touch loads are issued from inline asm into ONE register that stays reserved for the whole loop and is never read:
  // the compiler does not count them, so nothing ever waits for their HBM misses (a builtin load has to be "used" somewhere,
  // and that use became a wait of 1-2 k cycles per unit).  Uncounted loads only make the compiler's own vmcnt waits
  // conservative (results return in order).
  constexpr int NTOUCH = MW * (DD / 8) / 8 + 1;
  static_assert(EW / 4 == 8, "one touch instruction covers the wave's gi slice");
  float sink = 0.0f;
  auto touch = [&](const Unit& un) {
    const float* tb = tile_base(un);
    const float* slice = tb + (int64_t)wave * MW * (DD / 8) * 256 + lane * 32;
#pragma unroll
    for (int i = 0; i < NTOUCH - 1; ++i) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(slice + i * 2048) : "memory");
    asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(tb + G::kTileExFloats + wave * (EW / 4) * 256 + lane * 32) : "memory");
  };
  // the previous unit's output is folded and stored by ONE wave after the next barrier: lane = (query of the group, item)
  bool have_prev = false;
  Unit prev{};
  auto combine_store = [&](const Unit& un, [[maybe_unused]] int parity) {
    float mn[NWT], dn[NWT], nm[NWT];
#pragma unroll
    for (int w = 0; w < NWT; ++w) {
      mn[w] = sPart[(w * 3 + 0) * 64 + lane];
      dn[w] = sPart[(w * 3 + 1) * 64 + lane];
      nm[w] = sPart[(w * 3 + 2) * 64 + lane];
    }
    float m = mn[0];
#pragma unroll
    for (int w = 1; w < NWT; ++w) m = fminf(m, mn[w]);
    float den = 0.0f, num = 0.0f;
#pragma unroll
    for (int w = 0; w < NWT; ++w) {
      const float s = __builtin_amdgcn_exp2f(m - mn[w]);   // <= 1
      den = __builtin_fmaf(dn[w], s, den);
      num = __builtin_fmaf(nm[w], s, num);
    }
    // pi = ex/den, then the eval-time renormalisation pi / clamp(sum pi, 1e-6) (similarity_fn.py:42-46): sum pi = den * (1/den)
    const float rden = __builtin_amdgcn_rcpf(den);
    float out = (num * rden) / fmaxf(den * rden, 1e-6f);
    if constexpr (UPPER) {
      float c = sCmax[(parity * NWT) * 64 + lane];
#pragma unroll
      for (int w = 1; w < NWT; ++w) c = fmaxf(c, sCmax[(parity * NWT + w) * 64 + lane]);
      out += __builtin_fmaf(__builtin_fmaf(p.ub2, c, p.ub1), c, p.ub0);
    }
    const int q = un.g * QT + (lane >> 5);
    const int64_t item = (int64_t)un.tile * kTileItems + (lane & 31);
    if (q < p.B && (un.row < 0 || q == un.row) && item < p.n_items) p.logits[(int64_t)q * p.ld + item] = out;
  };

  unsigned u = bx;
  Unit cur{};
  if (u < n_units) {
    cur = decode(u);
    prefetch_first(cur);
  }
  [[maybe_unused]] int n_done = 0;
  for (int it = 0; u < n_units; u += stride, ++it, ++n_done) {
    const float* tile_ptr = tile_base(cur);
    const WsBuf tileb(tile_ptr, (unsigned)(G::kTileFloats * sizeof(float)));
    const WsBuf eqb(p.eqfrag + (int64_t)cur.g * G::kEqGroupFloats, (unsigned)(G::kEqGroupFloats * sizeof(float)));
    WS_STAMP(0);
    const unsigned un = u + stride;
    Unit nxt{};
    if (un < n_units) nxt = decode(un);

    // ---- phase 1: GEMM1 of this wave's item groups, both queries of the group
    f32x16 D1w[MW];
#pragma unroll
    for (int mm = 0; mm < MW; ++mm)
#pragma unroll
      for (int r = 0; r < 16; ++r) D1w[mm][r] = 0.0f;
    {
      Op a[PD1 + 1], b[PD1 + 1][MW];
#pragma unroll
      for (int c = 0; c < PD1; ++c) {
        a[c] = a_pf[c];
#pragma unroll
        for (int mm = 0; mm < MW; ++mm) b[c][mm] = b_pf[c][mm];
      }
      ws_static_for<NC1>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c + PD1 < NC1) {
          constexpr int slot = (c + PD1) % (PD1 + 1);
          a[slot] = P::template eq_op<G, DD>(eqb, c + PD1, lane16);
#pragma unroll
          for (int mm = 0; mm < MW; ++mm) b[slot][mm] = P::template ex_op<G, DD>(tileb, wave * MW + mm, c + PD1, lane16);
        }
        __builtin_amdgcn_sched_barrier(0);   // exactly PD1 chunks ahead: the requests stay ABOVE this chunk's MFMAs (a memory clobber alone lets the scheduler hoist MFMAs over them)
        P::template mma_a<MW>(D1w, a[c % (PD1 + 1)], b[c % (PD1 + 1)]);
      });
    }
    WS_STAMP(1);
    if constexpr (UPPER) {   // max |cl| over this wave's logits of (query = lane half, item): accumulator registers [Q RPQ, (Q + 1) RPQ) are query Q
      float cq[QT];
#pragma unroll
      for (int Q = 0; Q < QT; ++Q) {
        float c = 0.0f;
#pragma unroll
        for (int mm = 0; mm < MW; ++mm)
#pragma unroll
          for (int r = 0; r < G::RPQ; ++r) c = fmaxf(c, fabsf(D1w[mm][Q * G::RPQ + r]));
        cq[Q] = fmaxf(c, ws_xor32(c));
      }
      sCmax[((it & 1) * NWT + wave) * 64 + lane] = hi ? cq[1] : cq[0];
    }
    // this wave's cl values of both queries as B-operand chunks -> LDS (all-gather over the team)
    ws_static_for<QT * NC2W>([&](auto ic) {
      constexpr int I = decltype(ic)::value, Q = I / NC2W, c = I % NC2W;
      constexpr int e0 = c * CE;   // first wave-local K-step of the chunk
      const Op o = pol.template pack<Q * G::RPQ + e0 % G::RPQ>(D1w[e0 / G::RPQ]);
      P::st(sCl + ((Q * NC2 + wave * NC2W + c) * OPV) * 64, lane, o);
    });
    ws_barrier();   // B1: every wave's cl chunks are in LDS; every wave is done with the previous unit's hid and partials
    WS_STAMP(2);
    if (have_prev && wave == (it & 3)) combine_store(prev, (it - 1) & 1);
    // Memory requests that nothing waits for soon go HERE, behind the barrier: vector-memory results return in order, so in
    // phase 1 they sat in front of GEMM1's operand loads and every chunk waited for their HBM misses (GEMM1 at half rate).
    // Phase 2 reads LDS only.
    // this wave's item-gate fragments (shared by the two queries): requested here, parked in the wave's own LDS slot before B2
    // (32 registers that phase 3 needs for the second query's accumulators), read back slice by slice in phase 3
    float4 gi[EW / 4];
#pragma unroll
    for (int ec = 0; ec < EW / 4; ++ec) {
      const ws_u32x4 v = tileb.frag(G::kTileExFloats / 256 + wave * (EW / 4) + ec, lane16);
      gi[ec] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    // the unit's query-gate rows, this wave's logits: [query][lane half][EW] -> the wave's own LDS slot (read back by the
    // same wave only: DS operations of a wave are in order).  Rows past the batch end (padding of the last group) read the
    // last real row; their output is never stored.
    float4 gq_stage = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < QT * 2 * (EW / 4)) {
      const int qi = lane / (2 * (EW / 4)), hh = (lane / (EW / 4)) & 1, j = lane % (EW / 4);
      const int q = cur.g * QT + qi;
      const int qq = q < p.B ? q : p.B - 1;
      gq_stage = *reinterpret_cast<const float4*>(p.gqfrag + (int64_t)qq * G::L + hh * G::E + wave * EW + 4 * j);
    }
    // ---- phase 2: GEMM2 of hidden row tile `wave`, both queries:  D2 = -log2e * (b1 + W1 cl)
    Op w1s[P::kW1Stream > 0 ? P::kW1Stream : 1];
#pragma unroll
    for (int c = 0; c < P::kW1Stream; ++c) w1s[c] = P::template w1_op<G>(wb, NC2R + c, wave, lane16);
    f32x16 D2[QT];
#pragma unroll
    for (int Q = 0; Q < QT; ++Q)
#pragma unroll
      for (int r = 0; r < 16; ++r) D2[Q][r] = sB1[wave * 32 + hi * 16 + r];
    {
      constexpr int PD = P::PD2;
      Op ring[PD + 1][QT];
#pragma unroll
      for (int c = 0; c < PD; ++c)
#pragma unroll
        for (int Q = 0; Q < QT; ++Q) ring[c][Q] = P::ldl(sCl + ((Q * NC2 + c) * OPV) * 64, lane);
      ws_static_for<NC2>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c + PD < NC2) {
#pragma unroll
          for (int Q = 0; Q < QT; ++Q) ring[(c + PD) % (PD + 1)][Q] = P::ldl(sCl + ((Q * NC2 + c + PD) * OPV) * 64, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (c < NC2R) P::template mma_a<QT>(D2, w1r[c], ring[c % (PD + 1)]);
        else P::template mma_a<QT>(D2, w1s[c - NC2R], ring[c % (PD + 1)]);
      });
    }
    WS_STAMP(3);
    // hid' = t / (1 + 2^t) = -log2e * silu(pre), then the tile's hidden values as B-operand chunks -> LDS
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int Q = 0; Q < QT; ++Q) P::silu16(D2[Q]);
    ws_static_for<QT * NC3W>([&](auto ic) {
      constexpr int I = decltype(ic)::value, Q = I / NC3W, c = I % NC3W;
      const Op o = pol.template pack<c * CE>(D2[Q]);
      P::st(sHid + ((Q * NC3 + wave * NC3W + c) * OPV) * 64, lane, o);
    });
    if (lane < QT * 2 * (EW / 4)) sGq[wave * (QT * 2 * (EW / 4)) + lane] = gq_stage;
#pragma unroll
    for (int ec = 0; ec < EW / 4; ++ec) sGi[(wave * (EW / 4) + ec) * 64 + lane] = gi[ec];
    __builtin_amdgcn_sched_barrier(0);
    ws_barrier();   // B2: the whole hidden layer of both queries is in LDS; every wave is done with the cl chunks
    WS_STAMP(4);

    // the unit after next: L2 touch
    if (un + stride < n_units) touch(decode(un + stride));

    // ---- phase 3: GEMM3 of the wave's logit rows  D3 = -log2e * (b2 + W2 hid), gate, softmax numerators.
    //   GEMM3(query 0)  ->  GEMM3(query 1) || gate + softmax of query 0, dealt slice by slice between the MFMA chunks
    //   ->  gate + softmax of query 1
    float pmn[QT], pden[QT], pnum[QT];
    f32x16 D3[QT][TLW];
    using EpiT = typename P::template Epi<G, MW, TLW, EW>;
    constexpr int PD = P::PD3;
    Op ring3[PD + 1];
    auto gemm3_begin = [&](auto qc) {
      constexpr int Q = decltype(qc)::value;
#pragma unroll
      for (int v = 0; v < TLW; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) D3[Q][v][r] = sB2[hi * G::E + wave * EW + v * 16 + r];
#pragma unroll
      for (int c = 0; c < PD; ++c) ring3[c] = P::ldl(sHid + ((Q * NC3 + c) * OPV) * 64, lane);
    };
    auto gemm3_chunk = [&](auto qc, auto cc) {
      constexpr int Q = decltype(qc)::value, c = decltype(cc)::value;
      if constexpr (c + PD < NC3) ring3[(c + PD) % (PD + 1)] = P::ldl(sHid + ((Q * NC3 + c + PD) * OPV) * 64, lane);
      __builtin_amdgcn_sched_barrier(0);
      P::template mma_b<TLW>(D3[Q], w2r[c], ring3[c % (PD + 1)]);
    };
    auto epi_args = [&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      // the wave's own cl chunks are still in LDS (other waves only read them in phase 2 and never write them): a policy may
      // re-read them there instead of keeping D1w alive through phases 2 and 3
      struct A { const float4* gq4; const float4* cl_own; const float4* gi_own; };
      return A{sGq + wave * (QT * 2 * (EW / 4)) + (Q * 2 + hi) * (EW / 4), sCl + ((Q * NC2 + wave * NC2W) * OPV) * 64 + lane,
               sGi + wave * (EW / 4) * 64 + lane};
    };
    {
      using Q0 = std::integral_constant<int, 0>;
      using Q1 = std::integral_constant<int, 1>;
      gemm3_begin(Q0{});
      ws_static_for<NC3>([&](auto cc) { gemm3_chunk(Q0{}, cc); });
      __builtin_amdgcn_sched_barrier(0);
      WS_STAMP(7);
      EpiT ep;
      const auto a0 = epi_args(Q0{});
      ep.begin(a0.cl_own, a0.gi_own, a0.gq4);
      gemm3_begin(Q1{});
      ws_static_for<NC3>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        gemm3_chunk(Q1{}, cc);
        constexpr int s0 = c * EpiT::NS / NC3, s1 = (c + 1) * EpiT::NS / NC3;
        ws_static_for<s1 - s0>([&](auto sc) { ep.template slice<0, s0 + decltype(sc)::value>(D3[0], D1w, a0.cl_own, a0.gi_own, a0.gq4, p.combine_none); });
        __builtin_amdgcn_sched_barrier(0);
      });
      WS_STAMP(8);
      ep.template end<0>(D3[0], D1w, a0.cl_own, pmn[0], pden[0], pnum[0]);
      __builtin_amdgcn_sched_barrier(0);
      WS_STAMP(9);
      // next unit: request the first GEMM1 chunks now (L2 hits: touched a unit ago), behind the register peak of the overlapped
      // part and one gate + softmax pass ahead of their use
      if (un < n_units) prefetch_first(nxt);
      const auto a1 = epi_args(Q1{});
      ep.begin(a1.cl_own, a1.gi_own, a1.gq4);
      ws_static_for<EpiT::NS>([&](auto sc) {
        ep.template slice<1, decltype(sc)::value>(D3[1], D1w, a1.cl_own, a1.gi_own, a1.gq4, p.combine_none);
        __builtin_amdgcn_sched_barrier(0);   // slices keep their own instruction order (they are laid out for dependent-issue distance)
      });
      ep.template end<1>(D3[1], D1w, a1.cl_own, pmn[1], pden[1], pnum[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    WS_STAMP(5);
    // partial softmax state of (query = lane half, item) -> LDS; both lane halves hold the wave's totals
    sPart[(wave * 3 + 0) * 64 + lane] = hi ? pmn[1] : pmn[0];
    sPart[(wave * 3 + 1) * 64 + lane] = hi ? pden[1] : pden[0];
    sPart[(wave * 3 + 2) * 64 + lane] = hi ? pnum[1] : pnum[0];
    prev = cur;
    have_prev = true;
    cur = nxt;
    WS_STAMP(6);
  }
  if (have_prev) {
    __syncthreads();   // also drains the uncounted touch loads before the wave ends
    if (wave == 0) combine_store(prev, (n_done - 1) & 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(sink));
}

template <class P, int PQ, int PX, int DD, int H, bool UPPER = false>
static int launch_wsplit(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using W = WsGeo<P, PQ, PX, DD, H>;
  constexpr size_t lds = ((size_t)W::kLdsFloats + (UPPER ? 2 * W::NWT * 64 : 0)) * sizeof(float);
  static_assert(lds <= 160 * 1024, "exchange buffers must fit LDS");
  if (a.cand_pos) { set_error("indexed candidates are not available for the 256-logit team kernel (gather them: rails_mol_index_gather)"); return kErrUnsupported; }
  if (a.dry_run) return kOk;
  static DynLdsOnce once;
  if (ensure_dyn_lds(once, reinterpret_cast<const void*>(&mol_score_wsplit_kernel<P, PQ, PX, DD, H, UPPER>), (int)lds) != kOk) return kErrLaunch;
  const int64_t n_units = a.per_row ? (int64_t)a.B * a.n_tiles : a.n_tiles * a.n_groups;
  if (n_units >= (1LL << 31)) { set_error("scoring launch of %lld units: split the corpus", (long long)n_units); return kErrInvalid; }
  int64_t grid = n_units < n_cu ? n_units : n_cu;
  if (grid < 1) return kOk;
  hipLaunchKernelGGL((mol_score_wsplit_kernel<P, PQ, PX, DD, H, UPPER>), dim3((unsigned)grid), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace mol
