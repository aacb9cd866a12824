// Small-unit shell of the fused MoL scoring pass (exact fp32): v_mfma_f32_16x16x4_f32, one wave = 2 queries x 16 items,
// four waves per SIMD.
//
// Same arithmetic as mol_score_fp32_unit.h (reference: rails/similarities/mol/similarity_fn.py:389-413, gate :148-201, combiner
// :31-46; the top-k modules just call the module, rails/indexing/mol_top_k.py:118-130) on units a quarter the size.  Why: the
// 32x32x2 unit (4 queries x 32 items, 252-256 registers, two waves per SIMD) is right for B = 32 against 695 k items and wrong
// for small corpora and few queries --
//   * ML-1M (3 883 items) is 976 units on 1 024 SIMDs: one unit per SIMD and nothing to run under its load latency, its VALU
//     phases or its tail; ML-20M is 3.3 rounds of 256 workgroups;
//   * at B = 1 GEMM1 runs 32-row tiles with 8 real rows, and one or two waves per SIMD do not keep enough loads in flight to
//     stream the index.
// Here a unit is 2 queries x 16 items: D1 (P_X x 4) + D2 (8 x 4) + D3 (P_X/2 x 4) = 80 accumulator registers for 8x8x32, so a
// SIMD holds four waves (<= 128 registers); four times the units balance the tail, GEMM1's padding at B = 1 halves, and three
// other waves cover a wave's loads.
//
// Register chaining as in mol_layout.h: accumulator register i of lane group g = lane >> 4 is row 4g + i of item lane & 15 and, fed
// back as a B operand, one K = 4 step with k = g.  The rows of every GEMM are permuted (in the packed weights and in the way this
// kernel addresses the query pack) so that each contraction visits its terms in the order the 32x32x2 kernels do: the results are
// the same BITS (tests/test_gpu_parity.py compares the two shells with torch.equal), which is what lets a corpus shard take
// whichever shell is faster without changing any returned score.
//
// Operands.  Pair-gate weights: the second half of the gate pack (pack_gate16_kernel, mol_index.hip), copied to LDS once per
// workgroup, one ds_read_b128 = four K-steps of one row tile.  Item tiles and query fragments: read straight from the 32-layout
// buffers -- every packed float4 there (four K-steps of one lane half) holds, for lane group g, its components (g >> 1) and
// (g >> 1) + 2 of the half (g & 1); slots are fetched in pairs, one coalesced 16-byte load per lane and two half-wave swaps (ld_pair).  Waves are independent (no barrier after the weight copy): consecutive waves take the
// query pairs of one 16-item half tile, so its fragments are shared through L1 / L2.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "mol_kernels.h"
#include "mol_layout.h"

#ifndef RAILS_SMALL_ABL
#define RAILS_SMALL_ABL 0
#endif
#ifndef RAILS_SMALL_NW
#define RAILS_SMALL_NW 8      // waves per workgroup
#endif
#ifndef RAILS_SMALL_WGCU
#define RAILS_SMALL_WGCU 2    // workgroups per CU: NW * WGCU / 4 waves per SIMD
#endif
#ifndef RAILS_SMALL_ASM
#define RAILS_SMALL_ASM 1     // 1: the unit's operand loads are pinned inline-asm requests with hand-counted vmcnt waits
#endif
#ifndef RAILS_SMALL_HD
#define RAILS_SMALL_HD 0      // GEMM1 load rounds in flight; 0: what fits ~40 registers (8x8x32: 1, P_X = 4: 2)
#endif
#ifndef RAILS_SMALL_IL
#define RAILS_SMALL_IL 2
#endif

namespace mol {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// The lane's share of TWO packed 32-layout float4 slots A and B in one coalesced 16-byte load: lanes 0-31 (c = 0) fetch slot A,
// lanes 32-63 (c = 1) slot B of the same (lane half, item); two v_permlane32_swap (lanes l <-> l + 32 exchange one register each)
// leave every lane with components c and c + 2 of both:  .x = A[c]  .z = A[c + 2]  .y = B[c]  .w = B[c + 2].
// (First version: two 4-byte loads per slot and lane -- 16-byte-strided dwords, four times the L1 transactions per useful byte;
// ML-20M sat at 0.305 ms whatever else changed.)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 swap_pair(const float4 v) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.x), __float_as_uint(v.y), false, false);
  const u32x2 s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.z), __float_as_uint(v.w), false, false);
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(s.x), __uint_as_float(s.y));
}
// one 16-byte vector load (a float4 struct load may be split into four dword loads -- 16-byte-strided lanes, a quarter of the L1 rate)
__device__ __forceinline__ float4 ld16(const float4* __restrict__ p) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 ld_pair(const float4* __restrict__ p) { return swap_pair(ld16(p)); }

// Loads the compiler cannot move (RAILS_SMALL_ASM): left to itself it sinks a unit's operand loads towards their uses -- and, vmcnt
// being in order, puts late requests in front of waits for early ones -- so a unit pays two or three memory round trips where one
// would do.  These are issued exactly where written; vm_wait<N>() waits until at most N requests are outstanding and ties the
// registers it covers (a use cannot be scheduled above it).  The compiler's own loads (query fragments, gate weights via LDS) only
// ever make either side wait LONGER than needed (vmcnt counts every request of the wave), never too little.
__device__ __forceinline__ void ld16_pinned(f32x4& v, const float4* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
}
template <int N, int CNT>
__device__ __forceinline__ void vm_wait(f32x4 (&v)[CNT]) {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
  for (int i = 0; i < CNT; ++i) asm volatile("" : "+v"(v[i]));
}
__device__ __forceinline__ float4 swap_pair4(const f32x4 v) { return swap_pair(make_float4(v[0], v[1], v[2], v[3])); }

__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_sig(f32x2 t) {  // 1 / (1 + 2^t), the ops of pk_sigmoid_arg (mol_score_fp32_unit.h)
  f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  e = e + 1.0f;
  return f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}

template <int PX, int DD, int H>
struct Geo16 {
  static constexpr int L = 8 * PX;
  static constexpr int TH = H / 16;   // row tiles of the hidden layer
  static constexpr int TL = L / 16;   // row tiles of the gate output
  static constexpr int KC = DD / 8;   // packed float4 chunks (= two K-steps here) of the sub-embedding contraction
  static constexpr int kW1Floats = H * L, kW2Floats = L * H;
  static constexpr int kPackFloats = kW1Floats + kW2Floats + H + L;
  static constexpr int kTileExFloats = 32 * PX * DD, kTileGiFloats = 32 * L, kTileFloats = kTileExFloats + kTileGiFloats;
  static_assert(PX % 2 == 0 && DD % 16 == 0 && H % 16 == 0, "small-unit geometry");
};

// One query of the unit: gate MLP, combine, softmax, mixture.  Its cl values are registers 2Q, 2Q + 1 of every D1 tile.
template <class G, int PX, int Q>
__device__ __forceinline__ float query_mlp16(const f32x4 (&D1)[PX], const float4* sW1, const float4* sW2, const float4* sB1, const float4* sB2,
                                             const float4 (&gi)[PX / 2], const float4* __restrict__ gq, int lane, int g, int combine_none) {
  // GEMM2: t[h, x] = -log2e (b1[h] + sum_l W1[h, l] cl[l, x])
  f32x4 D2[G::TH];
#pragma unroll
  for (int t = 0; t < G::TH; ++t) {
    const float4 b = sB1[t * 4 + g];
    D2[t] = f32x4{b.x, b.y, b.z, b.w};
  }
  // v_mfma_f32_16x16x4_f32 issues every 32 cycles but a dependent one (same accumulator) only after 40: consecutive MFMAs go to
  // IL different accumulators
  constexpr int IL = RAILS_SMALL_IL;
#pragma unroll
  for (int mc = 0; mc < PX / 2; ++mc) {
#pragma unroll
    for (int t0 = 0; t0 < G::TH; t0 += IL) {
      float4 a[IL];
#pragma unroll
      for (int i = 0; i < IL; ++i) a[i] = sW1[(mc * G::TH + t0 + i) * 64 + lane];
#pragma unroll
      for (int i = 0; i < IL; ++i) D2[t0 + i] = mfma16(a[i].x, D1[2 * mc][2 * Q], D2[t0 + i]);
#pragma unroll
      for (int i = 0; i < IL; ++i) D2[t0 + i] = mfma16(a[i].y, D1[2 * mc][2 * Q + 1], D2[t0 + i]);
#pragma unroll
      for (int i = 0; i < IL; ++i) D2[t0 + i] = mfma16(a[i].z, D1[2 * mc + 1][2 * Q], D2[t0 + i]);
#pragma unroll
      for (int i = 0; i < IL; ++i) D2[t0 + i] = mfma16(a[i].w, D1[2 * mc + 1][2 * Q + 1], D2[t0 + i]);
    }
  }
  // hid' = t / (1 + 2^t) = -log2e silu(pre)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < G::TH; ++t)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const f32x2 tv = {D2[t][r], D2[t][r + 1]};
#if RAILS_SMALL_ABL & 1   // timing ablation only: wrong results
      const f32x2 h = tv;
#else
      const f32x2 h = tv * pk_sig(tv);
#endif
      D2[t][r] = h.x;
      D2[t][r + 1] = h.y;
    }
  __builtin_amdgcn_sched_barrier(0);
  // GEMM3: gqi'[l, x] = -log2e (b2[l] + sum_h W2[l, h] hid[h, x])
  f32x4 D3[G::TL];
#pragma unroll
  for (int v = 0; v < G::TL; ++v) {
    const float4 b = sB2[v * 4 + g];
    D3[v] = f32x4{b.x, b.y, b.z, b.w};
  }
  constexpr int IL3 = IL < G::TL ? IL : G::TL;
#pragma unroll
  for (int t = 0; t < G::TH; ++t) {
#pragma unroll
    for (int v0 = 0; v0 < G::TL; v0 += IL3) {
      float4 a[IL3];
#pragma unroll
      for (int i = 0; i < IL3; ++i) a[i] = sW2[(t * G::TL + v0 + i) * 64 + lane];
#pragma unroll
      for (int i = 0; i < IL3; ++i) D3[v0 + i] = mfma16(a[i].x, D2[t][0], D3[v0 + i]);
#pragma unroll
      for (int i = 0; i < IL3; ++i) D3[v0 + i] = mfma16(a[i].y, D2[t][1], D3[v0 + i]);
#pragma unroll
      for (int i = 0; i < IL3; ++i) D3[v0 + i] = mfma16(a[i].z, D2[t][2], D3[v0 + i]);
#pragma unroll
      for (int i = 0; i < IL3; ++i) D3[v0 + i] = mfma16(a[i].w, D2[t][3], D3[v0 + i]);
    }
  }
  // epilogue: register pair index e16 = 2m + b  <->  D3[m / 2][2 (m & 1) + b], cl = D1[m][2Q + b]; the lane's gi / gq values of item
  // group m are components (g >> 1) and (g >> 1) + 2 of the packed float4 of lane half (g & 1)
  __builtin_amdgcn_sched_barrier(0);
  float mn = INFINITY;
#pragma unroll
  for (int mm = 0; mm < PX / 2; ++mm) {
    // item groups 2mm (x, z) and 2mm + 1 (y, w): lane half c fetched the slot of group 2mm + c
    const float4 gi4 = swap_pair(gi[mm]);   // raw until here: nothing before the epilogue waits for the gate rows
    const float4 gq4 = ld_pair(gq + 2 * mm);
    const float giv[2][2] = {{gi4.x, gi4.z}, {gi4.y, gi4.w}}, gqv[2][2] = {{gq4.x, gq4.z}, {gq4.y, gq4.w}};
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      // m = 2mm + o: D3[m / 2][2 (m & 1) + b] = D3[mm][2o + b]
      if (combine_none) {
        const float u0 = __builtin_fmaf(giv[o][0], -kLog2e, gqv[o][0] + D3[mm][2 * o]);
        const float u1 = __builtin_fmaf(giv[o][1], -kLog2e, gqv[o][1] + D3[mm][2 * o + 1]);
        D3[mm][2 * o] = u0;
        D3[mm][2 * o + 1] = u1;
        mn = fminf(mn, fminf(u0, u1));
      } else {
        const f32x2 t2 = pk_fma2(f32x2{gqv[o][0], gqv[o][1]}, f32x2{giv[o][0], giv[o][1]}, f32x2{D3[mm][2 * o], D3[mm][2 * o + 1]});
#if RAILS_SMALL_ABL & 2
        const f32x2 u = t2;
#else
        const f32x2 u = t2 * pk_sig(t2);
#endif
        D3[mm][2 * o] = u.x;
        D3[mm][2 * o + 1] = u.y;
        mn = fminf(mn, fminf(u.x, u.y));
      }
    }
  }
  mn = fminf(mn, __shfl_xor(mn, 16, 64));
  mn = fminf(mn, __shfl_xor(mn, 32, 64));
  // the sums run in register order per lane group: the chain of the 32-layout's packed accumulator halves (x: groups 0 / 1, y: 2 / 3)
  float den = 0.0f, num = 0.0f;
#pragma unroll
  for (int m = 0; m < PX; ++m)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#if RAILS_SMALL_ABL & 2
      const float ex = mn - D3[m / 2][2 * (m & 1) + b];
#else
      const float ex = __builtin_amdgcn_exp2f(mn - D3[m / 2][2 * (m & 1) + b]);
#endif
      den = den + ex;
      num = __builtin_fmaf(ex, D1[m][2 * Q + b], num);
    }
  den += __shfl_xor(den, 32, 64);   // x + y of one lane half
  num += __shfl_xor(num, 32, 64);
  den += __shfl_xor(den, 16, 64);   // the two lane halves
  num += __shfl_xor(num, 16, 64);
  const float rden = __builtin_amdgcn_rcpf(den);
  return (num * rden) / fmaxf(den * rden, 1e-6f);
}

template <int PX, int DD, int H, int NW>
__global__ __launch_bounds__(NW * 64, NW * RAILS_SMALL_WGCU / 4) void mol_score_small_kernel(ScoreArgs p) {
  using G = Geo16<PX, DD, H>;
  MOL_RUN_IF(p.run_if);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    // The 16-layout half of the gate pack -> LDS by LDS-DMA (1 KiB per wave-instruction, no registers), NOT waited for here: the
    // first unit's GEMM1 needs no weights, so the copy's round trip runs beside the first tile's instead of in front of it (small
    // corpora are a single round of units: every microsecond of prologue is on the critical path).  The barrier follows GEMM1.
    const float* src = p.wpack + G::kPackFloats;
    constexpr int kPieces = G::kPackFloats / 256;
    for (int piece = wave; piece < kPieces; piece += NW)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(smem + piece * 256), 16, 0, 0);
    for (int i = kPieces * 256 + threadIdx.x; i < G::kPackFloats; i += NW * 64) smem[i] = src[i];
  }
  bool staged = false;
  const float4* sW1 = reinterpret_cast<const float4*>(smem);
  const float4* sW2 = sW1 + G::kW1Floats / 4;
  const float4* sB1 = sW2 + G::kW2Floats / 4;
  const float4* sB2 = sB1 + H / 4;

  const int g = lane >> 4, j = lane & 15;
  const int hi = g & 1, c = g >> 1;
  // A operand of GEMM1: lane (g, rho = 4 go + i) supplies row rho = (query i >> 1, p16(i & 1, go)) at k = g
  const int a_row = (j & 3) >> 1, a_p = p16(j & 1, j >> 2);
  const int n_pairs = (p.B + 1) / 2;
  const int64_t n_half = (p.n_items + 15) / 16;
  const int64_t n_units = n_half * n_pairs;
  // consecutive waves = the query pairs of one half tile; XCD-aware numbering and the wave-major leftover round as in the
  // independent-wave shell of mol_score_shell.h
  const int64_t stride = (int64_t)gridDim.x * NW;
  const int64_t rounds = n_units / stride;
  const int64_t bx = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  for (int64_t it = 0; it <= rounds; ++it) {
    const int64_t u = it < rounds ? it * stride + bx * NW + wave : rounds * stride + (int64_t)wave * gridDim.x + bx;
    if (u >= n_units) break;
    const int64_t half = u / n_pairs;
    const int pair = (int)(u - half * n_pairs);
    const int hsel = (int)(half & 1);
    // the lane's float4 inside every packed slot of the tile: lane half (g & 1), item 16 hsel + j; lanes of c = 1 sit one slot further
    // (ld_pair: slots come in pairs)
    const float4* tEx = reinterpret_cast<const float4*>(p.ipack + (half >> 1) * (int64_t)G::kTileFloats) + (hi * 32 + 16 * hsel + j) + c * 64;
    const float4* tGi = tEx + G::kTileExFloats / 4;
    const float4* eq = reinterpret_cast<const float4*>(p.eqfrag + (int64_t)(pair >> 1) * (32 * DD)) + (hi * 32 + (2 * (pair & 1) + a_row) * 8 + a_p) + c * 64;

    // GEMM1.  Rounds of two packed chunks (four K-steps) per item group; the loads of round r + HD are issued when round r's
    // registers have been consumed (HD rounds in flight), the item gate rows of the unit behind the last round's.
#if RAILS_SMALL_ASM
    // every round (up to ~80 registers of them) requested up front, the item gate rows into the first registers a round frees
    constexpr int R = G::KC / 2, NL = PX + 1, HD = (80 / (4 * NL) < R ? 80 / (4 * NL) : R), NG = PX / 2;
    f32x4 ring[HD][NL], gir[NG];
    float4 gi[NG];
#pragma unroll
    for (int r = 0; r < HD; ++r) {
      ld16_pinned(ring[r][0], eq + 2 * r * 64);
#pragma unroll
      for (int m = 0; m < PX; ++m) ld16_pinned(ring[r][1 + m], tEx + (m * G::KC + 2 * r) * 64);
    }
    f32x4 D1[PX];
#pragma unroll
    for (int m = 0; m < PX; ++m) D1[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < R; ++r) {
      constexpr int dummy = 0; (void)dummy;
      // requests issued after round r's: rounds r + 1 .. min(r + HD - 1, R - 1), and the gate rows once round R - HD has been consumed
      const int last = r + HD - 1 < R - 1 ? r + HD - 1 : R - 1;
      const int newer = (last - r) * NL + (r > R - HD ? NG : 0);
      switch (newer) {   // vmcnt takes an immediate: `newer` is a compile-time value of the unrolled loop
#define RAILS_VMW(n) case n: vm_wait<n, NL>(ring[r % HD]); break;
        RAILS_VMW(0) RAILS_VMW(2) RAILS_VMW(4) RAILS_VMW(5) RAILS_VMW(7) RAILS_VMW(9) RAILS_VMW(10) RAILS_VMW(12) RAILS_VMW(13) RAILS_VMW(15) RAILS_VMW(17) RAILS_VMW(18) RAILS_VMW(20)
        RAILS_VMW(22) RAILS_VMW(27)
#undef RAILS_VMW
        default: vm_wait<0, NL>(ring[r % HD]); break;
      }
      const float4 a = swap_pair4(ring[r % HD][0]);
      float4 b[PX];
#pragma unroll
      for (int m = 0; m < PX; ++m) b[m] = swap_pair4(ring[r % HD][1 + m]);
      if (r + HD < R) {
        ld16_pinned(ring[r % HD][0], eq + 2 * (r + HD) * 64);
#pragma unroll
        for (int m = 0; m < PX; ++m) ld16_pinned(ring[r % HD][1 + m], tEx + (m * G::KC + 2 * (r + HD)) * 64);
      } else if (r == R - HD) {
#pragma unroll
        for (int mm = 0; mm < NG; ++mm) ld16_pinned(gir[mm], tGi + 2 * mm * 64);
      }
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a.x, b[m].x, D1[m]);
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a.z, b[m].z, D1[m]);
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a.y, b[m].y, D1[m]);
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a.w, b[m].w, D1[m]);
    }
#pragma unroll
    for (int m = 0; m < PX; ++m) asm volatile("" : "+v"(D1[m]));   // the gate rows' wait stays behind GEMM1's last MFMAs (asm statements keep their order)
    vm_wait<0, NG>(gir);
#pragma unroll
    for (int mm = 0; mm < NG; ++mm) gi[mm] = make_float4(gir[mm][0], gir[mm][1], gir[mm][2], gir[mm][3]);
#else
    constexpr int R = G::KC / 2, kHdFit = 40 / (4 * (PX + 1)) > 0 ? 40 / (4 * (PX + 1)) : 1;   // rounds in flight that fit ~40 registers
    constexpr int kHdWant = RAILS_SMALL_HD > 0 ? RAILS_SMALL_HD : kHdFit, HD = kHdWant < R ? kHdWant : R;
    float4 ra[HD], rb[HD][PX], gi[PX / 2];
#pragma unroll
    for (int r = 0; r < HD; ++r) {
      ra[r] = ld16(eq + 2 * r * 64);
#pragma unroll
      for (int m = 0; m < PX; ++m) rb[r][m] = ld16(tEx + (m * G::KC + 2 * r) * 64);
    }
    f32x4 D1[PX];
#pragma unroll
    for (int m = 0; m < PX; ++m) D1[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < R; ++r) {   // chunks 2r (x, z) and 2r + 1 (y, w): four K-steps
      const float4 a = swap_pair(ra[r % HD]);
      float4 b[PX];
#pragma unroll
      for (int m = 0; m < PX; ++m) b[m] = swap_pair(rb[r % HD][m]);
      if (r + HD < R) {
        ra[r % HD] = ld16(eq + 2 * (r + HD) * 64);
#pragma unroll
        for (int m = 0; m < PX; ++m) rb[r % HD][m] = ld16(tEx + (m * G::KC + 2 * (r + HD)) * 64);
      }
      if (r == R - 1) {   // the item gate rows, behind the last round's requests: in flight under this round and the first query's MLP
#pragma unroll
        for (int mm = 0; mm < PX / 2; ++mm) gi[mm] = ld16(tGi + 2 * mm * 64);
      }
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a.x, b[m].x, D1[m]);
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a.z, b[m].z, D1[m]);
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a.y, b[m].y, D1[m]);
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = mfma16(a.w, b[m].w, D1[m]);
    }
#endif
    if (!staged) {   // wave-uniform; every wave of the workgroup passes exactly one of the two barriers
      __syncthreads();   // drains this wave's DMA pieces (vmcnt(0)) and meets the others
      staged = true;
    }
    const int64_t item = half * 16 + j;
    const int q0 = 2 * pair;
    {
      const float out = query_mlp16<G, PX, 0>(D1, sW1, sW2, sB1, sB2, gi, reinterpret_cast<const float4*>(p.gqfrag + (int64_t)q0 * G::L + hi * (G::L / 2)) + c, lane, g, p.combine_none);
      if (g == 0 && item < p.n_items) p.logits[(int64_t)q0 * p.ld + item] = out;
    }
    if (q0 + 1 < p.B) {
      const float out = query_mlp16<G, PX, 1>(D1, sW1, sW2, sB1, sB2, gi, reinterpret_cast<const float4*>(p.gqfrag + (int64_t)(q0 + 1) * G::L + hi * (G::L / 2)) + c, lane, g, p.combine_none);
      if (g == 0 && item < p.n_items) p.logits[(int64_t)(q0 + 1) * p.ld + item] = out;
    }
  }
  if (!staged) __syncthreads();   // a wave without a unit still meets its workgroup
}

template <int PX, int DD, int H>
int launch_small(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using G = Geo16<PX, DD, H>;
  constexpr int NW = RAILS_SMALL_NW, kWgPerCu = RAILS_SMALL_WGCU;
  constexpr size_t lds = (size_t)G::kPackFloats * sizeof(float);
  static_assert(kWgPerCu * lds <= 160 * 1024, "workgroups per CU");
  if (a.per_row || a.cand_pos || a.split) { set_error("the small-unit kernel scores a shared corpus densely in fp32 only"); return kErrUnsupported; }
  if (a.dry_run) return kOk;
  const int64_t n_units = ((a.n_items + 15) / 16) * ((a.B + 1) / 2);
  int64_t grid = (n_units + NW - 1) / NW;
  if (grid > kWgPerCu * (int64_t)n_cu) grid = kWgPerCu * (int64_t)n_cu;
  if (grid < 1) return kOk;
  auto kernel = &mol_score_small_kernel<PX, DD, H, NW>;
  static DynLdsOnce once;
  if (ensure_dyn_lds(once, reinterpret_cast<const void*>(kernel), (int)lds) != kOk) return kErrLaunch;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace

int score_launch_small(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream) {
  if (!score_small_shape(s)) { set_error("no small-unit kernel for this shape"); return kErrUnsupported; }
  const int px = s.item_dot_product_groups, dd = s.dot_product_dimension;
  if (px == 4 && dd == 64) return launch_small<4, 64, 128>(a, n_cu, stream);
  if (px == 4 && dd == 128) return launch_small<4, 128, 128>(a, n_cu, stream);
  if (px == 8 && dd == 32) return launch_small<8, 32, 128>(a, n_cu, stream);
  return kErrUnsupported;
}

}  // namespace mol
