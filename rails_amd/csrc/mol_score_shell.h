// Kernel shells of the fused MoL scoring pass, shared by the exact-fp32 build (mol_score.hip) and the f16x3 build
// (mol_score_f16.hip).  A shell decides WHICH (query group, item tile) unit a wave works on and where the tile's operands
// come from (HBM/L2 directly, or LDS filled by LDS-DMA one tile ahead); the arithmetic of a unit is a policy class U:
//
//   U::gemm1<G, PX, DD>(D1, eq, tEx, lane)              sub-embedding contraction of the unit -> D1[PX] accumulators
//   U::queries<G, PX>(D1, p, g, only, item0, smem, tGi, lane, hi, x)
//                                                       gate MLP + softmax mixture of every query of the group, stores
//
// Both builds use the same buffer sizes (the f16 hi/lo fragments of the f16x3 mode take exactly the bytes of the fp32
// fragments they replace, mol_layout.h), so tile geometry, LDS budgets and DMA are common.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>

#include "mol_kernels.h"
#include "mol_layout.h"

namespace mol {

struct Fp32Unit;       // mol_score_fp32_unit.h: the exact-fp32 unit policy in tile order ...
struct Fp32UnitRows;   // ... and with the row-major copy's slot stride (mol_score_rows_kernel)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Gate pack -> LDS.  By LDS-DMA (1 KiB per wave-instruction, no registers, NOT waited for here): every shell passes a __syncthreads()
// -- which drains the wave's DMA pieces (vmcnt(0)) -- before its first weight read, and the staged shells issue their first tile's DMA
// right behind this, so the two round trips run side by side instead of one after the other (a copy through registers waits for its
// loads before it can store; small corpora are one round of tiles: every microsecond of prologue is on the critical path).
#ifndef RAILS_STAGE_DMA
#define RAILS_STAGE_DMA 1
#endif
template <class G, int NW>
__device__ __forceinline__ void stage_weights(const ScoreArgs& p, float* smem) {
#if RAILS_STAGE_DMA
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int kPieces = G::kWpackFloats / 256;
  for (int piece = wave; piece < kPieces; piece += NW)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.wpack + piece * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(smem + piece * 256), 16, 0, 0);
  for (int i = kPieces * 256 + threadIdx.x; i < G::kWpackFloats; i += NW * 64) smem[i] = p.wpack[i];
#else
  const float4* src = reinterpret_cast<const float4*>(p.wpack);
  float4* dst = reinterpret_cast<float4*>(smem);
  for (int i = threadIdx.x; i < G::kWpackFloats / 4; i += NW * 64) dst[i] = src[i];
#endif
}

// ---------------------------------------------------------------------------------------------
// Kernel A ("direct"): every wave is independent and reads its tile straight from HBM/L2.
// Used when fewer than 8 query groups exist (B < 8 * 32/P_Q), for per-row candidates, and for shapes whose
// tile does not fit LDS twice.  unit = (tile, query group), groups fastest.
// ---------------------------------------------------------------------------------------------
// INDEXED (per-row candidates only): the candidates are not gathered into tiles of their own first; lane x of a unit reads the
// fragments of ITS candidate straight from the shared index -- item i's share of fragment slot s sits at float4
// (i / 32) * tile + s * 64 + (lane & 32) + i % 32, i.e. at a per-lane base plus the same slot offsets a tile has.  Handing the unit
// policies `base - lane` as the tile pointer makes their `tile[slot * 64 + lane]` reads land there with no change to them.  A separate
// instantiation: the dense direct path keeps its wave-uniform tile pointer.
template <class U, int PQ, int PX, int DD, int H, int NW, bool INDEXED = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void mol_score_direct_kernel(ScoreArgs p) {
  using G = Geo<PQ, PX, DD, H>;
  MOL_RUN_IF(p.run_if);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  U::template stage<G, NW>(p, smem);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, x = lane & 31;
  const int inner = p.per_row ? (int)p.n_tiles : p.n_groups;
  const int64_t n_units = p.per_row ? (int64_t)p.B * p.n_tiles : p.n_tiles * p.n_groups;
  // Full rounds: the NW waves of a workgroup take NW consecutive units (the query groups of one item tile, so the tile's
  // fragments are shared through L1).  The leftover round is dealt wave-major instead -- unit r goes to workgroup
  // r % grid, wave r / grid -- so that it lands one unit per SIMD across the whole chip rather than two per SIMD on the
  // first few CUs (ML-20M: 6824 units on 1024 SIMDs, worst SIMD 7 passes instead of 8; ML-1M: all 256 CUs busy).
  const int64_t stride = (int64_t)gridDim.x * NW;
  const int64_t rounds = n_units / stride;
  // XCD-aware numbering: hardware workgroup b runs on XCD b % 8 (each XCD has its own L2).  Consecutive workgroups share
  // a tile's query groups; logical id = (b % 8) * (grid / 8) + b / 8 keeps consecutive logical workgroups on one XCD, so
  // a tile is fetched into one L2 instead of several.  Placement is only a speed matter: any mapping is correct.
  const int64_t bx = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  for (int64_t it = 0; it <= rounds; ++it) {
    const int64_t u = it < rounds ? it * stride + bx * NW + wave
                                  : rounds * stride + (int64_t)wave * gridDim.x + bx;
    if (u >= n_units) break;
    const int64_t outer = u / inner;
    const int innr = (int)(u - outer * inner);
    const int64_t tile = p.per_row ? innr : outer;  // tile index inside the row / corpus
    const int row = p.per_row ? (int)outer : -1;    // per-row mode: the only query of this unit
    const int g = p.per_row ? row / G::QT : innr;
    const int64_t tile_addr = p.per_row ? (int64_t)row * p.n_tiles + tile : tile;
    const float4* tEx = reinterpret_cast<const float4*>(p.ipack + tile_addr * (int64_t)G::kTileFloats);
    if constexpr (INDEXED) {
      int col = (int)tile * kTileItems + x;                        // ragged last tile: the row's last candidate again, those columns are not stored
      col = col < (int)p.n_items ? col : (int)p.n_items - 1;
      int64_t src = p.cand_pos[(int64_t)row * p.n_items + col];
      src = src < 0 ? 0 : (src >= p.index_items ? p.index_items - 1 : src);   // callers pass positions of the index; clamped for memory safety only
      tEx = reinterpret_cast<const float4*>(p.ipack) + (src >> 5) * (int64_t)(G::kTileFloats / 4) + (lane & 32) + (src & 31) - lane;
    }
    const float4* tGi = tEx + G::kTileExFloats / 4;
    const float* eq = p.eqfrag + (int64_t)g * G::kEqGroupFloats;
    f32x16 D1[PX];
#ifndef RAILS_DIRECT_PIPE
#define RAILS_DIRECT_PIPE 1
#endif
    U::template gemm1<G, PX, DD, (NW == 4), ((NW == 8 && RAILS_DIRECT_PIPE != 0) ? 1 : 0)>(D1, eq, tEx, lane);   // one wave per SIMD: the whole tile requested up front; two: one K-chunk ahead
    // (Measured three times and not kept: an L2 touch of this wave's next tile from here -- untracked asm loads in rounds 1 and 3,
    // ordinary loads consumed at the end of the unit in round 3: B = 1 ... 8 all 2-5 % slower.  What does help these shells is the
    // K-chunk lookahead inside GEMM1, above.  An early touch of THIS tile's gate rows, read at the head of the epilogue, changed nothing
    // either: the epilogue's own request-ahead ring already covers them.)
    U::template queries<G, PX>(D1, p, g, row, tile * kTileItems, smem, tGi, lane, hi, x);
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel A' ("rows"): the INDEXED form of kernel A reading each candidate from the ROW-MAJOR copy of the index
// (rails_mol_index_rows_build): lane (h, x) of a unit reads the float4 of slot s of ITS candidate at rows[i * RP + 2 s + h] -- the
// candidate's RP float4 are consecutive, so its 128-byte lines are fetched from HBM once and used whole (in the tile-packed index every
// one of the RP pieces sits in a line of its own: 8 x the bytes).  UR = the unit policy with slot stride 2 (Fp32UnitRows): same values,
// same order, same bits as kernel A.  Per-row candidates only.
// ---------------------------------------------------------------------------------------------
template <class UR, int PQ, int PX, int DD, int H, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void mol_score_rows_kernel(ScoreArgs p) {
  using G = Geo<PQ, PX, DD, H>;
  MOL_RUN_IF(p.run_if);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  UR::template stage<G, NW>(p, smem);
  __syncthreads();
  constexpr int RP = G::kTileFloats / 4 / kTileItems;      // float4 per item
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, x = lane & 31;
  const int64_t n_units = (int64_t)p.B * p.n_tiles;
  const int64_t stride = (int64_t)gridDim.x * NW;
  const int64_t rounds = n_units / stride;
  const int64_t bx = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  for (int64_t it = 0; it <= rounds; ++it) {
    const int64_t u = it < rounds ? it * stride + bx * NW + wave : rounds * stride + (int64_t)wave * gridDim.x + bx;
    if (u >= n_units) break;
    const int64_t row64 = u / p.n_tiles;
    const int tile = (int)(u - row64 * p.n_tiles), row = (int)row64;
    int cnt = (int)p.n_items;
    if (p.cand_count) {                                                // per-row candidate counts: the tiles past a row's candidates do nothing
      const int c = __builtin_amdgcn_readfirstlane(p.cand_count[row]);
      cnt = c < cnt ? c : cnt;
      if (tile * kTileItems >= cnt) continue;
    }
    int col = tile * kTileItems + x;                                   // ragged last tile: the row's last candidate again, those columns are not stored
    col = col < cnt ? col : cnt - 1;
    int64_t src = p.cand_pos[(int64_t)row * p.n_items + col];
    src = src < 0 ? 0 : (src >= p.index_items ? p.index_items - 1 : src);   // callers pass positions of the index; clamped for memory safety only
    const float4* tEx = reinterpret_cast<const float4*>(p.irows) + src * RP + hi - lane;
    const float4* tGi = tEx + 2 * (G::kTileExFloats / 256);            // the Ex slots, two float4 each
    const float* eq = p.eqfrag + (int64_t)(row / G::QT) * G::kEqGroupFloats;
    f32x16 D1[PX];
    UR::template gemm1<G, PX, DD, false, 1>(D1, eq, tEx, lane);
    UR::template queries<G, PX>(D1, p, row / G::QT, row, (int64_t)tile * kTileItems, smem, tGi, lane, hi, x);
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel B ("staged"): the workgroup's waves share one item tile per step.  The tile is copied
// HBM -> LDS by LDS-DMA (global_load_lds, 1 KiB per wave-instruction, no registers) one tile ahead of
// its use, double buffered, so each tile is read from HBM exactly once per batch and its latency is
// hidden behind a whole tile of MFMA work.  One barrier per tile.
// ---------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void dma_floats(const float* __restrict__ src, float* lds, int n_floats, int wave, int lane) {
  for (int piece = wave; piece < n_floats / 256; piece += NW) {   // 1 KiB pieces
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(src + piece * 256 + lane * 4),
        (__attribute__((address_space(3))) void*)(lds + piece * 256), 16, 0, 0);
  }
}

template <class U, int PQ, int PX, int DD, int H, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void mol_score_staged_kernel(ScoreArgs p) {
  using G = Geo<PQ, PX, DD, H>;
  MOL_RUN_IF(p.run_if);
  static_assert(G::kTileFloats % 256 == 0, "tile must be a whole number of 1 KiB pieces");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tiles = smem + U::template kLdsWeightFloats<G>;  // two tile buffers
  U::template stage<G, NW>(p, smem);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, x = lane & 31;
  const int64_t first = blockIdx.x;
  if (first < p.n_tiles) dma_floats<NW>(p.ipack + first * (int64_t)G::kTileFloats, tiles, G::kTileFloats, wave, lane);
  int cur = 0;
  int64_t it = 0;
  for (int64_t tile = first; tile < p.n_tiles; tile += gridDim.x, cur ^= 1, ++it) {
    // (1) my pieces of `tile` have landed (vmcnt(0)); (2) barrier: every wave's pieces have, and every wave
    // is done with the previous tile, so the other buffer may be overwritten
    __syncthreads();
    const int64_t next = tile + gridDim.x;
    if (next < p.n_tiles) dma_floats<NW>(p.ipack + next * (int64_t)G::kTileFloats, tiles + (cur ^ 1) * G::kTileFloats, G::kTileFloats, wave, lane);
    const float4* tEx = reinterpret_cast<const float4*>(tiles + cur * G::kTileFloats);
    const float4* tGi = tEx + G::kTileExFloats / 4;
    for (int g = wave; g < p.n_groups; g += NW) {
      const float* eq = p.eqfrag + (int64_t)g * G::kEqGroupFloats;
      f32x16 D1[PX];
#ifndef RAILS_STAGED_PIPE
#define RAILS_STAGED_PIPE 0   // GEMM1 one K-chunk ahead with the tile in LDS: measured no difference (6.34 ms either way)
#endif
      U::template gemm1<G, PX, DD, false, RAILS_STAGED_PIPE>(D1, eq, tEx, lane);
      U::template queries<G, PX>(D1, p, g, -1, tile * kTileItems, smem, tGi, lane, hi, x);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel B1 ("staged, single Ex buffer"): for shapes whose tile does not fit LDS twice next to the gate weights
// (8x4x128: 68 KiB tiles).  The sub-embedding part of a tile (Ex, 94 % of it) is only read by GEMM1, the first ~15 % of a
// unit; the gate part (gi) is read at the end.  So ONE Ex buffer is enough: after a second barrier ("every wave is past
// its last GEMM1") the next tile's Ex is DMA'd into the same buffer while the waves run the long gate MLP; gi is double
// buffered (4 KiB each).  LDS: weights + Ex + 2 gi = 107 KiB for 8x4x128.
//
// Leftover round: when the tiles left after the full rounds are at most half the grid, each is shared by
// nsub = grid / leftover workgroups that split its query groups (group = sub + nsub * wave), so the round runs one unit
// per SIMD instead of two on a few CUs (ML-20M: 853 tiles on 256 CUs -> 3 full rounds + 85 leftover tiles x 3 workgroups).
// ---------------------------------------------------------------------------------------------
template <class U, int PQ, int PX, int DD, int H, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void mol_score_staged1_kernel(ScoreArgs p) {
  using G = Geo<PQ, PX, DD, H>;
  MOL_RUN_IF(p.run_if);
  static_assert(G::kTileExFloats % 256 == 0 && G::kTileGiFloats % 256 == 0, "1 KiB DMA pieces");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sEx = smem + U::template kLdsWeightFloats<G>;
  float* sGi = sEx + G::kTileExFloats;   // two gi buffers

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, x = lane & 31;
  const int64_t grid = gridDim.x, b = blockIdx.x;
  const int64_t rounds = p.n_tiles / grid;
  const int64_t left = p.n_tiles - rounds * grid;
  int nsub = 1;
  if (left > 0 && left * 2 <= grid) {
    nsub = (int)(grid / left);
    if (nsub > NW) nsub = NW;
    if (nsub > p.n_groups) nsub = p.n_groups;
  }
  const bool has_left = b < left * nsub;
  const int64_t mine = rounds + (has_left ? 1 : 0);
  if (mine == 0) return;
  U::template stage<G, NW>(p, smem);
  auto tile_of = [&](int64_t i) -> int64_t { return i < rounds ? b + i * grid : rounds * grid + b % left; };

  {
    const float* t0 = p.ipack + tile_of(0) * (int64_t)G::kTileFloats;
    dma_floats<NW>(t0, sEx, G::kTileExFloats, wave, lane);
    dma_floats<NW>(t0 + G::kTileExFloats, sGi, G::kTileGiFloats, wave, lane);
  }
  int cur = 0;
  for (int64_t i = 0; i < mine; ++i, cur ^= 1) {
    const int64_t tile = tile_of(i);
    const bool split_round = i >= rounds && nsub > 1;
    const int off = split_round ? (int)(b / left) : 0, stride = split_round ? nsub : 1;
    const int cnt = (p.n_groups - off + stride - 1) / stride;   // query groups of this tile handled here (>= 1)
    const int n_it = (cnt + NW - 1) / NW;
    // (1) my DMA pieces of `tile` have landed (vmcnt(0)); (2) barrier: every wave's pieces have, and every wave is
    // done with the previous tile's gi buffer
    __syncthreads();
    const float4* tEx = reinterpret_cast<const float4*>(sEx);
    const float4* tGi = reinterpret_cast<const float4*>(sGi + cur * G::kTileGiFloats);
    for (int it = 0; it < n_it; ++it) {
      const int gi_idx = wave + it * NW;
      const bool has = gi_idx < cnt;
      const int g = off + stride * gi_idx;
      f32x16 D1[PX];
#ifndef RAILS_STAGED1_PIPE
#define RAILS_STAGED1_PIPE 3   // a ring of three K-chunks: ML-20M 64 x 221 184 2.820 -> 2.765 ms, B = 32 0.200 -> 0.194 ms (one ahead: 2.780); ML-1M 0.030 ms either way
#endif
      // the ring costs PIPE * (PX + 1) float4 registers: built where that is <= 16 and K is deeper than the ring (4 x 128 shapes); elsewhere none (8 x 32 would spill)
      constexpr int kPipe = (RAILS_STAGED1_PIPE > 1 && (RAILS_STAGED1_PIPE * (PX + 1) > 16 || DD / 8 <= RAILS_STAGED1_PIPE)) ? 0 : RAILS_STAGED1_PIPE;
      if (has) U::template gemm1<G, PX, DD, false, kPipe>(D1, p.eqfrag + (int64_t)g * G::kEqGroupFloats, tEx, lane);
      if (it == n_it - 1) {
        __syncthreads();   // every wave is past its last GEMM1 of this tile: the Ex buffer is free
        if (i + 1 < mine) {
          const float* tn = p.ipack + tile_of(i + 1) * (int64_t)G::kTileFloats;
          dma_floats<NW>(tn, sEx, G::kTileExFloats, wave, lane);
          dma_floats<NW>(tn + G::kTileExFloats, sGi + (cur ^ 1) * G::kTileGiFloats, G::kTileGiFloats, wave, lane);
        }
      }
      if (has) U::template queries<G, PX>(D1, p, g, -1, tile * kTileItems, smem, tGi, lane, hi, x);
    }
  }
}

// ---- launch helpers ----------------------------------------------------------------------------------------------
// RAILS_SCORE_VARIANT: 0 = pick automatically; 1 / 2 = force direct / staged with 8 waves (2 per SIMD);
// 3 / 4 = direct / staged with 4 waves (1 per SIMD, 512 registers); 5 = staged with a single Ex buffer (8 waves);
// 6 = staged with a single Ex buffer, 4 waves
inline int score_variant() {
  const char* e = getenv("RAILS_SCORE_VARIANT");
  return e ? atoi(e) : 0;
}

template <class U, int PQ, int PX, int DD, int H, int NW, bool STAGED>
static int launch_kernel(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using G = Geo<PQ, PX, DD, H>;
  constexpr size_t lds = ((size_t)U::template kLdsWeightFloats<G> + (STAGED ? 2 * (size_t)G::kTileFloats : 0)) * sizeof(float);
  if constexpr (lds > 160 * 1024) {
    set_error("staged scoring kernel needs %zu B of LDS", lds);
    return kErrUnsupported;
  } else {
    if (a.dry_run) return kOk;
    const int wg_per_cu = (NW == 4 && lds <= 80 * 1024) ? 2 : 1;
    int64_t grid;
    if (STAGED) {
      grid = a.n_tiles;
    } else {
      const int64_t n_units = a.per_row ? (int64_t)a.B * a.n_tiles : a.n_tiles * a.n_groups;
      grid = n_units;   // fewer units than wave slots: one unit per workgroup first (wave-major remainder mapping)
    }
    if (grid > (int64_t)n_cu * wg_per_cu) grid = (int64_t)n_cu * wg_per_cu;
    if (grid < 1) return kOk;
    auto go = [&](auto kernel) {
      static DynLdsOnce once;   // one per kernel (the lambda is instantiated per kernel type)
      if (ensure_dyn_lds(once, reinterpret_cast<const void*>(kernel), (int)lds) != kOk) return (int)kErrLaunch;
      hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
      return hipGetLastError() == hipSuccess ? (int)kOk : (int)kErrLaunch;
    };
    if constexpr (STAGED) {
      if (a.cand_pos) { set_error("indexed candidates need the independent-wave shell"); return kErrUnsupported; }
      return go(&mol_score_staged_kernel<U, PQ, PX, DD, H, NW>);
    } else {
      if constexpr (U::kIndexedCandidates && NW == 8) {
        if constexpr (std::is_same_v<U, Fp32Unit>) {
          if (a.cand_pos && a.irows) return go(&mol_score_rows_kernel<Fp32UnitRows, PQ, PX, DD, H, NW>);
        }
        if (a.cand_pos) return go(&mol_score_direct_kernel<U, PQ, PX, DD, H, NW, true>);
      }
      if (a.cand_pos) { set_error("indexed candidates are built for the exact-fp32 kernels only"); return kErrUnsupported; }
      return go(&mol_score_direct_kernel<U, PQ, PX, DD, H, NW>);
    }
  }
}

template <class U, int PQ, int PX, int DD, int H, int NW>
static int launch_staged1(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using G = Geo<PQ, PX, DD, H>;
  constexpr size_t lds = ((size_t)U::template kLdsWeightFloats<G> + (size_t)G::kTileExFloats + 2 * (size_t)G::kTileGiFloats) * sizeof(float);
  if constexpr (lds > 160 * 1024) {
    set_error("single-buffer staged scoring kernel needs %zu B of LDS", lds);
    return kErrUnsupported;
  } else {
    if (a.dry_run) return kOk;
    if (a.n_tiles < 1) return kOk;
    auto go = [&](auto kernel) {
      static DynLdsOnce once;
      if (ensure_dyn_lds(once, reinterpret_cast<const void*>(kernel), (int)lds) != kOk) return (int)kErrLaunch;
      // always a full grid: the leftover-round split needs the idle workgroups (they exit at once otherwise)
      hipLaunchKernelGGL(kernel, dim3((unsigned)n_cu), dim3(NW * 64), lds, stream, a);
      return hipGetLastError() == hipSuccess ? (int)kOk : (int)kErrLaunch;
    };
    return go(&mol_score_staged1_kernel<U, PQ, PX, DD, H, NW>);
  }
}

// Which shell for this launch (0 = automatic): few query groups / per-row candidates -> independent waves (1);
// double-buffered tiles where they fit and the corpus fills the chip for many rounds (2, the measured headline path);
// otherwise the single-Ex-buffer kernel (5): 8x4x128 tiles only fit once, and its leftover-round split is what keeps small
// corpora (ML-1M: 122 tiles, ML-20M: 853) spread over all CUs.
template <int PQ, int PX, int DD, int H>
inline int choose_variant(const ScoreArgs& a, int n_cu) {
  using G = Geo<PQ, PX, DD, H>;
  constexpr bool staged_fits = ((size_t)G::kWpackFloats + 2 * (size_t)G::kTileFloats) * sizeof(float) <= 160 * 1024;
  constexpr bool staged1_fits = ((size_t)G::kWpackFloats + (size_t)G::kTileExFloats + 2 * (size_t)G::kTileGiFloats) * sizeof(float) <= 160 * 1024;
  int variant = score_variant();
  if (variant == 0) {
    variant = 1;
    if (!a.per_row && a.n_groups >= kScoreWaves) {
      if (staged_fits && a.n_tiles >= 8 * (int64_t)n_cu) variant = 2;
      else if (staged1_fits) variant = 5;
      else if (staged_fits) variant = 2;
    }
  }
  return variant;
}

}  // namespace mol
