// Coarse pass of the two-pass approximate top-k (MoLAvgTopK, reference rails/indexing/mol_top_k.py:296-429).
//
// The reference keeps the component embeddings in bf16 (mol_top_k.py:37,73), averages them over the P_X
// groups into a (d, N) bf16 table (mol_top_k.py:321-325) and scores sum_p Eq[b,p,:] against it with a bf16
// `mm` (mol_top_k.py:351-354).  These kernels restate that arithmetic: every intermediate the reference rounds
// to bf16 is rounded to bf16 (round-to-nearest-even) here, accumulation is fp32.
//   table[x, :] = bf16( bf16( sum_m bf16(Ex[x,m,:]) ) / P_X )                     2*d bytes per item
//   qsum[b, :]  = bf16( sum_p Eq[b,p,:] )      (forward)   or   bf16( sum_p Eq / P_Q )   (topk_ids)
//   score[b, x] = bf16( sum_d qsum[b,d] * table[x,d] )                            returned as fp32
// The scan is HBM-bound: 2*d bytes per item against 2*d*B flops.
#include <hip/hip_runtime.h>
#include <math.h>

#include "mol_kernels.h"
#include <utility>
#include "mol_layout.h"

namespace mol {

__device__ __forceinline__ float bf16_rn(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return x;  // NaN stays NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xFFFF0000u);
}
__device__ __forceinline__ unsigned short bf16_bits(float x) { return (unsigned short)(__float_as_uint(bf16_rn(x)) >> 16); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }

// one thread per (item, d): reads the tile-packed index, writes table[item][dd] row-major bf16
__global__ void coarse_build_kernel(const float* __restrict__ ipack, int64_t n, int PQ, int PX, int d,
                                    unsigned short* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * d) return;
  const int64_t item = i / d;
  const int dd = (int)(i - item * d);
  const int64_t tile = item >> 5;
  const int x = (int)(item & 31);
  const float* tEx = ipack + tile * (int64_t)(kTileItems * (PX * d + PQ * PX));
  // Ex slot (m, c8, lane)[j] holds Ex[x][m][hi*d/2 + 4*c8 + j], lane = hi*32 + x
  const int hi = dd / (d / 2), s = dd - hi * (d / 2);
  float acc = 0.0f;
  for (int m = 0; m < PX; ++m) acc += bf16_rn(tEx[((m * (d / 8) + (s >> 2)) * 64 + hi * 32 + x) * 4 + (s & 3)]);
  table[i] = bf16_bits(bf16_rn(acc) / (float)PX);
}

// ---- the coarse scan (bf16 MFMA) ------------------------------------------------------------------------------
// 2*d*B flops against 2*d bytes per item: at B = 32 that is 2048 flop per 64-byte item -- 6x what the VALU can stream at
// HBM rate, nothing for the bf16 MFMA.  A wave takes tiles of 32 items: v_mfma_f32_32x32x16_bf16 with the 32 queries of a
// query tile on the row axis (A: the P_Q-summed query rows, bf16, fragment order in LDS -> registers) and the items on the
// column axis (B: lane (x, h) reads 16 contiguous bytes of table row x per 16-wide K chunk, so a wave reads the tile's
// 32*2d bytes exactly once, fully coalesced).  fp32 accumulate, result rounded to bf16 like the reference's bf16 mm.
//
// Three modes share the arithmetic (so their scores are bit-identical):
//   kScanAll     scores[b][x] for every item                         (the materialising path)
//   kScanSample  every `stride`-th tile; each wave keeps the running MAXIMUM of its scores per (query, tile column) and writes
//                one 32 x 32 block of maxima at the end               (threshold estimation of the fused top-K', see below)
//   kScanSelect  keys (score, position) of the items whose score is >= thr[b] appended to per-query candidate lists
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float cf32x16 __attribute__((ext_vector_type(16)));
enum { kScanAll = 0, kScanSample = 1, kScanSelect = 2 };
// A query's candidate list is split into sub-lists by tile index (t % kSubLists): one device-scope counter per query
// serialised ~600k appends at K' = 4000 (2.4 ms for a 0.75 ms scan); 16 counters per query spread them, and taking the
// sub-list from the tile index keeps the split even for any item order.
constexpr int kSubLists = 16;
constexpr int kScanThreads = 256;
constexpr int kSampleMaxQT = 4;       // query tiles (of 32) a sample launch keeps running maxima for: B <= 128
constexpr int kSelectMaxRows = 512;   // query rows of a select launch (its workgroup flush keeps a counter per row in LDS)
constexpr int kSampleGrid = 256;      // workgroups of a sample launch: 1 024 waves -> 32 768 maxima per query, one row_select launch
#ifndef RAILS_SCAN_NT
#define RAILS_SCAN_NT 1   // the select scan of a single query tile (B <= 32) reads the table with non-temporal loads -- each byte is used once:
                          // 125 M x 32 bf16, B = 32: 1 388 -> 1 315 us (5.8 -> 6.1 TB/s); with four query tiles (B = 128) the scan is no longer
                          // bound by the reads and the hint costs 1.5 %, so those launches keep the default policy
#endif
#ifndef RAILS_SCAN_WAVES
#define RAILS_SCAN_WAVES 3   // waves per SIMD the select scan is compiled for (168 VGPRs; 3 over 2: 3.07 -> 2.87 ms per config-5 step at B = 128,
                           // nothing at B = 32); the store modes hold sixteen addresses per tile and keep 2
#endif
#ifndef RAILS_SCAN_TU
#define RAILS_SCAN_TU 0   // item tiles per trip of the scan (0: by d)
#endif

struct CoarseScanArgs {
  const float* eq; int B, PQ, d, avg;
  // GROUPS (round 6: the per-component scans of MoLNaiveTopK / MoLCombTopK run on this kernel): groups > 1 = one table per item group m
  // (blockIdx.y), group_stride elements apart, the SAME query rows against each; comp = 1: the query rows are the B * P_Q sub-embeddings
  // bf16(Eq[b, i, :]) themselves (no sum over the groups).  Row (q, m) of every per-row quantity -- thresholds, sample maxima, scores,
  // candidate lists -- is q * groups + m.  groups = 1, comp = 0: the coarse pass of MoLAvgTopK.
  int groups, comp; int64_t group_stride;
  int stage_cap;                        // kScanSelect: entries of the workgroup's LDS list of hits (set by launch_coarse_scan)
  int no_hits;                          // RAILS_COMP_DEBUG=1 (measurement): the select scan with every threshold at +inf -- its cost without candidates
  const unsigned short* qfrag;          // the queries' A fragments, made once by workgroup 0 of the sample scan (NULL: every workgroup makes them from eq)
  const unsigned short* table; int64_t n;
  float* scores; int64_t ld;            // kScanAll: scores[b * ld + item]
  unsigned short* scores16;             // kScanSample: scores16[b * ld + wave * 32 + x] = the wave's running maximum in column x, as bf16 bits
  int stride;                           // kScanSample: the tiles t with t % stride == 0
  unsigned short* qfrag_out;            // kScanSample: workgroup 0 leaves the queries' A fragments here for the select scan ...
  unsigned int* zero_words; int n_zero; // ... and zeroes these words (the candidate counters)
  int32_t* zero_flag;                   // ... and the caller's out-of-range flag, which the key selection may raise
  signed char* q8_out; float* qmeta_out; // ... and, for the int8 pre-filter, the queries' int8 fragments and (s_q, |q|_1) per query
  const float* thr; int64_t thr_stride; // kScanSelect: thr[b * thr_stride], a bf16 value
  unsigned long long* keys; int cap;    // kScanSelect: keys[b * cap + sub * (cap / kSubLists) + slot]
  unsigned int* counts;                 // kScanSelect: counts[b * kSubLists + sub], candidates seen (may exceed the sub-list)
  const int32_t* run_if;                // launch predicate (mol_kernels.h); set for the materialising scan only
};

__device__ __forceinline__ unsigned int coarse_orderable(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float coarse_unorderable(unsigned int k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Append path of the select scans.  A hit's slot comes from a device-scope atomic whose result takes ~2 us to return;
// issued one by one inside the scan they serialise (a wave of the component scan met ~40 per tile: 2 ms for a 0.1 ms
// scan).  Hits are therefore staged in a per-wave LDS list (LDS atomic cursor) and flushed once per tile, one entry per
// lane, so up to 64 global atomics are in flight per wave.  DS operations of one wave execute in issue order, so the
// wave reads back its own staged entries without a barrier; `volatile` keeps the compiler from reordering them.
struct StageEntry { unsigned long long key; unsigned int orow; unsigned int pad; };
constexpr int kStage = 128;   // entries per wave (2 KiB)

__device__ __forceinline__ void append_candidate(unsigned long long* keys, unsigned int* counts, int cap, int sub,
                                                 unsigned int orow, unsigned long long key) {
  const int subcap = cap / kSubLists;
  const unsigned int slot = counts ? atomicAdd(&counts[(int64_t)orow * kSubLists + sub], 1u) : 0u;   // (counts == NULL: RAILS_COMP_DEBUG=2, a measurement of the path without its atomics)
  if (slot < (unsigned int)subcap) keys[(int64_t)orow * cap + sub * subcap + slot] = key;
}
__device__ __forceinline__ void stage_push(volatile StageEntry* st, unsigned int* cnt, unsigned long long* keys,
                                           unsigned int* counts, int cap, int sub, unsigned int orow, unsigned long long key) {
  const unsigned int i = atomicAdd(cnt, 1u);   // LDS
  if (i < (unsigned int)kStage) { st[i].key = key; st[i].orow = orow; st[i].pad = (unsigned int)sub; }
  else append_candidate(keys, counts, cap, sub, orow, key);   // list full: straight to global
}
__device__ __forceinline__ void stage_flush(volatile StageEntry* st, volatile unsigned int* cnt, int lane, unsigned long long* keys,
                                            unsigned int* counts, int cap, int sub) {
  unsigned int n = *cnt;
  if (n == 0) return;        // wave-uniform
  if (n > (unsigned int)kStage) n = kStage;
  for (unsigned int e = lane; e < n; e += 64) append_candidate(keys, counts, cap, sub, st[e].orow, st[e].key);
  if (lane == 0) *cnt = 0u;
}
// The same with each entry's own sub-list (stage_push records it): a wave that meets a hit every few tiles keeps staging
// across tiles and flushes once a wave's worth of entries has gathered (and at the end of its scan) -- flushed per tile, the
// one or two entries of a hit cost the wave the ~2 us of a device-scope atomic each time.
__device__ __forceinline__ void stage_flush_mixed(volatile StageEntry* st, volatile unsigned int* cnt, int lane, unsigned long long* keys,
                                                  unsigned int* counts, int cap, unsigned int at_least) {
  unsigned int n = *cnt;
  if (n < at_least || n == 0) return;        // wave-uniform
  if (n > (unsigned int)kStage) n = kStage;
  for (unsigned int e = lane; e < n; e += 64) append_candidate(keys, counts, cap, (int)st[e].pad, st[e].orow, st[e].key);
  if (lane == 0) *cnt = 0u;
}

// The same staging with the wave's fill count in a scalar register (component scans, round 6: a hit every few blocks): no LDS atomic and no
// LDS read on the way -- slots from a ballot's prefix count, the entries written and forgotten; the flush reads them back (DS operations of a
// wave execute in issue order).  `sel` lanes append (key, orow, sub).
__device__ __forceinline__ void stage_push_reg(volatile StageEntry* st, unsigned int& staged, bool sel, unsigned long long key, unsigned int orow, int sub,
                                               unsigned long long* keys, unsigned int* counts, int cap, int lane) {
  const unsigned long long m = __ballot(sel);
  if (m == 0ull) return;
  const unsigned int slot = staged + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
  if (sel) {
    if (slot < (unsigned int)kStage) { st[slot].key = key; st[slot].orow = orow; st[slot].pad = (unsigned int)sub; }
    else append_candidate(keys, counts, cap, sub, orow, key);      // list full: straight to global
  }
  staged += (unsigned int)__popcll(m);
}
__device__ __forceinline__ void stage_flush_reg(volatile StageEntry* st, unsigned int& staged, int lane, unsigned long long* keys, unsigned int* counts, int cap,
                                                unsigned int at_least) {
  if (staged < at_least || staged == 0u) return;      // wave-uniform
  const unsigned int n = staged < (unsigned int)kStage ? staged : (unsigned int)kStage;
  for (unsigned int e = lane; e < n; e += 64) append_candidate(keys, counts, cap, (int)st[e].pad, st[e].orow, st[e].key);
  staged = 0u;
}

// Pre-test of the select scan: the accumulator STARTS at minus the pre-test bound of its (query row, register), so "some score of
// this lane reaches its bound" is "some accumulator has a clear sign bit" -- an unsigned minimum over the sixteen registers (seven
// v_min3_u32 and a v_min_u32) and one compare per tile, instead of sixteen compares and sixteen mask ORs.  The bound is one bf16
// step below the threshold, far more than the rounding the shifted accumulation adds; a tile that passes is scored again from
// zero, so the scores that are kept are the materialising path's bits.  (A NaN passes the pre-test and fails the exact one.)
// What "far more" means: a score that the exact test keeps sits at least half a bf16 step (2^-9 |thr|) above the bound, and the two
// accumulation orders differ by at most ~d * 2^-24 * (|bound| + sum |q_k x_k|); the pre-test could only lose a kept score if the
// threshold were below ~5e-4 of the row's sum of |products| -- the top of the score distribution cancelling to zero -- AND that
// score were among the K' best, i.e. the K'-th best within that rounding noise of the threshold although ~4 K' candidates are
// expected above it.  The fused == materialised tests (ties included) and tools/fuzz_fused_scans.py have never seen a difference.
__device__ __forceinline__ unsigned int umin3(unsigned int a, unsigned int b, unsigned int c) { return min(min(a, b), c); }
__device__ __forceinline__ bool any_sign_clear(const cf32x16& v) {
  auto u = [&](int i) { return __float_as_uint(v[i]); };
  const unsigned int a = umin3(u(0), u(1), u(2)), b = umin3(u(3), u(4), u(5)), c = umin3(u(6), u(7), u(8));
  const unsigned int d = umin3(u(9), u(10), u(11)), e = umin3(u(12), u(13), u(14));
  return min(umin3(a, b, c), umin3(d, e, u(15))) < 0x80000000u;
}

// Element i = (query b, dimension dd) of the coarse query  bf16(sum or mean over the P_Q groups of Eq[b])  into its slot of the
// MFMA A fragments [query tile][K chunk][64 lanes][8] (lane = 32 * (k half) + row).
__device__ __forceinline__ void coarse_query_element(const float* __restrict__ eq, int B, int PQ, int d, int avg, int i, unsigned short* frag) {
  const int DC = d / 16;
  const int b = i / d, dd = i - b * d;
  float acc = 0.0f;
  if (b < B) {
    // eight loads in flight, then their sum in the same order (one load per add waits a round trip per query group: P_Q round
    // trips at the head of every workgroup of the sample scan -- 26.9 -> 19.7 us for that launch)
    for (int p0 = 0; p0 < PQ; p0 += 8) {
      float v8[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) v8[jj] = p0 + jj < PQ ? eq[((int64_t)b * PQ + p0 + jj) * d + dd] : 0.0f;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        if (p0 + jj < PQ) acc += v8[jj];
    }
  }
  const float v = b < B ? bf16_rn(avg ? acc / (float)PQ : acc) : 0.0f;
  const int qt = b >> 5, row = b & 31, c = dd >> 4, h = (dd >> 3) & 1, j = dd & 7;
  frag[(((size_t)qt * DC + c) * 64 + h * 32 + row) * 8 + j] = (unsigned short)(__float_as_uint(v) >> 16);
}
// component mode: element i = (query row = b * P_Q + group, dimension dd) of bf16(Eq) itself
__device__ __forceinline__ void component_query_element(const float* __restrict__ eq, int R, int d, int i, unsigned short* frag) {
  const int DC = d / 16;
  const int row = i / d, dd = i - row * d;
  const float v = row < R ? bf16_rn(eq[(int64_t)row * d + dd]) : 0.0f;
  const int qt = row >> 5, rr = row & 31, c = dd >> 4, h = (dd >> 3) & 1, j = dd & 7;
  frag[(((size_t)qt * DC + c) * 64 + h * 32 + rr) * 8 + j] = (unsigned short)(__float_as_uint(v) >> 16);
}
__device__ __forceinline__ void quantise_query(const unsigned short* qfrag, int DC, int d, int q, signed char* q8, float* qmeta);   // int8 pre-filter, below

template <int DC, int MODE, bool NT = false, int QTS = kSampleMaxQT, bool COMP = false>   // DC = d / 16 K chunks; NT: non-temporal table loads; QTS: query tiles a sample launch keeps maxima for; COMP: the component scans' select / sample blocks
__global__ __launch_bounds__(kScanThreads) __attribute__((amdgpu_waves_per_eu((MODE == kScanSelect && !COMP) ? RAILS_SCAN_WAVES : 2, (MODE == kScanSelect && !COMP) ? RAILS_SCAN_WAVES : 2))) void coarse_scan_kernel(CoarseScanArgs a) {
  MOL_RUN_IF(a.run_if);
  extern __shared__ __attribute__((aligned(16))) unsigned short qfrag[];   // [n_qt][DC][64 lanes][8] bf16, then thr
  const int d = a.d, B = a.comp ? a.B * a.PQ : a.B;      // B: query ROWS from here on
  const int gm = blockIdx.y, groups = a.groups;          // item group of this workgroup
  const unsigned short* const table = a.table + (int64_t)gm * a.group_stride;
  const int n_qt = (B + 31) / 32;
  float* thr_s = reinterpret_cast<float*>(qfrag + (size_t)n_qt * DC * 64 * 8);   // [n_qt * 32]
  __shared__ StageEntry stage_s[COMP ? kScanThreads / 64 : 1][COMP ? kStage : 1];      // (the per-wave lists of the COMP select block)
  __shared__ unsigned int stage_n[kScanThreads / 64];
  __shared__ float acc_s[MODE == kScanSelect ? (kScanThreads / 64) * 16 * 64 : 1];   // a fired tile's scores, per wave
  // The select scan's appends (round 6).  A hit used to take its slot from a device-scope atomic on counts[row][tile % 16]: with few query
  // rows (MoLAvgTopK: 32 rows = 512 counters) the ~500 k appends of K' = 4 000 queue up on the same addresses -- 257 us of a scan that takes 46
  // without the atomics and 10 without candidates (tools/r06_probe_t.sh).  Now a workgroup keeps ALL its hits in one LDS list (wg_stage, LDS
  // cursor; a.stage_cap entries behind the thresholds in dynamic LDS: 1 024 where that leaves three workgroups per CU, else 512) and flushes
  // once, at its end: the entries of a row are counted (LDS), ONE device-scope atomic per (row, workgroup) reserves their slots in sub-list
  // blockIdx.x % 16, and the keys go out.  Entries beyond the list take their slots one by one as before.
  const int kWgStage = a.stage_cap;
  unsigned int* const row_cnt_s = reinterpret_cast<unsigned int*>(thr_s + 2 * n_qt * 32);      // [n_qt * 32]
  unsigned int* const row_base_s = row_cnt_s + n_qt * 32;                                     // [n_qt * 32]
  StageEntry* const wg_stage = reinterpret_cast<StageEntry*>(row_base_s + n_qt * 32);        // [a.stage_cap], 16-byte aligned (every part is a multiple of 128 bytes)
  __shared__ unsigned int wg_n;
  if (threadIdx.x < kScanThreads / 64) stage_n[threadIdx.x] = 0u;
  if (threadIdx.x == 0) wg_n = 0u;
  if (a.qfrag) {   // 16 bytes per thread and step instead of P_Q dependent loads per element (2 048 workgroups each made them)
    for (int i = threadIdx.x; i < n_qt * DC * 64; i += kScanThreads)
      reinterpret_cast<bf16x8*>(qfrag)[i] = reinterpret_cast<const bf16x8*>(a.qfrag)[i];
  } else if (a.comp) {
    for (int i = threadIdx.x; i < n_qt * 32 * d; i += kScanThreads) component_query_element(a.eq, B, d, i, qfrag);
  } else {
    for (int i = threadIdx.x; i < n_qt * 32 * d; i += kScanThreads) coarse_query_element(a.eq, B, a.PQ, d, a.avg, i, qfrag);
  }
  float* ntlo_s = thr_s + n_qt * 32;                                              // [n_qt * 32]: minus the pre-test bound
  if constexpr (MODE == kScanSelect)
    for (int i = threadIdx.x; i < n_qt * 32; i += kScanThreads) {
      const float thr = (i < B && !a.no_hits) ? a.thr[((int64_t)i * groups + gm) * a.thr_stride] : INFINITY;
      thr_s[i] = thr;
      ntlo_s[i] = -coarse_unorderable(coarse_orderable(thr) - 0x10000u);
    }
  __syncthreads();
  if constexpr (MODE == kScanSample) {
    if (blockIdx.x == 0 && blockIdx.y == 0) {
      if (a.qfrag_out)
        for (int i = threadIdx.x; i < n_qt * DC * 64; i += kScanThreads) reinterpret_cast<bf16x8*>(a.qfrag_out)[i] = reinterpret_cast<const bf16x8*>(qfrag)[i];
      for (int i = threadIdx.x; i < a.n_zero; i += kScanThreads) a.zero_words[i] = 0u;
      if (a.zero_flag && threadIdx.x == 0) *a.zero_flag = 0;
      if (a.q8_out)
        for (int q = threadIdx.x; q < n_qt * 32; q += kScanThreads) quantise_query(qfrag, DC, d, q, a.q8_out, a.qmeta_out);
    }
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = lane & 31, h = lane >> 5;
  const int64_t n_tiles = (a.n + 31) >> 5;
  const int64_t step = MODE == kScanSample ? a.stride : 1;
  const int64_t n_work = (n_tiles + step - 1) / step;            // tiles this launch visits
  const int64_t gw = (int64_t)blockIdx.x * (kScanThreads / 64) + wave, n_waves = (int64_t)gridDim.x * (kScanThreads / 64);

  // One trip = TU item tiles of a wave, held as B fragments in registers.  The trips are DOUBLE-BUFFERED: the 16-byte loads of
  // the next trip are issued before the current one is scored, so a wave always has a trip of table bytes in flight (with one
  // trip per wave and two waves per SIMD only 4 MB of the chip's reads were outstanding: 4.6 TB/s by Little's law).
  constexpr int TU = (MODE == kScanSample && QTS > kSampleMaxQT) ? (DC <= 2 ? 2 : 1)      // eight row tiles of running maxima are 128 registers: shorter trips
                     : (MODE == kScanSelect && RAILS_SCAN_TU > 0) ? RAILS_SCAN_TU : (MODE == kScanAll ? (DC <= 4 ? 2 : 1) : (DC <= 2 ? 4 : (DC <= 4 ? 2 : 1)));   // the score stores of kScanAll hold 16 addresses per tile
  struct Trip {
    bf16x8 Bv[TU][DC];
    int64_t item[TU];
    bool in[TU];
  };
  auto load_trip = [&](int64_t w0, Trip& T) {
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      int64_t item = (w0 + u) * step * 32 + x;
      T.in[u] = (w0 + u) < n_work && item < a.n;
      if (!T.in[u]) item = a.n - 1;
      T.item[u] = item;
      const unsigned short* rowp = table + item * d + 8 * h;
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        if constexpr (NT) T.Bv[u][c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(rowp + 16 * c));
        else T.Bv[u][c] = *reinterpret_cast<const bf16x8*>(rowp + 16 * c);
      }
    }
  };
  auto load_query_tile = [&](int qt, bf16x8 (&A)[DC], cf32x16& ntlo) {
#pragma unroll
    for (int c = 0; c < DC; ++c) A[c] = *reinterpret_cast<const bf16x8*>(qfrag + (((size_t)qt * DC + c) * 64 + lane) * 8);
    if constexpr (MODE == kScanSelect) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ntlo[r] = ntlo_s[qt * 32 + acc_row(r, h)];
    }
  };
  // Select mode, the rare part (K'/N of the scores pass, a few percent of the tiles at shard scale): the tiles of the trip whose
  // pre-test fired are scored again FROM ZERO -- the bits of the materialising path -- and their scores at or above the query's
  // threshold are appended.  One copy of this code per kernel: the tile is picked out of the trip's registers by `u`.
  auto keep_candidates = [&](int qt, const bf16x8 (&A)[DC], const cf32x16& ntlo, const Trip& T, unsigned int fired, int64_t w0) {
#pragma unroll 1
    for (int u = 0; u < TU; ++u) {
      if (!((fired >> u) & 1u)) continue;
      bf16x8 Bu[DC];
      int64_t item = 0;
      bool in = false;
      // constant indices from the front end on, so that the trip stays in registers
      auto pick = [&]<int V, int... C>(std::integral_constant<int, V>, std::integer_sequence<int, C...>) {
        item = T.item[V];
        in = T.in[V];
        ((Bu[C] = T.Bv[V][C]), ...);
      };
      [&]<int... V>(std::integer_sequence<int, V...>) {
        ((u == V ? pick(std::integral_constant<int, V>{}, std::make_integer_sequence<int, DC>{}) : (void)0), ...);
      }(std::make_integer_sequence<int, TU>{});
      cf32x16 acc = {0};
#pragma unroll
      for (int c = 0; c < DC; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c], Bu[c], acc, 0, 0, 0);
      // Which of the lane's sixteen scores reach their bound (still in the accumulator-start registers, negated): a bit mask per
      // lane, one compare each.  Then a ROLLED loop over the set bits -- normally one bit in one lane -- with the scores read
      // back from LDS by register number.  (Unrolled over the registers, this rare block set the register allocation of the
      // whole kernel: 256 VGPRs and a spill in the middle of the prefetch; rolled over all sixteen with a ballot each, it cost
      // as many VALU instructions per fired tile as 25 tiles of the pre-test.)
      unsigned int mask = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) mask |= acc[r] >= -ntlo[r] ? 1u << r : 0u;   // -ntlo = the bf16 value just below the threshold
      if (!in) mask = 0u;
      if (__any(mask != 0u)) {
        float* mine = acc_s + (wave * 16) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r * 64] = acc[r];
        while (__any(mask != 0u)) {
          if (mask != 0u) {
            const int r = __ffs(mask) - 1;
            mask &= mask - 1u;
            const int q = qt * 32 + acc_row(r, h);
            const float thr = thr_s[q];
            const float sc = bf16_rn(mine[r * 64]);   // an un-rounded sum at or above the bound may round up to thr
            if (q < B && sc >= thr) {
              const unsigned long long key = ((unsigned long long)coarse_orderable(sc) << 32) | (unsigned int)(~(unsigned int)item);
              const unsigned int i = atomicAdd(&wg_n, 1u);   // LDS
              if (i < (unsigned int)kWgStage) { wg_stage[i].key = key; wg_stage[i].orow = (unsigned int)q; }
              else append_candidate(a.keys, a.counts, a.cap, (int)(blockIdx.x % kSubLists), (unsigned int)(q * groups + gm), key);   // list full: one by one
            }
          }
        }
      }
    }
    (void)w0;
  };
  // More than one tile of 32 queries (B > 32): the table is still read ONCE -- the query tiles walk over the trip in registers,
  // their A fragments and accumulator bounds come from LDS once per trip.  (Reading the table once per query tile cost 7.75 ms
  // per 125 M-item shard at B = 128.)
  bf16x8 A[DC];
  cf32x16 ntlo = {0};
  load_query_tile(0, A, ntlo);
  // Sample mode: mx[qt][r] = the largest score this wave has seen for (query qt * 32 + acc_row(r, h), column x).  The r-th largest
  // of these maxima over disjoint groups of items is at most the r-th largest sample score, so it bounds the K'-th score of the
  // corpus from below just as the sample's own r-th largest does (coarse_topk below) -- and the two coincide unless two of the
  // sample's top r fell into one group (r^2 / 2 groups expected: 0.02 for r = 40 and 32 768 groups).  It replaces B * n / stride
  // two-byte stores and their read-back by the selection (64 + 64 MB per batch of 32 on a 125 M-item shard: 40 + 48 us) with a
  // 2 MB block of maxima.
  cf32x16 mx[MODE == kScanSample ? QTS : 1];
  if constexpr (MODE == kScanSample) {
#pragma unroll
    for (int qt = 0; qt < QTS; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx[qt][r] = -INFINITY;
  }
  unsigned int staged = 0u;      // COMP select: entries in this wave's stage (wave-uniform)
  auto score_trip = [&](int64_t w0, const Trip& T) {
    if constexpr (MODE == kScanSample) {
#pragma unroll
      for (int qt = 0; qt < QTS; ++qt) {
        if (qt < n_qt) {
          if (n_qt > 1) load_query_tile(qt, A, ntlo);
          if constexpr (COMP) {
            // whole trips inside the corpus (all but a wave's last): the maxima of two tiles per v_max3 -- 8 VALU instructions per block
            // instead of 32 (an add and a max per score), which is what bounds this launch at 256 query rows
            bool whole = w0 + TU <= n_work;
#pragma unroll
            for (int u = 0; u < TU; ++u) whole = whole && T.in[u];
            if (__all(whole)) {
              static_assert(TU == 1 || TU % 2 == 0, "tile pairs");
#pragma unroll
              for (int u = 0; u < TU; u += 2) {
                cf32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c], T.Bv[u][c], acc0, 0, 0, 0);
                  if constexpr (TU > 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c], T.Bv[u + (TU > 1 ? 1 : 0)][c], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx[qt][r] = TU > 1 ? fmaxf(fmaxf(mx[qt][r], acc0[r]), acc1[r]) : fmaxf(mx[qt][r], acc0[r]);
              }
              continue;
            }
          }
#pragma unroll
          for (int u = 0; u < TU; ++u) {
            cf32x16 acc = {0};
#pragma unroll
            for (int c = 0; c < DC; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c], T.Bv[u][c], acc, 0, 0, 0);
            if (w0 + u < n_work) {                                   // a trip past the end holds the last row again: not a sample
              const float pen = T.in[u] ? 0.0f : -INFINITY;          // nor are the columns past the end of a ragged last tile
#pragma unroll
              for (int r = 0; r < 16; ++r) mx[qt][r] = fmaxf(mx[qt][r], acc[r] + pen);
            }
          }
        }
      }
      return;
    }
    if constexpr (COMP && MODE == kScanSelect) {
      // Component scans: 256 query rows (eight row tiles walk over every trip) and a candidate in every fifth to fifteenth block.
      unsigned long long in_m[TU];      // lanes whose item of tile u is inside the corpus (all of them but on a ragged last tile)
#pragma unroll
      for (int u = 0; u < TU; ++u) in_m[u] = __ballot(T.in[u]);
      for (int qt = 0; qt < n_qt; ++qt) {
        if (n_qt > 1) load_query_tile(qt, A, ntlo);
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          // pre-test as in the coarse scan (accumulator started at minus the bound, one sign test over the sixteen registers: 9 VALU
          // instructions per block; sixteen compares into scalar masks measured 2.2 x the whole scan's time) ...
          cf32x16 pre = ntlo;
#pragma unroll
          for (int c = 0; c < DC; ++c) pre = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c], T.Bv[u][c], pre, 0, 0, 0);
          if (__any(any_sign_clear(pre))) {
            // ... and a block that fires (one in five to fifteen here, where the coarse scan's fire once in a hundred) is scored again from
            // zero -- the materialising path's bits -- and walked register by register through the compares' scalar lane masks: no trip
            // through LDS to index a register, no LDS counter, no second look at the thresholds
            cf32x16 acc = {0};
#pragma unroll
            for (int c = 0; c < DC; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c], T.Bv[u][c], acc, 0, 0, 0);
            const int64_t t = (w0 + u) * step;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const unsigned long long mk = __builtin_amdgcn_fcmpf(acc[r], -ntlo[r], 3 /* oge */) & in_m[u];
              if (mk != 0ull) {      // scalar
                const float sc = bf16_rn(acc[r]);     // an un-rounded sum at or above the bound may round up to the threshold
                const float thr = coarse_unorderable(coarse_orderable(-ntlo[r]) + 0x10000u);
                const int q = qt * 32 + acc_row(r, h);
                const bool keep = T.in[u] && acc[r] >= -ntlo[r] && sc >= thr && q < B;
                stage_push_reg(stage_s[wave], staged, keep, ((unsigned long long)coarse_orderable(sc) << 32) | (unsigned int)(~(unsigned int)T.item[u]),
                               (unsigned int)(q * groups + gm), (int)(t % kSubLists), a.keys, a.counts, a.cap, lane);
              }
            }
            stage_flush_reg(stage_s[wave], staged, lane, a.keys, a.counts, a.cap, 64u);
          }
        }
      }
      return;
    }
    for (int qt = 0; qt < n_qt; ++qt) {
      if (n_qt > 1) load_query_tile(qt, A, ntlo);
      unsigned int fired = 0u;
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        cf32x16 acc = MODE == kScanSelect ? ntlo : cf32x16{0};
#pragma unroll
        for (int c = 0; c < DC; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c], T.Bv[u][c], acc, 0, 0, 0);
        if constexpr (MODE == kScanSelect) {
          if (__any(any_sign_clear(acc))) fired |= 1u << u;   // a tile past the end (the last row again) may fire: nothing of it is kept
        } else {
          const int64_t w = w0 + u;
          if (w < n_work) {
            const int64_t colx = T.item[u];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int q = qt * 32 + acc_row(r, h);
              if (q < B && T.in[u]) a.scores[((int64_t)q * groups + gm) * a.ld + colx] = bf16_rn(acc[r]);
            }
          }
        }
      }
      if constexpr (MODE == kScanSelect)
        if (fired) keep_candidates(qt, A, ntlo, T, fired, w0);
    }
  };
  int64_t w0 = gw * TU;
  if (MODE == kScanAll && w0 >= n_work) return;      // (the select scan's waves all meet at the workgroup's flush below)
  const int64_t hop = n_waves * TU;
  Trip T, N;
  if (w0 < n_work) {
  load_trip(w0, T);
  for (;;) {
    const int64_t w1 = w0 + hop;
    load_trip(w1, N);          // past the end: the last row again (in = false), harmless
    // The wait for T's loads belongs HERE, with N's still in flight: left to the loop over the query tiles below, the compiler's
    // counter bookkeeping merges the loop's entry and back edge into a wait for everything outstanding -- the prefetch included.
#pragma unroll
    for (int u = 0; u < TU; ++u)
#pragma unroll
      for (int c = 0; c < DC; ++c) asm volatile("" : "+v"(T.Bv[u][c]));
    score_trip(w0, T);
    if (w1 >= n_work) break;
    T = N;
    w0 = w1;
  }
  }
  if constexpr (MODE == kScanSelect && COMP) stage_flush_reg(stage_s[wave], staged, lane, a.keys, a.counts, a.cap, 1u);
  else if constexpr (MODE == kScanSelect) {
    __syncthreads();                                   // every wave is through its tiles: the list is complete
    const unsigned int n_st = wg_n < (unsigned int)kWgStage ? wg_n : (unsigned int)kWgStage;
    if (n_st > 0u) {                                   // workgroup-uniform
      const int rows_wg = n_qt * 32;
      const int sub = (int)(blockIdx.x % kSubLists), subcap = a.cap / kSubLists;
      for (int r = threadIdx.x; r < rows_wg; r += kScanThreads) row_cnt_s[r] = 0u;
      __syncthreads();
      for (unsigned int e = threadIdx.x; e < n_st; e += kScanThreads) wg_stage[e].pad = atomicAdd(&row_cnt_s[wg_stage[e].orow], 1u);   // the entry's place among its row's
      __syncthreads();
      for (int r = threadIdx.x; r < rows_wg; r += kScanThreads) {
        const unsigned int c = row_cnt_s[r];
        if (c) row_base_s[r] = a.counts ? atomicAdd(&a.counts[((int64_t)r * groups + gm) * kSubLists + sub], c) : 0u;   // (counts == NULL: RAILS_COMP_DEBUG=2)
      }
      __syncthreads();
      for (unsigned int e = threadIdx.x; e < n_st; e += kScanThreads) {
        const unsigned int r = wg_stage[e].orow, slot = row_base_s[r] + wg_stage[e].pad;
        if (slot < (unsigned int)subcap) a.keys[((int64_t)r * groups + gm) * a.cap + (int64_t)sub * subcap + slot] = wg_stage[e].key;
      }
    }
  }
  if constexpr (MODE == kScanSample) {   // a wave that saw no tile writes -inf: the row of maxima has no holes
#pragma unroll
    for (int qt = 0; qt < QTS; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qt * 32 + acc_row(r, h);
        if (q < B) a.scores16[((int64_t)q * groups + gm) * a.ld + gw * 32 + x] = (unsigned short)(__float_as_uint(bf16_rn(mx[qt][r])) >> 16);
      }
  }
}

constexpr int kSampleMaxQTComp = 8;   // ... and of a component sample launch: B * P_Q <= 256 query rows

template <int MODE>
static int launch_coarse_scan(const CoarseScanArgs& a, hipStream_t stream) {
  const int rows = a.comp ? a.B * a.PQ : a.B;
  const int groups = a.groups > 0 ? a.groups : 1;
  const int n_qt = (rows + 31) / 32;
  const int dc = a.d / 16;
  size_t lds = (size_t)n_qt * dc * 64 * 8 * sizeof(unsigned short) + 2 * (size_t)n_qt * 32 * sizeof(float);
  int stage_cap = 0;
  if (MODE == kScanSelect) {
    // + the flush's two counters per row and the workgroup's list of hits: 1 024 entries where three workgroups still fit a CU's 160 KiB next
    // to the kernel's 16 KiB of static LDS (the component scans' 256 rows at d = 32: 512 -- a third workgroup per CU is worth more there)
    lds += 2 * (size_t)n_qt * 32 * sizeof(unsigned int);
    stage_cap = (lds + 1024 * sizeof(StageEntry) + 17 * 1024) * 3 <= 156 * 1024 ? 1024 : 512;
    lds += (size_t)stage_cap * sizeof(StageEntry);
  }
  if (lds > 96 * 1024) { set_error("coarse scan: %d query rows x d %d do not fit LDS", rows, a.d); return kErrUnsupported; }
  const int64_t n_tiles = (a.n + 31) >> 5;
  const int64_t step = MODE == kScanSample ? a.stride : 1;
  const int64_t n_work = (n_tiles + step - 1) / step;
  constexpr int tu = MODE != kScanSelect ? 2 : 4;   // tiles per trip at d = 32 (fewer at larger d: then some waves get no trip)
  int64_t grid = (n_work + 4 * tu - 1) / (4 * tu);
  static const int64_t grid_cap = [] { const char* e = getenv("RAILS_SCAN_GRID"); const int64_t v = e ? atoll(e) : 0; return v > 0 ? v : (int64_t)2048; }();
  const int64_t cap_g = (grid_cap + groups - 1) / groups;   // 2048 workgroups over all groups: 8 workgroups of 4 waves per CU
  if (grid > cap_g) grid = cap_g;
  const bool wide = a.comp && n_qt > kSampleMaxQT;
  if (MODE == kScanSelect && n_qt * 32 > kSelectMaxRows) { set_error("coarse select scan: %d query rows exceed %d", rows, kSelectMaxRows); return kErrUnsupported; }
  if constexpr (MODE == kScanSample) {
    if (n_qt > (a.comp ? kSampleMaxQTComp : kSampleMaxQT)) { set_error("coarse sample scan: %d query rows exceed %d", rows, 32 * (a.comp ? kSampleMaxQTComp : kSampleMaxQT)); return kErrUnsupported; }
    grid = a.ld / (32 * (kScanThreads / 64));   // the plan's: every wave writes its 32 columns of the row's block of maxima
  }
  if (grid < 1) return kOk;
  CoarseScanArgs b = a;
  b.groups = groups;
  b.stage_cap = stage_cap;
  auto go = [&](auto nt) {
    constexpr bool NT = decltype(nt)::value;
    auto fire = [&](auto kernel) {
      if (lds > 48 * 1024 &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
        return false;
      hipLaunchKernelGGL(kernel, dim3((unsigned)grid, (unsigned)groups), dim3(kScanThreads), lds, stream, b);
      return true;
    };
    if constexpr (MODE == kScanSample) {
      if (a.comp) {      // the component sample: up to eight row tiles of running maxima, two tiles per v_max3
        if (wide) {
          switch (dc) {      // (d = 128 keeps four row tiles: eight would spill -- component_max_rows)
            case 2: return fire(&coarse_scan_kernel<2, MODE, NT, kSampleMaxQTComp, true>);
            case 4: return fire(&coarse_scan_kernel<4, MODE, NT, kSampleMaxQTComp, true>);
            default: return false;
          }
        }
        switch (dc) {
          case 2: return fire(&coarse_scan_kernel<2, MODE, NT, kSampleMaxQT, true>);
          case 4: return fire(&coarse_scan_kernel<4, MODE, NT, kSampleMaxQT, true>);
          case 8: return fire(&coarse_scan_kernel<8, MODE, NT, kSampleMaxQT, true>);
          default: return false;
        }
      }
    }
    if constexpr (MODE == kScanSelect) {
      // RAILS_COMP_SELECT=1 (measurement): the select block that walks a fired block's registers by the compares' scalar masks.  Measured at
      // amzn-books, B = 32, k_g = 5, stride 4: 198 us (116 without candidates, at two waves per SIMD) against 162 (96) for the coarse scan's
      // own block at three waves -- the default
      static const bool comp_block = [] { const char* e = getenv("RAILS_COMP_SELECT"); return e && atoi(e) == 1; }();
      if (a.comp && comp_block) {
        switch (dc) {
          case 2: return fire(&coarse_scan_kernel<2, MODE, NT, kSampleMaxQT, true>);
          case 4: return fire(&coarse_scan_kernel<4, MODE, NT, kSampleMaxQT, true>);
          case 8: return fire(&coarse_scan_kernel<8, MODE, NT, kSampleMaxQT, true>);
          default: return false;
        }
      }
    }
    switch (dc) {
      case 2: return fire(&coarse_scan_kernel<2, MODE, NT>);
      case 4: return fire(&coarse_scan_kernel<4, MODE, NT>);
      case 8: return fire(&coarse_scan_kernel<8, MODE, NT>);
      default: return false;
    }
  };
  bool known;
  if constexpr (MODE == kScanSelect && RAILS_SCAN_NT != 0) known = n_qt == 1 ? go(std::true_type{}) : go(std::false_type{});
  else known = go(std::false_type{});
  if (!known) { set_error("coarse scan: d = %d (supported: 32, 64, 128)", a.d); return kErrUnsupported; }
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int coarse_build(const Shape& s, const float* ipack, int64_t n, void* table, hipStream_t stream) {
  const int d = s.dot_product_dimension;
  const int64_t total = n * d;
  if (total == 0) return kOk;
  hipLaunchKernelGGL(coarse_build_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ipack, n,
                     s.query_dot_product_groups, s.item_dot_product_groups, d, static_cast<unsigned short*>(table));
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int coarse_score(const Shape& s, const float* eq, int B, int avg, const void* table, int64_t n, float* scores, int64_t ld,
                 hipStream_t stream, const int32_t* run_if) {
  if (B <= 0 || n <= 0) return kOk;
  CoarseScanArgs a{};
  a.eq = eq; a.B = B; a.PQ = s.query_dot_product_groups; a.d = s.dot_product_dimension; a.avg = avg; a.groups = 1;
  a.table = static_cast<const unsigned short*>(table); a.n = n; a.scores = scores; a.ld = ld; a.stride = 1;
  a.run_if = run_if;
  return launch_coarse_scan<kScanAll>(a, stream);
}

// ---- fused coarse top-K' -----------------------------------------------------------------------------------------
// Exact top-K' of the coarse scores without materialising the (B, N) score matrix (16 GB per 125 M-item shard at
// B = 32, and five more passes over it for the selection):
//   1. kScanSample scores every stride-th tile                      (N / stride items per query)
//   2. top-r of the sample -> thr[b] = its r-th largest score.  With m = K'/stride expected sample hits above the true
//      K'-th score, r = 2m + 4 sqrt(m) + 8 puts thr below it with overwhelming probability, while only ~r*stride items
//      of the corpus are expected at or above thr
//   3. kScanSelect streams the whole table once and appends the keys (score, position) with score >= thr[b] to query b's
//      candidate list (`cap` slots)
//   4. row_select over the candidate keys: the K' largest, ties by position -- exactly what top-K' over the materialised
//      scores returns, PROVIDED K' <= counts[b] <= cap for every query.  counts come back to the caller, who falls back
//      to the materialising path otherwise (heavy ties at the threshold, adversarial item order).
struct CoarseTopkPlan { int stride, r, cap; bool sample16; int64_t n_sample; size_t off_keys, off_sample, off_top_s, off_top_i, off_ws, off_qfrag, off_q8, off_qmeta, total, topk_ws; };

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

#ifndef RAILS_SAMPLE16
#define RAILS_SAMPLE16 1   // 0: fp32 threshold samples (measurement)
#endif
// group_max: the sample is the coarse scan's block of per-wave running maxima (kScanSample above) instead of every sampled score
// comp_rows > 0: the plan of the component scans (B = all B * P_Q * P_X rows; comp_rows = the B * P_Q query rows of one item group): a denser
// sample (the threshold's Poisson noise sets how many candidates the select scan appends: ~k + 6 sqrt(k stride) per row, and the appends are
// what that scan costs at 2 048 rows), a 32-workgroup block of maxima per item group and 2 048-slot lists
static bool coarse_topk_plan(int B, int64_t n, int k_prime, CoarseTopkPlan* p, bool group_max = false, int comp_rows = 0) {
  if (k_prime < 1 || k_prime > 4096 || n < k_prime) return false;
  if (group_max && (comp_rows > 0 ? comp_rows > 32 * kSampleMaxQTComp : B > 32 * kSampleMaxQT)) return false;
  const int64_t n_tiles = (n + 31) >> 5;
  // sample every stride-th tile: m = ~8 expected hits above the true K'-th score for large K', never denser than 1/64 of
  // the table (small K' just get fewer expected hits, and r below keeps the miss probability ~1e-9).  m = 16 (stride K'/16)
  // gave ~3.1 K' candidates instead of ~4.1 K' (of 8 K' slots: r * stride +- sqrt(r) * stride = 4.1 +- 0.7 K'), but the sample
  // and the selection of its r-th largest are B * n / stride of writes and reads: 0.75 ms of a 2.9 ms step at B = 128 on a
  // 125 M-item shard, 0.18 of 1.8 ms at B = 32 -- halved by the sparser sample.
  int stride = k_prime / 8;
  if (stride < 64) stride = 64;
  if (stride > 256) stride = 256;
  if (comp_rows > 0) {
    static const int forced = [] { const char* e = getenv("RAILS_COMP_STRIDE"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 256 ? v : 0; }();   // measurement override
    stride = forced ? forced : 4;      // sample + select scan at amzn-books, B = 32, k_g = 5 (round-6 kernels before the COMP select block): 16 -> 42 + 228 us, 4 -> 81 + 162, 1 -> 237 + 136
  }
  // Corpora of a few million items (round 6): the sample is cheap there (a 64 k-item sample of a 45 MB table is microseconds), and what the
  // step pays for is every candidate beyond K' -- appended by the scan, loaded and ranked by the key selection.  m = 64 expected hits and the
  // exact Poisson rank put ~1.8 K' candidates at or above the threshold where m = 8-16 and the Chernoff rank put 3-4 K'
  // (amzn-books, K' = 4 000: 16 400 -> 7 400 per query).  Shard-sized corpora keep the sparse sample above.
  const bool small_corpus = comp_rows == 0 && n <= (4ll << 20);
  if (small_corpus) {
    stride = k_prime / 64;
    const int64_t floor_stride = (n + 65535) / 65536;      // at most ~64 k sampled items per query
    if (stride < floor_stride) stride = (int)floor_stride;
    if (stride < 4) stride = 4;
    if (stride > 256) stride = 256;
  }
  // r = the smallest rank with P(Poisson(m) >= r) <= e^-m (e m / r)^r < 1e-9 (Chernoff), and at least 2m + 4 sqrt(m):
  // the r-th largest sample score is then below the true K'-th score except with negligible probability (and a miss
  // only costs the fallback), while ~r * stride items are expected at or above it
  auto r_of = [&](int st) {
    const float m = (float)k_prime / st;
    if (comp_rows > 0 || small_corpus) {
      // component scans: every candidate beyond the k_g wanted costs the select scan (a fired tile is ~3 x a quiet one, and at k_g = 100 a third
      // of the (tile, query tile) steps fire), so r is the EXACT smallest rank with P(Poisson(m) >= r) < 1e-9 per row -- the Poisson tail
      // dominates the binomial one of k_g items landing in every stride-th tile -- instead of the Chernoff rank with its 2m floor
      // (k_g = 100, stride 4: 59 instead of 72 -> ~ 240 instead of ~ 290 appended per row)
      double pmf = exp(-(double)m), below = 0.0;      // below = P(X <= r - 1)
      int r = 0;
      while (r < 512 && 1.0 - below >= 1e-9) { below += pmf; ++r; pmf *= (double)m / r; }
      return r < 2 ? 2 : r;
    }
    int r = (int)(2.0f * m + 4.0f * sqrtf(m)) + 2;
    while (r < 512 && (-m + r * (1.0f + logf(m / r))) > -20.7f) ++r;
    return r > 512 ? 512 : r;
  };
  while (stride > 1 && ((n_tiles + stride - 1) / stride) * 32 < 8 * (int64_t)r_of(stride)) stride /= 2;   // sample >= 8r
  const int r = r_of(stride);
  if ((float)k_prime / stride > 200.0f) return false;   // corpus too small for a sparse sample: nothing to gain
  p->stride = stride; p->r = r;
  p->n_sample = ((n_tiles + stride - 1) / stride) * 32;
  if (p->n_sample < r) return false;
  if (group_max) {
    // one column block of 32 maxima per wave: up to kSampleGrid workgroups (fewer for small samples: a wave takes whole trips
    // of up to four tiles), at least 16 so that the row has the 2 048 entries the register-resident selection wants; the waves beyond the
    // sample write -inf.  The threshold needs r real maxima; 4r keeps two of the top r from sharing a group too often.
    const int64_t n_work = (n_tiles + stride - 1) / stride;
    const int waves_per_wg = kScanThreads / 64;
    // (small corpora: a whole trip of four tiles per wave -- half the maxima for the threshold launch to rank: 32 768 -> ~17 000 per query at amzn-books)
    int64_t grid = (n_work + (small_corpus ? 4 : 2) * waves_per_wg - 1) / ((small_corpus ? 4 : 2) * waves_per_wg);
    if (grid > (comp_rows > 0 ? 32 : kSampleGrid)) grid = comp_rows > 0 ? 32 : kSampleGrid;
    const int64_t trips = (n_work + 3) / 4;   // a wave's trip is up to four tiles (d = 32)
    const int64_t groups = (trips < grid * waves_per_wg ? trips : grid * waves_per_wg) * 32;
    if (groups < 4 * (int64_t)r) return false;
    if (grid < 16) grid = 16;
    p->n_sample = grid * waves_per_wg * 32;
  }
  int cap = (small_corpus ? 4 : 8) * k_prime;      // (~1.8 K' candidates expected under the dense sample, 3-4 K' under the sparse one)
  if (cap < (comp_rows > 0 ? 2048 : 4096)) cap = comp_rows > 0 ? 2048 : 4096;
  if (cap > 24 * 1024) cap = 24 * 1024;
  cap = (cap + kSubLists * 4 - 1) / (kSubLists * 4) * (kSubLists * 4);
  p->cap = cap;
  size_t o = align256(sizeof(unsigned int) * (size_t)B * kSubLists);
  p->off_keys = o; o += align256(sizeof(unsigned long long) * (size_t)B * cap);
  // the sample holds bf16 values: kept as 16-bit patterns where the selection of its r-th largest reads them (B * n_sample
  // elements written by the sample scan and read back once: 0.5 GB per batch of 128 on a 125 M-item shard as fp32)
  p->sample16 = (RAILS_SAMPLE16 || group_max) && topk_bf16_source_ok(B, p->n_sample, r);
  if (group_max && !p->sample16) return false;
  p->off_sample = o; o += align256((p->sample16 ? sizeof(unsigned short) : sizeof(float)) * (size_t)B * p->n_sample);
  p->off_top_s = o; o += align256(sizeof(float) * (size_t)B * r);
  p->off_top_i = o; o += align256(sizeof(int64_t) * (size_t)B * r);
  p->topk_ws = topk_workspace_bytes(B, p->n_sample, r);
  p->off_ws = o; o += align256(p->topk_ws);
  p->off_qfrag = o; o += align256(sizeof(unsigned short) * (size_t)(((comp_rows > 0 ? comp_rows : B) + 31) / 32) * 32 * 128);   // the query fragments (d <= 128)
  p->off_q8 = o; o += align256((size_t)((B + 31) / 32) * 32 * 128);                                // the same as int8 (pre-filter)
  p->off_qmeta = o; o += align256(sizeof(float) * 2 * (size_t)((B + 31) / 32) * 32);
  p->total = o;
  return true;
}

// ---- int8 pre-filter of the select scan -------------------------------------------------------------------------------------------
// The select scan is bound by reading the table: 2d bytes per item.  A second copy of the table as int8 with ONE scale for the whole
// table (d bytes per item) is enough to decide which tiles can hold a candidate at all:
//   x_k = s (i_k + e_k), q_k = s_q (j_k + f_k), |e_k|, |f_k| <= 1/2   =>
//   | sum_k q_k x_k  -  s s_q sum_k j_k i_k |  <=  eps_b = s |q_b|_1 / 2 + s_q,b max_x |x|_1 / 2 + 3 s s_q,b d / 4
// so an item whose bf16 score reaches query b's threshold has the INTEGER dot product I >= (thr_lo,b - eps_b) / (s s_q,b), thr_lo the
// bf16 value below the threshold (the margin the bf16 pre-test keeps for the rounding of the sum).  The scan streams the int8 table,
// one v_mfma_i32_32x32x32_i8 per tile and query tile with the accumulator started at minus that integer bound -- the same sign-bit
// pre-test as the bf16 scan -- and only the tiles that fire (a few per cent) are read from the bf16 table and scored as before, with
// the materialising path's bits: candidates, counts and keys are exactly those of the scan without the pre-filter.  HBM bytes per
// batch: N d (+ 2d per item of a fired tile) instead of 2 N d.
// The buffer: 256 bytes of header [scale, 127 / max|x|, max_x |x|_1, ..., fired, scanned] then N * d int8, item-major like the table.
constexpr size_t kPrefilterHeader = 256;
struct PrefilterHeader {
  float scale, inv_scale, x1max; unsigned int maxabs_bits, x1max_bits; unsigned int pad[3];
  unsigned long long fired, scanned;   // byte 32 / 40: (tile, query tile) blocks that fired / were tested, summed over the select scans so far --
};                                     // a caller that sees most of them fire (a table whose scale is set by a few outliers) drops the pre-filter
static_assert(sizeof(PrefilterHeader) <= 256 && offsetof(PrefilterHeader, fired) == 32, "header layout (include/rails_amd.h)");

__global__ void prefilter_stats_kernel(const unsigned short* __restrict__ table, int64_t n, int d, PrefilterHeader* hdr) {
  const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float mx = 0.0f, l1 = 0.0f;
  if (item < n) {
    const unsigned short* row = table + item * d;
    for (int k = 0; k < d; k += 8) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + k);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float a = fabsf((float)v[j]); mx = fmaxf(mx, a); l1 += a; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); l1 = fmaxf(l1, __shfl_xor(l1, o, 64)); }
  if ((threadIdx.x & 63) == 0) {   // non-negative floats order like their bit patterns.  fmaxf drops NaN: an item with a NaN sum never passes the
    atomicMax(&hdr->maxabs_bits, __float_as_uint(mx));   // exact test, so it need not enter the bound; inf stays inf
    atomicMax(&hdr->x1max_bits, __float_as_uint(l1));
  }
}
__global__ void prefilter_finish_kernel(PrefilterHeader* hdr) {
  const float mx = __uint_as_float(hdr->maxabs_bits);
  hdr->scale = mx > 0.0f ? mx / 127.0f : 1.0f;              // inf stays inf: every tile then fires (the bound below becomes -inf)
  hdr->inv_scale = mx > 0.0f ? 127.0f / mx : 1.0f;
  hdr->x1max = __uint_as_float(hdr->x1max_bits);
}
__device__ __forceinline__ int quant8(float v, float inv_scale) {
  const float t = rintf(v * inv_scale);
  return t >= -127.0f && t <= 127.0f ? (int)t : 0;           // NaN (and inf / inf): 0 -- such an item never passes the exact test either
}
__global__ void prefilter_quant_kernel(const unsigned short* __restrict__ table, int64_t n_elems, const PrefilterHeader* __restrict__ hdr,
                                       signed char* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i >= n_elems) return;
  const float inv = hdr->inv_scale;
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(table + i), b = *reinterpret_cast<const bf16x8*>(table + i + 8);
  unsigned int w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    w[j >> 2] |= ((unsigned int)quant8((float)a[j], inv) & 255u) << (8 * (j & 3));
    w[2 + (j >> 2)] |= ((unsigned int)quant8((float)b[j], inv) & 255u) << (8 * (j & 3));
  }
  *reinterpret_cast<uint4*>(out + i) = uint4{w[0], w[1], w[2], w[3]};
}

size_t coarse_prefilter_bytes(const Shape& s, int64_t n) {
  const int d = s.dot_product_dimension;
  return d % 32 == 0 && d <= 128 && n > 0 ? kPrefilterHeader + (size_t)n * d : 0;
}
int coarse_prefilter_build(const Shape& s, const void* table, int64_t n, void* prefilter, hipStream_t stream) {
  const int d = s.dot_product_dimension;
  if (coarse_prefilter_bytes(s, n) == 0) { set_error("coarse pre-filter: d = %d (supported: 32, 64, 128)", d); return kErrUnsupported; }
  PrefilterHeader* hdr = static_cast<PrefilterHeader*>(prefilter);
  if (hipMemsetAsync(hdr, 0, kPrefilterHeader, stream) != hipSuccess) return kErrLaunch;
  hipLaunchKernelGGL(prefilter_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, static_cast<const unsigned short*>(table), n, d, hdr);
  hipLaunchKernelGGL(prefilter_finish_kernel, dim3(1), dim3(1), 0, stream, hdr);
  const int64_t n_elems = n * d;
  hipLaunchKernelGGL(prefilter_quant_kernel, dim3((unsigned)((n_elems / 16 + 255) / 256)), dim3(256), 0, stream, static_cast<const unsigned short*>(table), n_elems,
                     hdr, static_cast<signed char*>(prefilter) + kPrefilterHeader);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

#ifndef RAILS_SCAN8_WAVES
#define RAILS_SCAN8_WAVES 3
#endif
#ifndef RAILS_SCAN8_TU
#define RAILS_SCAN8_TU 8      // KiB of int8 table per wave and trip (tiles of d = 32)
#endif
typedef int ci32x4 __attribute__((ext_vector_type(4)));
typedef int ci32x16 __attribute__((ext_vector_type(16)));

// The queries for the int8 scan, from their bf16 A fragments in LDS (fragment order [qt][c][h * 32 + row][8]): one thread per query
// writes its int8 fragment bytes [qt][c8][h * 32 + row][16] (k = 32 c8 + 16 h + j) and (s_q, |q|_1).
__device__ __forceinline__ void quantise_query(const unsigned short* qfrag, int DC, int d, int q, signed char* q8, float* qmeta) {
  const int qt = q >> 5, row = q & 31;
  auto at = [&](int dd) { return bf16_to_f32(qfrag[(((size_t)qt * DC + (dd >> 4)) * 64 + ((dd >> 3) & 1) * 32 + row) * 8 + (dd & 7)]); };
  float mx = 0.0f, l1 = 0.0f;
  for (int dd = 0; dd < d; ++dd) { const float a = fabsf(at(dd)); mx = fmaxf(mx, a); l1 += a; }
  const float inv = mx > 0.0f ? 127.0f / mx : 1.0f;
  for (int dd = 0; dd < d; ++dd)
    q8[(((size_t)qt * (d / 32) + (dd >> 5)) * 64 + ((dd >> 4) & 1) * 32 + row) * 16 + (dd & 15)] = (signed char)quant8(at(dd), inv);
  qmeta[2 * q] = mx > 0.0f ? mx / 127.0f : 1.0f;
  qmeta[2 * q + 1] = l1;
}

struct CoarseI8Args {
  const unsigned short* qfrag; const signed char* q8; const float* qmeta;   // made by the sample scan's workgroup 0
  const unsigned short* table; const signed char* table8; PrefilterHeader* hdr; int64_t n; int B, d;
  const float* thr; int64_t thr_stride;
  unsigned long long* keys; int cap; unsigned int* counts;
};

template <int DC8, bool NT>   // DC8 = d / 32 K chunks of the int8 MFMA; NT: non-temporal table loads
__global__ __launch_bounds__(kScanThreads) __attribute__((amdgpu_waves_per_eu(RAILS_SCAN8_WAVES, RAILS_SCAN8_WAVES))) void coarse_scan_i8_kernel(CoarseI8Args a) {
  constexpr int DC = 2 * DC8;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];   // bf16 fragments, int8 fragments, thr, thr_lo, integer starts
  const int d = a.d, B = a.B;
  const int n_qt = (B + 31) / 32;
  unsigned short* qfrag = lds;                                                             // [n_qt][DC][64][8]
  signed char* q8 = reinterpret_cast<signed char*>(qfrag + (size_t)n_qt * DC * 64 * 8);   // [n_qt][DC8][64][16]
  float* thr_s = reinterpret_cast<float*>(q8 + (size_t)n_qt * DC8 * 64 * 16);             // [n_qt * 32]
  float* tlo_s = thr_s + n_qt * 32;                                                        // the bf16 value below thr
  int* nb_s = reinterpret_cast<int*>(tlo_s + n_qt * 32);                                   // minus the integer bound
  __shared__ StageEntry stage_s[kScanThreads / 64][kStage];
  __shared__ unsigned int stage_n[kScanThreads / 64];
  __shared__ float acc_s[(kScanThreads / 64) * 16 * 64];
  if (threadIdx.x < kScanThreads / 64) stage_n[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < n_qt * DC * 64; i += kScanThreads) reinterpret_cast<bf16x8*>(qfrag)[i] = reinterpret_cast<const bf16x8*>(a.qfrag)[i];
  for (int i = threadIdx.x; i < n_qt * DC8 * 64; i += kScanThreads) reinterpret_cast<ci32x4*>(q8)[i] = reinterpret_cast<const ci32x4*>(a.q8)[i];
  for (int i = threadIdx.x; i < n_qt * 32; i += kScanThreads) {
    const float thr = i < B ? a.thr[(int64_t)i * a.thr_stride] : INFINITY;
    const float tlo = coarse_unorderable(coarse_orderable(thr) - 0x10000u);
    const float s = a.hdr->scale, sq = a.qmeta[2 * i], l1 = a.qmeta[2 * i + 1];
    const float eps = 1.002f * (0.5f * s * l1 + 0.5f * sq * a.hdr->x1max + 0.75f * s * sq * (float)d);
    float bound = floorf((tlo - eps) / (s * sq)) - 2.0f;          // integer dot products below it cannot reach thr
    if (!(bound > -1073741824.0f)) bound = -1073741824.0f;        // -inf / NaN (a table with inf or NaN sums): every tile fires
    if (bound > 1073741824.0f) bound = 1073741824.0f;             // rows past B (thr = inf): never
    thr_s[i] = thr;
    tlo_s[i] = tlo;
    nb_s[i] = -(int)bound;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = lane & 31, h = lane >> 5;
  const int64_t n_tiles = (a.n + 31) >> 5;
  const int64_t gw = (int64_t)blockIdx.x * (kScanThreads / 64) + wave, n_waves = (int64_t)gridDim.x * (kScanThreads / 64);
  constexpr int TU = RAILS_SCAN8_TU / DC8;   // item tiles per trip: 8 KiB of table per wave in flight
  struct Trip { ci32x4 Bq[TU][DC8]; };
  auto load_trip = [&](int64_t w0, Trip& T) {
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      int64_t item = (w0 + u) * 32 + x;
      if (item >= a.n) item = a.n - 1;
      const signed char* rowp = a.table8 + item * d + 16 * h;
#pragma unroll
      for (int c = 0; c < DC8; ++c) {
        if constexpr (NT) T.Bq[u][c] = __builtin_nontemporal_load(reinterpret_cast<const ci32x4*>(rowp + 32 * c));
        else T.Bq[u][c] = *reinterpret_cast<const ci32x4*>(rowp + 32 * c);
      }
    }
  };
  // a fired tile: its rows of the bf16 table, scored from zero like the materialising path, the scores at or above the threshold appended
  // Only the COLUMNS whose pre-test fired are read (`need`: this lane's item, in either half of its rows): the bound holds item by
  // item, so the other columns of the tile cannot hold a candidate of this query tile -- 64 B of bf16 table per suspect instead of
  // the tile's 2 KiB (the fired tiles' rows were 0.65 GB of the launch's 4.65 GB).
  auto exact_tile = [&](int qt, int64_t tile, bool need) {
    int64_t item = tile * 32 + x;
    const bool in = item < a.n;
    if (!in) item = a.n - 1;
    const unsigned short* rowp = a.table + item * d + 8 * h;
    bf16x8 Bu[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) Bu[c] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (need) {
#pragma unroll
      for (int c = 0; c < DC; ++c) Bu[c] = *reinterpret_cast<const bf16x8*>(rowp + 16 * c);
    }
    cf32x16 acc = {0};
#pragma unroll
    for (int c = 0; c < DC; ++c)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(qfrag + (((size_t)qt * DC + c) * 64 + lane) * 8), Bu[c], acc, 0, 0, 0);
    unsigned int mask = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) mask |= acc[r] >= tlo_s[qt * 32 + acc_row(r, h)] ? 1u << r : 0u;
    if (!in || !need) mask = 0u;
    if (__any(mask != 0u)) {
      float* mine = acc_s + (wave * 16) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[r * 64] = acc[r];
      while (__any(mask != 0u)) {
        if (mask != 0u) {
          const int r = __ffs(mask) - 1;
          mask &= mask - 1u;
          const int q = qt * 32 + acc_row(r, h);
          const float sc = bf16_rn(mine[r * 64]);
          if (q < B && sc >= thr_s[q])
            stage_push(stage_s[wave], &stage_n[wave], a.keys, a.counts, a.cap, (int)(tile % kSubLists), (unsigned int)q,
                       ((unsigned long long)coarse_orderable(sc) << 32) | (unsigned int)(~(unsigned int)item));
        }
      }
    }
    stage_flush_mixed(stage_s[wave], &stage_n[wave], lane, a.keys, a.counts, a.cap, 64u);
  };
  ci32x4 A8[DC8];
  ci32x16 nb;
  auto load_query_tile = [&](int qt) {
#pragma unroll
    for (int c = 0; c < DC8; ++c) A8[c] = *reinterpret_cast<const ci32x4*>(q8 + (((size_t)qt * DC8 + c) * 64 + lane) * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) nb[r] = nb_s[qt * 32 + acc_row(r, h)];
  };
  load_query_tile(0);
  unsigned int n_fired = 0u;
  int64_t w0 = gw * TU;
  if (w0 < n_tiles) {
    const int64_t hop = n_waves * TU;
    Trip T, N;
    load_trip(w0, T);
    for (;;) {
      const int64_t w1 = w0 + hop;
      load_trip(w1 < n_tiles ? w1 : w0, N);       // past the end: this trip again, harmless
#pragma unroll
      for (int u = 0; u < TU; ++u)
#pragma unroll
        for (int c = 0; c < DC8; ++c) asm volatile("" : "+v"(T.Bq[u][c]));   // the wait for T belongs here, with N in flight (see coarse_scan_kernel)
      for (int qt = 0; qt < n_qt; ++qt) {
        if (n_qt > 1) load_query_tile(qt);
        unsigned int fired = 0u, mine = 0u;   // tiles of the trip with a suspect (wave-uniform); this lane's own suspects
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          ci32x16 acc = nb;
#pragma unroll
          for (int c = 0; c < DC8; ++c) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A8[c], T.Bq[u][c], acc, 0, 0, 0);
          const bool hit = any_sign_clear(__builtin_bit_cast(cf32x16, acc));
          if (__any(hit)) fired |= 1u << u;
          mine |= hit ? 1u << u : 0u;
        }
        if (fired) {   // wave-uniform; rare
          mine |= (unsigned int)__shfl_xor((int)mine, 32, 64);   // the other half of the column's rows
          n_fired += (unsigned int)__popc(fired);
          while (fired) {
            const int u = __ffs(fired) - 1;
            fired &= fired - 1u;
            if (w0 + u < n_tiles) exact_tile(qt, w0 + u, ((mine >> u) & 1u) != 0u);
          }
        }
      }
      if (w1 >= n_tiles) break;
      T = N;
      w0 = w1;
    }
  }
  stage_flush_mixed(stage_s[wave], &stage_n[wave], lane, a.keys, a.counts, a.cap, 1u);
  if (lane == 0 && n_fired) atomicAdd(&a.hdr->fired, (unsigned long long)n_fired);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.hdr->scanned, (unsigned long long)(n_tiles * n_qt));
}

// dynamic LDS of the int8 select scan: the queries' bf16 fragments, their int8 fragments and three bound rows per query tile
static size_t coarse_scan_i8_lds(int B, int d) {
  const int n_qt = (B + 31) / 32, dc8 = d / 32;
  return (size_t)n_qt * (2 * dc8 * 1024 + dc8 * 1024 + 3 * 32 * 4);
}
// d = 128 with 97..128 queries needs 50 688 B, over the 48 KiB a launch gets without opting in: coarse_topk() then runs the bf16 select
// scan, whose output is the same bit for bit (the pre-filter only decides WHICH tiles are scored from the bf16 table)
static bool coarse_scan_i8_fits(int B, int d) { return (d == 32 || d == 64 || d == 128) && coarse_scan_i8_lds(B, d) <= 48 * 1024; }

static int launch_coarse_scan_i8(const CoarseI8Args& a, hipStream_t stream) {
  const int n_qt = (a.B + 31) / 32, dc8 = a.d / 32;
  const size_t lds = coarse_scan_i8_lds(a.B, a.d);
  if (lds > 48 * 1024) { set_error("coarse int8 scan: batch %d x d %d does not fit LDS", a.B, a.d); return kErrUnsupported; }
  const int64_t n_tiles = (a.n + 31) >> 5;
  const int tu = RAILS_SCAN8_TU / dc8;
  int64_t grid = (n_tiles + 4 * tu - 1) / (4 * tu);
  static const int64_t grid_cap = [] { const char* e = getenv("RAILS_SCAN8_GRID"); const int64_t v = e ? atoll(e) : 0; return v > 0 ? v : (int64_t)2048; }();
  if (grid > grid_cap) grid = grid_cap;
  if (grid < 1) return kOk;
  auto go = [&](auto nt) {
    constexpr bool NT = decltype(nt)::value;
    switch (dc8) {
      case 1: hipLaunchKernelGGL((coarse_scan_i8_kernel<1, NT>), dim3((unsigned)grid), dim3(kScanThreads), lds, stream, a); return true;
      case 2: hipLaunchKernelGGL((coarse_scan_i8_kernel<2, NT>), dim3((unsigned)grid), dim3(kScanThreads), lds, stream, a); return true;
      case 4: hipLaunchKernelGGL((coarse_scan_i8_kernel<4, NT>), dim3((unsigned)grid), dim3(kScanThreads), lds, stream, a); return true;
      default: return false;
    }
  };
  const bool known = (RAILS_SCAN_NT != 0 && n_qt == 1) ? go(std::true_type{}) : go(std::false_type{});
  if (!known) { set_error("coarse int8 scan: d = %d (supported: 32, 64, 128)", a.d); return kErrUnsupported; }
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

size_t coarse_topk_workspace_bytes(const Shape& s, int B, int64_t n, int k_prime) {
  CoarseTopkPlan p;
  return coarse_topk_plan(B, n, k_prime, &p, true) ? p.total : 0;
}

int coarse_topk_capacity(int B, int64_t n, int k_prime) {
  CoarseTopkPlan p;
  return coarse_topk_plan(B, n, k_prime, &p, true) ? p.cap : 0;
}

// flag |= any(v[i] < lo || v[i] > hi): the validity check of a fused scan's candidate counts, on the device
__global__ void range_flag_kernel(const int32_t* __restrict__ v, int n, int lo, int hi, int32_t* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (v[i] < lo || v[i] > hi)) atomicOr(flag, 1);
}
int range_flag(const int32_t* v, int n, int lo, int hi, int32_t* flag, hipStream_t stream) {
  if (n <= 0) return kOk;
  hipLaunchKernelGGL(range_flag_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, v, n, lo, hi, flag);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int coarse_topk(const Shape& s, const float* eq, int B, int avg, const void* table, int64_t n, int k_prime, void* ws,
                size_t ws_bytes, float* out_scores, int64_t* out_pos, int32_t* out_counts, int32_t* out_flag, void* prefilter, int n_cu,
                hipStream_t stream) {
  CoarseTopkPlan p;
  if (!coarse_topk_plan(B, n, k_prime, &p, true)) { set_error("coarse_topk: unsupported size (B = %d, K' = %d, n = %lld)", B, k_prime, (long long)n); return kErrUnsupported; }
  if (n >= (1ll << 32)) { set_error("coarse_topk: n does not fit 32-bit positions; shard the corpus"); return kErrUnsupported; }
  if (ws_bytes < p.total) { set_error("coarse_topk: workspace too small"); return kErrNoMem; }
  if (prefilter && !coarse_scan_i8_fits(B, s.dot_product_dimension)) prefilter = nullptr;   // decided BEFORE anything is enqueued: the bf16 select scan, same output
  char* base = static_cast<char*>(ws);
  unsigned int* counts = reinterpret_cast<unsigned int*>(base);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(base + p.off_keys);
  unsigned short* sample = reinterpret_cast<unsigned short*>(base + p.off_sample);
  float* top_s = reinterpret_cast<float*>(base + p.off_top_s);
  int64_t* top_i = reinterpret_cast<int64_t*>(base + p.off_top_i);
  unsigned short* frag = reinterpret_cast<unsigned short*>(base + p.off_qfrag);

  // Four launches: sample scan (which also makes the queries' A fragments for the select scan and zeroes the candidate counters),
  // the r-th largest of each row of maxima, select scan, key selection (which also reports the counts and raises out_flag).  The candidate lists are
  // not zeroed: the key selection reads the filled slots only.
  CoarseScanArgs a{};
  a.eq = eq; a.B = B; a.PQ = s.query_dot_product_groups; a.d = s.dot_product_dimension; a.avg = avg; a.groups = 1;
  a.table = static_cast<const unsigned short*>(table); a.n = n;
  a.qfrag_out = frag; a.zero_words = counts; a.n_zero = B * kSubLists; a.zero_flag = out_flag;
  signed char* q8 = reinterpret_cast<signed char*>(base + p.off_q8);
  float* qmeta = reinterpret_cast<float*>(base + p.off_qmeta);
  if (prefilter) { a.q8_out = q8; a.qmeta_out = qmeta; }
  a.scores16 = sample; a.ld = p.n_sample; a.stride = p.stride;
  int rc = launch_coarse_scan<kScanSample>(a, stream);
  if (rc != kOk) return rc;
  rc = topk(nullptr, p.n_sample, B, p.n_sample, p.r, nullptr, 0, top_s, top_i, base + p.off_ws, p.topk_ws, n_cu, stream, nullptr, 0, 0, sample);
  if (rc != kOk) return rc;
  if (prefilter) {   // the select scan over the int8 copy of the table; fired tiles are scored from the bf16 table
    CoarseI8Args i8{};
    i8.qfrag = frag; i8.q8 = q8; i8.qmeta = qmeta;
    i8.table = static_cast<const unsigned short*>(table); i8.hdr = static_cast<PrefilterHeader*>(prefilter);   // the header's two statistics words are updated
    i8.table8 = static_cast<const signed char*>(prefilter) + kPrefilterHeader; i8.n = n; i8.B = B; i8.d = a.d;
    i8.thr = top_s + (p.r - 1); i8.thr_stride = p.r; i8.keys = keys; i8.cap = p.cap; i8.counts = counts;
    rc = launch_coarse_scan_i8(i8, stream);
  } else {
    a.qfrag = frag; a.qfrag_out = nullptr; a.zero_words = nullptr; a.n_zero = 0; a.zero_flag = nullptr; a.q8_out = nullptr; a.qmeta_out = nullptr;
    a.scores16 = nullptr; a.ld = 0; a.stride = 1;
    a.thr = top_s + (p.r - 1); a.thr_stride = p.r; a.keys = keys; a.cap = p.cap; a.counts = counts;
    static const int dbg = [] { const char* e = getenv("RAILS_COMP_DEBUG"); return e ? atoi(e) : 0; }();
    a.no_hits = dbg & 1;
    if (dbg & 2) a.counts = nullptr;      // (measurement: the appends without their global atomics -- wrong results)
    rc = launch_coarse_scan<kScanSelect>(a, stream);
  }
  if (rc != kOk) return rc;
  return select_sublists(keys, counts, B, p.cap, kSubLists, k_prime, out_scores, out_pos, out_counts, out_flag, stream);
}

// ---------------------------------------------------------------------------------------------
// Per-component candidate generation of MoLNaiveTopK / MoLCombTopK (reference rails/indexing/mol_top_k.py:
// component table :61-73 and :172-174, scoring :242-255 / :495-506).  For every query group i and item group m the
// reference scores  bf16(Eq[b,i,:]) . bf16(Ex[x,m,:])  with a bf16 mm and takes the top k_per_group per (b, m) row.
//   table[m][x][:] = bf16(Ex[x,m,:])                                     2*P_X*d bytes per item, ITEM-GROUP-major (round 6)
//   score[(b*P_Q + i)*P_X + m][x] = bf16( sum_d bf16(Eq[b,i,d]) * table[m][x][d] )   fp32 holding bf16 values
// Row order (b, i, m) makes the top-k output reshape to (B, P_Q*P_X*k_g) directly.
// Round 6: the scans ARE the coarse scan (coarse_scan_kernel with groups = P_X: blockIdx.y picks the item group's table, the B * P_Q
// sub-embedding rows are the query rows) -- double-buffered trips of tiles, the query tiles walking over a trip in registers, thresholds in
// the accumulator start -- where rounds 1-5 ran a kernel of their own that re-read a threshold block from LDS for every (item group, query
// tile) of every tile and took sixteen compares per block (0.40-0.57 ms per batch of 32 at amzn-books: 0.07 of the 356 MB table's HBM
// time); the table is item-group-major so that a wave streams one group's rows back to back.
// ---------------------------------------------------------------------------------------------
__global__ void component_build_kernel(const float* __restrict__ ipack, int64_t n, int PQ, int PX, int d, int64_t n_total, int64_t first,
                                       unsigned short* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int row = PX * d;
  if (i >= n * row) return;
  const int64_t item = i / row;
  const int rem = (int)(i - item * row);
  const int m = rem / d, dd = rem - m * d;
  const int64_t tile = item >> 5;
  const int x = (int)(item & 31);
  const float* tEx = ipack + tile * (int64_t)(kTileItems * (PX * d + PQ * PX));
  const int hi = dd / (d / 2), s = dd - hi * (d / 2);
  table[((int64_t)m * n_total + first + item) * d + dd] = bf16_bits(tEx[((m * (d / 8) + (s >> 2)) * 64 + hi * 32 + x) * 4 + (s & 3)]);
}

int component_build(const Shape& s, const float* ipack, int64_t n, void* table, int64_t n_total, int64_t first, hipStream_t stream) {
  const int64_t total = n * s.item_dot_product_groups * s.dot_product_dimension;
  if (total == 0) return kOk;
  hipLaunchKernelGGL(component_build_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ipack, n,
                     s.query_dot_product_groups, s.item_dot_product_groups, s.dot_product_dimension, n_total, first,
                     static_cast<unsigned short*>(table));
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

static void component_args(const Shape& s, const float* eq, int B, const void* table, int64_t n, CoarseScanArgs* a) {
  *a = CoarseScanArgs{};
  a->eq = eq; a->B = B; a->PQ = s.query_dot_product_groups; a->d = s.dot_product_dimension; a->avg = 0;
  a->groups = s.item_dot_product_groups; a->comp = 1; a->group_stride = n * (int64_t)s.dot_product_dimension;
  a->table = static_cast<const unsigned short*>(table); a->n = n;
}

int component_score(const Shape& s, const float* eq, int B, const void* table, int64_t n, float* scores, int64_t ld,
                    hipStream_t stream, const int32_t* run_if) {
  if (B <= 0 || n <= 0) return kOk;
  CoarseScanArgs a;
  component_args(s, eq, B, table, n, &a);
  a.scores = scores; a.ld = ld; a.stride = 1; a.run_if = run_if;
  return launch_coarse_scan<kScanAll>(a, stream);
}

// Fused per-component top-k_g: the fused coarse top-K' scheme (running-maxima sample, threshold, one select scan into sub-lists, radix key
// selection: four launches, no memset) over the B * P_Q * P_X (query group, item group) rows, instead of a (rows, N) score matrix (5.7 GB at
// amzn-books, B = 32).  out_flag (optional): raised when a row's candidate count left [k_group, capacity] (the caller redoes the call on
// the materialising path); zeroed by the first launch.
// query rows (B * P_Q) one fused component call takes: the sample launch keeps their running maxima in registers
static int component_max_rows(const Shape& s) { return 32 * (s.dot_product_dimension >= 128 ? kSampleMaxQT : kSampleMaxQTComp); }

size_t component_topk_workspace_bytes(const Shape& s, int B, int64_t n, int k_group) {
  if (B * s.query_dot_product_groups > component_max_rows(s)) return 0;
  CoarseTopkPlan p;
  return coarse_topk_plan(B * s.query_dot_product_groups * s.item_dot_product_groups, n, k_group, &p, true, B * s.query_dot_product_groups) ? p.total : 0;
}

int component_topk(const Shape& s, const float* eq, int B, const void* table, int64_t n, int k_group, void* ws, size_t ws_bytes,
                   float* out_scores, int64_t* out_pos, int32_t* out_counts, int32_t* out_flag, int n_cu, hipStream_t stream) {
  const int rows = B * s.query_dot_product_groups * s.item_dot_product_groups;
  CoarseTopkPlan p;
  if (B * s.query_dot_product_groups > component_max_rows(s) || !coarse_topk_plan(rows, n, k_group, &p, true, B * s.query_dot_product_groups)) { set_error("component_topk: unsupported size (batch = %d, k = %d, n = %lld)", B, k_group, (long long)n); return kErrUnsupported; }
  if (n >= (1ll << 32)) { set_error("component_topk: n does not fit 32-bit positions; shard the corpus"); return kErrUnsupported; }
  if (ws_bytes < p.total) { set_error("component_topk: workspace too small"); return kErrNoMem; }
  char* base = static_cast<char*>(ws);
  unsigned int* counts = reinterpret_cast<unsigned int*>(base);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(base + p.off_keys);
  unsigned short* sample = reinterpret_cast<unsigned short*>(base + p.off_sample);
  float* top_s = reinterpret_cast<float*>(base + p.off_top_s);
  int64_t* top_i = reinterpret_cast<int64_t*>(base + p.off_top_i);
  unsigned short* frag = reinterpret_cast<unsigned short*>(base + p.off_qfrag);
  CoarseScanArgs a;
  component_args(s, eq, B, table, n, &a);
  a.qfrag_out = frag; a.zero_words = counts; a.n_zero = rows * kSubLists; a.zero_flag = out_flag;
  a.scores16 = sample; a.ld = p.n_sample; a.stride = p.stride;
  int rc = launch_coarse_scan<kScanSample>(a, stream);
  if (rc != kOk) return rc;
  (void)top_i; (void)n_cu;
  rc = bf16_rows_kth(sample, p.n_sample, rows, (int)p.n_sample, p.r, top_s, stream);      // thr[row] = the r-th largest of the row's maxima
  if (rc != kOk) return rc;
  a.qfrag = frag; a.qfrag_out = nullptr; a.zero_words = nullptr; a.n_zero = 0; a.zero_flag = nullptr;
  a.scores16 = nullptr; a.ld = 0; a.stride = 1;
  a.thr = top_s; a.thr_stride = 1; a.keys = keys; a.cap = p.cap; a.counts = counts;
  static const int dbg = [] { const char* e = getenv("RAILS_COMP_DEBUG"); return e ? atoi(e) : 0; }();
  a.no_hits = dbg & 1;
  if (dbg & 2) a.counts = nullptr;
  rc = launch_coarse_scan<kScanSelect>(a, stream);
  a.counts = counts;
  if (rc != kOk) return rc;
  return select_sublists(keys, counts, rows, p.cap, kSubLists, k_group, out_scores, out_pos, out_counts, out_flag, stream);
}

int component_topk_capacity(const Shape& s, int B, int64_t n, int k_group) {
  CoarseTopkPlan p;
  return coarse_topk_plan(B * s.query_dot_product_groups * s.item_dot_product_groups, n, k_group, &p, true, B * s.query_dot_product_groups) ? p.cap : 0;
}

}  // namespace mol
