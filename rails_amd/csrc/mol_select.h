// Selection fused into the scoring kernels (rails_mol_score_topk): instead of (or next to) writing the (B, N) logits the
// reference materialises before torch.topk (rails/indexing/mol_top_k.py:118-130), every wave compares its logits with a
// running per-query lower bound on the k-th largest score and appends the survivors as 64-bit keys
// (orderable(score) << 32 | ~position, the keys of topk.hip) to a per-query list; one row_select launch then picks the k
// largest keys of each list -- the same keys in the same total order as the dense path, hence the same result bit for bit.
//
//   bound   the k-th largest of ANY subset of a query's scores is <= the k-th largest of all of them, so every value a
//           workgroup publishes (atomicMax on the orderable score) is valid for the rest of the launch, however stale a reader's
//           copy is.  It starts at "none" (everything passes: the first two tiles of every workgroup seed the lists) and is
//           raised at the doubling checkpoints it = 1, 2, 4, ... of the shells' loops by ONE workgroup per (checkpoint, query):
//           the k-th largest of the four largest keys of every thread over the keys appended so far (4 bits per step, LDS
//           counters -- the scheme of row_select_kernel's fast path).  Between checkpoints i and 2i about k (1 + 1/4) keys pass
//           per query; the other workgroups pick the new bounds up on the two iterations after a checkpoint.
//   list    a query's list is kSelSegs segments of kSelSegCap keys, one segment per workgroup: a workgroup counts its own
//           survivors in LDS, so appending costs one LDS atomic per wave and query and one plain global store per survivor.
//           (One global counter per query was measured first: a wave that waits ~1.5 us for its atomicAdd to return in a fifth
//           of its units cost the fp32 kernel 3 % and the f16x3 kernel 11 %.)  Empty slots are 0 -- no valid key is -- so neither
//           the checkpoints nor the final selection need counts; the selection zeroes the slots it consumed.
//   overflow a workgroup with more than kSelSegCap survivors for one query (adversarial orders: scores ascending in position)
//           raises the status word; the caller's dense pass, enqueued behind the selection under that word as launch
//           predicate, takes over.
#pragma once
#include <hip/hip_runtime.h>

#include "mol_kernels.h"

namespace mol {

constexpr int kSelMaxB = 256;      // per-workgroup LDS copies of the bounds and counters
constexpr int kSelMaxK = 384;      // k-th largest of the 4 x 512 per-thread maxima of a checkpoint
constexpr int kSelSegs = 256;      // workgroups of a fused launch (its grid is capped at this)
constexpr int kSelSegCap = 128;    // survivors per (query, workgroup)
constexpr int kSelCap = kSelSegs * kSelSegCap;   // keys per query: 32 768
constexpr int64_t kSelMinItems = 131072;         // below ~16 rounds of tiles the seeding rounds are most of the corpus: the dense path is as fast

__device__ __forceinline__ unsigned int sel_orderable(float f) {   // == orderable() of topk.hip
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sel_unorderable(unsigned int k) {   // == unorderable() of topk.hip
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
// The float a wave compares its logits with: hit = !(logit < bound).  A superset of the key comparison (NaN logits pass, -0 and +0
// compare equal), so nothing the selection needs is ever dropped; extra survivors are harmless.
__device__ __forceinline__ float sel_bound_of(unsigned int key) { return key ? sel_unorderable(key) : -INFINITY; }

struct SelNone {};   // placeholder of the dense instantiations

// LDS state of a workgroup
struct SelLds {
  float thr[kSelMaxB + 4];           // the bounds as floats (-inf: none yet); padded so that a group's four can be read past B
  unsigned int cnt[kSelMaxB];        // this workgroup's survivors per query so far = next free slot of its segment
  unsigned int ctr[3][16];           // counters of the checkpoint selection
  int fresh;                         // 1 during iteration 1: waves read the bounds from memory (the first checkpoint is being published)
  float stash[kScoreWaves][32][4];   // a wave's logits of the current unit: [item of the tile][query of the group]
};

__device__ __forceinline__ void sel_init(const ScoreArgs& p, SelLds& s) {   // before the first barrier of the kernel
  if (!p.sel_list) return;
  for (int i = threadIdx.x; i < p.B + 4; i += blockDim.x) s.thr[i] = -INFINITY;
  for (int i = threadIdx.x; i < p.B; i += blockDim.x) s.cnt[i] = 0u;
  if (threadIdx.x == 0) s.fresh = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *p.sel_status = 0;
}

// Append the survivors of ONE query held by the lanes of `m` (wave-uniform q): slot reservation in the workgroup's own segment.
__device__ __forceinline__ void sel_append(const ScoreArgs& p, SelLds& s, int q, unsigned long long m, bool mine, float out, int64_t item) {
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  unsigned int base = 0;
  if (lane == leader) base = atomicAdd(&s.cnt[q], (unsigned int)__popcll(m));
  base = (unsigned int)__shfl((int)base, leader, 64);
  const unsigned int at = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
  if (mine) {
    if (at < (unsigned int)kSelSegCap)
      p.sel_list[((int64_t)q * kSelSegs + blockIdx.x) * kSelSegCap + at] = ((unsigned long long)sel_orderable(out) << 32) | (unsigned int)(~(unsigned int)item);
    else
      *p.sel_status = 1;
  }
}

// The register-resident units hand over one query at a time, in the middle of their tightest stretch (accumulators of two
// GEMMs live), and every VALU instruction there is paid in full (fp32 MFMA and VALU share the issue port).  Per query the wave
// parks its 32 values in its LDS stash (one ds_write with an immediate offset, nothing waits for it), compares them with the
// query's bound and ORs the verdict into a wave-uniform mask; after the unit's last query ONE scalar branch decides whether
// anything has to be appended.  (First version: the unit-end code read the stash and the bounds back and compared there -- an
// LDS round trip per unit that showed as + 0.5 % on the fp32 kernel.)
struct SelUnit { unsigned long long hit = 0ull; };

template <int QT>
__device__ __forceinline__ void sel_query(const ScoreArgs& p, SelLds& s, SelUnit& u, int g, int Q, int x, int64_t item, float out, bool lane_holds) {
  static_assert(QT == 4, "the stash is laid out for four queries per group (P_Q = 8)");
  const int q = g * QT + Q;
  if (p.logits != nullptr) {   // the dense logits as well (the verified modes read them): what the dense instantiation does
    if (lane_holds && q < p.B && item < p.n_items) p.logits[(int64_t)q * p.ld + item] = out;
  }
  if (lane_holds) s.stash[threadIdx.x >> 6][x][Q] = out;
#ifndef RAILS_SEL_NOHIT   // (timing experiment: never append)
  u.hit |= __ballot(lane_holds && !(out < s.thr[q]));   // rows past the batch end / items past the corpus end are sorted out in sel_flush
#endif
}

template <int QT>
__device__ __forceinline__ void sel_flush(const ScoreArgs& p, SelLds& s, const SelUnit& u, int g, int64_t item0) {
  if (u.hit == 0ull) return;
  // rare from the second checkpoint on: a handful of lanes hold survivors.  All 64 lanes: lane = (item x, queries 2 * half and
  // 2 * half + 1 of the group); one reservation per query of the group.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = lane & 31, half = lane >> 5;
  const float2 v = *reinterpret_cast<const float2*>(&s.stash[wave][x][2 * half]);
  float2 t = *reinterpret_cast<const float2*>(&s.thr[g * QT + 2 * half]);
  const int64_t item = item0 + x;
  const int q0 = g * QT + 2 * half;
  if (s.fresh) {   // iteration 1 only: the first bounds are being published while this unit ran -- one round trip instead of a second seeding round
    t.x = q0 < p.B ? sel_bound_of(__hip_atomic_load(&p.sel_thr[q0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : INFINITY;
    t.y = q0 + 1 < p.B ? sel_bound_of(__hip_atomic_load(&p.sel_thr[q0 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : INFINITY;
  }
  const bool h0 = !(v.x < t.x), h1 = !(v.y < t.y);
  const bool ok = item < p.n_items;
#pragma unroll 1
  for (int Q = 0; Q < QT; ++Q) {
    const int q = g * QT + Q;
    const bool mine = (Q >> 1) == half && ok && q < p.B && ((Q & 1) ? h1 : h0);
    const unsigned long long m = __ballot(mine);
    if (m != 0ull) sel_append(p, s, q, m, mine, (Q & 1) ? v.y : v.x, item);
  }
}

// Raise query q's bound from the keys appended so far.  Called by ALL NT threads of a workgroup (barriers inside).
template <int NT>
__device__ __forceinline__ void sel_tighten(const ScoreArgs& p, SelLds& s, int q) {
  const int tid = threadIdx.x, lane = tid & 63;
  // upper words of the query's kSelCap slots: device-coherent loads (the writers are other CUs); an empty slot is 0.  Two threads
  // share one workgroup's segment, even and odd slots, and each keeps its FOUR largest: 4 NT distinct elements of the row, so the
  // k-th largest of them bounds the row's from below.  (Per segment, not strided across segments: a segment's late slots hold the
  // survivors of real bounds -- the large keys -- and a thread that strides over the same slot position of many segments sees
  // either none of them or dozens, of which it keeps a few: measured bound 6 x looser.)
  static_assert(NT * 64 == kSelCap, "two threads per segment");
  const unsigned int* hi32 = reinterpret_cast<const unsigned int*>(p.sel_list + (int64_t)q * kSelCap + (int64_t)(tid >> 1) * kSelSegCap + (tid & 1)) + 1;
  unsigned int t0 = 0u, t1 = 0u, t2 = 0u, t3 = 0u;
#pragma unroll 32   // cold code at the loop top, where the register file is free: two batches of 32 loads in flight
  for (int i = 0; i < kSelSegCap / 2; ++i) {
    unsigned int v = __hip_atomic_load(hi32 + 4 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int lo;
    lo = v < t0 ? v : t0; t0 = v > t0 ? v : t0; v = lo;
    lo = v < t1 ? v : t1; t1 = v > t1 ? v : t1; v = lo;
    lo = v < t2 ? v : t2; t2 = v > t2 ? v : t2; v = lo;
    t3 = v > t3 ? v : t3;
  }
  if (tid < 48) (&s.ctr[0][0])[tid] = 0u;
  __syncthreads();
  // L = the largest value with count(t >= L) >= k, four bits per step (the count is monotone in the digit)
  unsigned int L = 0u;
  int it = 0;
#pragma unroll 1
  for (int shift = 28; shift >= 0; shift -= 4) {
    unsigned int mine = 0u;
#pragma unroll 1   // cold code inside register-tight kernels: rolled, so that it holds one ballot mask at a time, not thirty
    for (int d = 1; d < 16; ++d) {
      const unsigned int x = L | ((unsigned int)d << shift);
      const unsigned int cd = (unsigned int)(__popcll(__ballot(t0 >= x)) + __popcll(__ballot(t1 >= x)) + __popcll(__ballot(t2 >= x)) + __popcll(__ballot(t3 >= x)));
      mine = lane == d ? cd : mine;
    }
    const int buf = it % 3;
    if (lane >= 1 && lane < 16 && mine) atomicAdd(&s.ctr[buf][lane], mine);
    if (tid < 16) s.ctr[(it + 1) % 3][tid] = 0u;
    __syncthreads();
    const unsigned int tot = lane < 16 ? s.ctr[buf][lane] : 0u;
    const unsigned long long ok = __ballot(lane >= 1 && lane < 16 && tot >= (unsigned int)p.sel_k);
    L |= (unsigned int)__popcll(ok) << shift;
    ++it;
  }
  if (tid == 0 && L) atomicMax(&p.sel_thr[q], L);
}

// Top of a shell's loop iteration `it` (0-based, the same for every thread of the workgroup), right after its barrier and BEFORE
// the shell requests anything from memory: on the two iterations after a checkpoint the workgroup refreshes its copy of the bounds
// (a device-coherent load the issuing waves wait for; vector-memory results return in order, so nothing else should be in flight).
// The refresh is not fenced from the waves' reads: they see the old or the new value, both valid.
template <int NT>
__device__ __forceinline__ void sel_refresh(const ScoreArgs& p, SelLds& s, int64_t it) {
  if (it <= 2 && threadIdx.x == 0) s.fresh = it == 1;
  if (it < 2) return;
  const int64_t a = it - 1, b = it - 2;
  if (!((a & (a - 1)) == 0 || (b > 0 && (b & (b - 1)) == 0))) return;
  for (int i = threadIdx.x; i < p.B; i += NT) s.thr[i] = sel_bound_of(__hip_atomic_load(&p.sel_thr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// The checkpoint work of this workgroup at iteration `it` (a power of two): pair (checkpoint c, query q) belongs to workgroup
// (c * B + q) % n_wg.  `wg` / `n_wg`: logical workgroup id and count.  Barriers inside: every thread of the workgroup calls it.
template <int NT>
__device__ __forceinline__ void sel_checkpoint(const ScoreArgs& p, SelLds& s, int64_t it, int wg, int n_wg) {
  if (!p.sel_list) return;
  if (it >= 1 && (it & (it - 1)) == 0) {
    const int c = 63 - __builtin_clzll((unsigned long long)it);   // checkpoint index
    const int first = (int)(((int64_t)wg - ((int64_t)c * p.B) % n_wg + n_wg) % n_wg);
#ifdef RAILS_SEL_PHASES   // debug build: duration of workgroup 0's first checkpoint selection in 10 ns ticks -> the word after the status
    const long long t0 = (long long)wall_clock64();
#endif
#ifndef RAILS_SEL_NOTIGHT   // (timing experiment: no checkpoint work)
    for (int q = first; q < p.B; q += n_wg) sel_tighten<NT>(p, s, q);
#endif
#ifdef RAILS_SEL_PHASES
    if (wg == 0 && it == 1 && threadIdx.x == 0) p.sel_status[1] = (int)((long long)wall_clock64() - t0);
#endif
  }
}

}  // namespace mol
