// Exact top-k selection and the seen-id filter.
//
// Replaces torch.topk(all_logits, dim=1, k, sorted=True, largest=True) and the id gather of
// MoLBruteForceTopK.forward (rails/indexing/mol_top_k.py:123-130), and the row-wise masking of
// CandidateIndex.get_top_k_outputs (indexing/candidate_index.py:154-178).
//
// Every score becomes a 64-bit key  (orderable(score) << 32) | ~position : all keys of a row are
// distinct, so "the k largest keys" is a unique set and its descending order is "score descending,
// then position ascending" -- the deterministic tie rule that makes 1/2/4/8-GPU results identical.
//
//   n <= 1024   one workgroup per row bitonic-sorts the whole row in LDS (also n <= 16384 with k > 4096).
//   n <= 49152  (k <= 4096) row_select_kernel: one workgroup per row holds the row in registers (<= 48 scores per thread).
//               k <= 512: a lower bound on the k-th score from the per-thread maxima (4 bits per step, per-lane LDS
//               counters), compaction of the ~220 survivors, counting-rank emit.  Otherwise / on overflow (heavy ties):
//               radix bisection of the k-th key, 2 bits per step, over the score bits and -- for a tied k-th score -- on
//               over the position bits.  One launch.
//   n <= ~4.5 M (k <= 512) two launches: row_select_kernel per (row, chunk <= 49152) writes each chunk's k winners as keys,
//               a second row_select_kernel selects among the chunks * k <= 24576 keys.  Reads the scores once.
//   otherwise   MSB-first radix select over the 32 score bits (11/11/10, LDS histograms) finds the k-th largest
//               score; if that score is tied, one in-order scan of the row finds the position of the last tied
//               element to take, which completes the 64-bit threshold key T; one compaction pass gathers the
//               exactly-k keys >= T; the LDS sort orders them.  Reads the scores five times.
// (Measured and dropped: LDS-sorting 16K-element chunks and merging their top-k -- a 16K-key bitonic sort costs ~150 us,
//  more than the whole radix path; and folding the bin pick into the histogram kernel with arrival tickets + fences.)
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "mol_kernels.h"

namespace mol {

constexpr int kSortCap = 16384;       // 64-bit keys in 128 KiB of LDS
constexpr int kSortThreads = 1024;
constexpr int kHistThreads = 512;
constexpr int kRadixPasses = 3;
constexpr int kBins = 2048;
__device__ __constant__ int kPassShift[kRadixPasses] = {53, 42, 32};
__device__ __constant__ int kPassBits[kRadixPasses] = {11, 11, 10};

struct SelectState {       // one per row, lives in the workspace
  unsigned long long prefix;  // resolved high bits of the k-th key (low bits zero)
  unsigned int need;          // how many keys with that prefix are still wanted
  unsigned int done;          // 1: every key >= prefix is selected, threshold final
  unsigned int count;         // compaction cursor
  unsigned int pad;
};

__device__ __forceinline__ unsigned int orderable(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorderable(unsigned int k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
// The id written for column `pos` of row `row`: ids[ids_row_stride * row + pos], the position itself without an id table, or --
// with ids_index (the rows are candidate lists, ids_index[row][pos] the candidate's position in the corpus) -- ids[ids_index[row][pos]]
// resp. that corpus position (torch.gather + the id lookup of rails/indexing/mol_top_k.py:379-382 inside the selection's launch).
__device__ __forceinline__ int64_t map_id(const int64_t* __restrict__ ids, int64_t ids_row_stride, const int64_t* __restrict__ ids_index,
                                          int64_t ids_index_ld, int row, unsigned int pos) {
  if (ids_index) {
    const int64_t at = ids_index[(int64_t)row * ids_index_ld + pos];
    return ids ? ids[ids_row_stride * row + at] : at;
  }
  return ids ? ids[ids_row_stride * row + pos] : (int64_t)pos;
}
__device__ __forceinline__ unsigned long long make_key(float score, unsigned int pos) {
  return ((unsigned long long)orderable(score) << 32) | (unsigned int)(~pos);
}

// ---- block bitonic sort, one key per thread (npad <= 1024) ----------------------------------------------------
// Thread i holds key i.  Compare-exchange partners at distance < 64 sit in the same wavefront and are exchanged with
// ds_bpermute (no barrier); only the distances >= 64 go through LDS (one barrier each, double-buffered): 3 barriers for
// 256 keys, 10 for 1024, against 36 / 55 barrier-separated LDS passes for the plain loop (~0.4 us each with 16 waves).
// Returns the key of descending rank threadIdx.x.  `buf` needs 2 * npad entries; all threads of the block must call.
__device__ __forceinline__ unsigned long long block_sort_desc(unsigned long long key, int npad, unsigned long long* buf) {
  const int i = threadIdx.x;
  int flip = 0;
  for (int size = 2; size <= npad; size <<= 1) {
    const bool desc = (i & size) == 0;
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      unsigned long long other;
      if (stride >= 64) {
        unsigned long long* b = buf + flip * npad;
        flip ^= 1;
        if (i < npad) b[i] = key;
        __syncthreads();
        other = i < npad ? b[i ^ stride] : 0ull;
      } else {
        other = __shfl_xor(key, stride, 64);
      }
      const bool lower = (i & stride) == 0;
      const bool take_max = lower == desc;
      const unsigned long long mx = key > other ? key : other, mn = key > other ? other : key;
      key = take_max ? mx : mn;
    }
  }
  return key;
}

// ---- more keys than threads: KPT = npad / 1024 keys per thread (npad = 2048 .. 16384) --------------------------------
// Thread t holds the keys of LDS slots [t * KPT, (t + 1) * KPT).  Compare-exchange distances below KPT stay inside the
// thread, distances below 64 * KPT are one ds_bpermute per key inside the wavefront, only the rest go through LDS (two
// barriers each, `keys` itself is the exchange buffer): 10 LDS steps of 78 for 4096 keys, where the plain loop took a
// barrier-separated LDS pass for every step (39 us per 4096-key row).
// In: keys[0, npad) in LDS, visible to all threads.  Out: the same, sorted descending, visible to all threads.
template <int KPT>
__device__ __forceinline__ void block_sort_desc_multi(unsigned long long* keys) {
  constexpr int npad = KPT * kSortThreads;
  const int t = threadIdx.x;
  unsigned long long key[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) key[j] = keys[t * KPT + j];
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride >= KPT; stride >>= 1) {
      unsigned long long other[KPT];
      if (stride >= 64 * KPT) {
        __syncthreads();                       // every thread is done reading the previous exchange
#pragma unroll
        for (int j = 0; j < KPT; ++j) keys[t * KPT + j] = key[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j) other[j] = keys[(t * KPT + j) ^ stride];
      } else {
#pragma unroll
        for (int j = 0; j < KPT; ++j) other[j] = __shfl_xor(key[j], stride / KPT, 64);
      }
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int e = t * KPT + j;
        const bool take_max = ((e & stride) == 0) == ((e & size) == 0);
        const unsigned long long mx = key[j] > other[j] ? key[j] : other[j], mn = key[j] > other[j] ? other[j] : key[j];
        key[j] = take_max ? mx : mn;
      }
    }
#pragma unroll
    for (int s = KPT / 2; s > 0; s >>= 1) {
      if (s < size) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
          if ((j & s) == 0) {
            const bool desc = (((t * KPT + j) & size) == 0);
            const unsigned long long a = key[j], b = key[j | s];
            const bool sw = (a < b) == desc;
            key[j] = sw ? b : a;
            key[j | s] = sw ? a : b;
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KPT; ++j) keys[t * KPT + j] = key[j];
  __syncthreads();
}

// ---- block rank (64 <= npad <= 512 keys in LDS, zero-padded, distinct) ------------------------------------------
// A key's descending rank is the number of larger keys.  The 1024 threads split into 1024/npad parts; thread (c, part)
// counts the keys of its slice that exceed key c (broadcast 16-byte LDS reads, no conflicts) and adds the count to
// rank_buf[c].  Two barriers and ~npad^2/2048 LDS reads per thread, against the 36-45 dependent shuffle steps of the
// bitonic network (5.6 us -> ~1.5 us for the ~220 candidates of a k = 200 selection).  Afterwards every thread with
// part 0 holds (key c, rank of key c).
__device__ __forceinline__ void block_rank_desc(const unsigned long long* keys, int npad, unsigned int* rank_buf,
                                                unsigned long long& mine, unsigned int& rank, bool& owner) {
  const int tid = threadIdx.x;
  const int c = tid & (npad - 1), part = tid / npad;
  const int span = npad / ((int)blockDim.x / npad);      // keys per slice (>= 4 for npad >= 64 at 1024 threads)
  if (tid < npad) rank_buf[tid] = 0u;
  __syncthreads();
  mine = keys[c];
  unsigned int cnt = 0;
  const ulonglong2* p = reinterpret_cast<const ulonglong2*>(keys + part * span);
  for (int i = 0; i < span / 2; ++i) {
    const ulonglong2 x = p[i];
    cnt += x.x > mine ? 1u : 0u;
    cnt += x.y > mine ? 1u : 0u;
  }
  if (cnt) atomicAdd(&rank_buf[c], cnt);
  __syncthreads();
  rank = rank_buf[c];
  owner = part == 0;
}

// ---- LDS bitonic sort (descending) + emit ------------------------------------------------------
// Input: keys from cand[row*cand_ld + i], i < count (cand != NULL), else from scores[row*ld + begin + i] with positions
// begin + i, where [begin, begin + count) is this workgroup's chunk of the row (blockIdx.y * chunk ...).
// Output: the k_out largest keys, descending -- decoded to (score, id) when out_scores != NULL, else raw keys to
// keys_out[row*keys_ld + blockIdx.y*k_out + j] (first level of the chunked path).
__global__ __launch_bounds__(kSortThreads) void sort_emit_kernel(const float* __restrict__ scores, int64_t ld,
                                                                int64_t n, int64_t chunk,
                                                                const unsigned long long* __restrict__ cand,
                                                                int64_t cand_ld, int cand_count, int k_out, int npad,
                                                                const int64_t* __restrict__ ids, int64_t ids_row_stride,
                                                                float* __restrict__ out_scores,
                                                                int64_t* __restrict__ out_ids,
                                                                unsigned long long* __restrict__ keys_out, int64_t keys_ld, const int32_t* __restrict__ run_if,
                                                                const SelectState* __restrict__ count_from = nullptr,
                                                                const int64_t* __restrict__ ids_index = nullptr, int64_t ids_index_ld = 0) {
  MOL_RUN_IF(run_if);
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const int row = blockIdx.x;
  const int64_t begin = (int64_t)blockIdx.y * chunk;
  // count_from: the radix path's compaction may have kept MORE than k keys (a threshold settled early at a bin edge): the row's
  // cursor says how many, at most cand_count = npad
  const int count = cand ? (count_from ? (int)(count_from[row].count < (unsigned int)cand_count ? count_from[row].count : (unsigned int)cand_count) : cand_count)
                         : (int)((begin + chunk < n ? begin + chunk : n) - begin);
  const int k = k_out;
  for (int i = threadIdx.x; i < npad; i += kSortThreads) {
    unsigned long long kv = 0ull;  // below every real key (orderable() never returns 0 for a finite/inf score)
    if (i < count) kv = cand ? cand[row * cand_ld + i] : make_key(scores[row * ld + begin + i], (unsigned int)(begin + i));
    keys[i] = kv;
  }
  __syncthreads();
  if (npad >= 64 && npad <= 512 && out_scores) {   // counting rank, each key written straight to its slot
    unsigned long long kv; unsigned int rank; bool owner;
    block_rank_desc(keys, npad, reinterpret_cast<unsigned int*>(keys + npad), kv, rank, owner);
    if (owner && kv != 0ull && rank < (unsigned int)k) {
      const unsigned int pos = ~(unsigned int)(kv & 0xFFFFFFFFull);
      out_scores[(int64_t)row * k + rank] = unorderable((unsigned int)(kv >> 32));
      out_ids[(int64_t)row * k + rank] = map_id(ids, ids_row_stride, ids_index, ids_index_ld, row, pos);
    }
    return;
  }
  if (npad <= kSortThreads) {   // one key per thread, sorted mostly in registers; keys[npad, 3 npad) is the exchange buffer
    unsigned long long kv = (int)threadIdx.x < npad ? keys[threadIdx.x] : 0ull;
    kv = block_sort_desc(kv, npad, keys + npad);
    __syncthreads();
    if ((int)threadIdx.x < npad) keys[threadIdx.x] = kv;
    __syncthreads();
  } else if (npad == 2 * kSortThreads) block_sort_desc_multi<2>(keys);
  else if (npad == 4 * kSortThreads) block_sort_desc_multi<4>(keys);
  else if (npad == 8 * kSortThreads) block_sort_desc_multi<8>(keys);
  else if (npad == 16 * kSortThreads) block_sort_desc_multi<16>(keys);
  else
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (npad >> 1); t += kSortThreads) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
        const int hi2 = lo | stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = keys[lo], b = keys[hi2];
        if ((a < b) == desc) { keys[lo] = b; keys[hi2] = a; }
      }
      __syncthreads();
    }
  }
  for (int j = threadIdx.x; j < k; j += kSortThreads) {
    const unsigned long long kv = keys[j];   // j >= count: padding key 0, sorts below everything at the next level
    if (out_scores) {
      const unsigned int pos = ~(unsigned int)(kv & 0xFFFFFFFFull);
      out_scores[(int64_t)row * k + j] = unorderable((unsigned int)(kv >> 32));
      out_ids[(int64_t)row * k + j] = map_id(ids, ids_row_stride, ids_index, ids_index_ld, row, pos);
    } else {
      keys_out[row * keys_ld + (int64_t)blockIdx.y * k + j] = kv;
    }
  }
}

// Walk [begin, end) of a row with 16-byte loads: scalar head up to the first 16-B aligned element, float4 body, scalar
// tail.  `f(score, position)` is called once per element (order unspecified).
template <class F>
__device__ __forceinline__ void for_each_in_chunk(const float* __restrict__ rowp, int64_t begin, int64_t end, int nthreads, F f) {
  int64_t a0 = begin + ((4 - (int64_t)((reinterpret_cast<uintptr_t>(rowp + begin) >> 2) & 3)) & 3);
  if (a0 > end) a0 = end;
  const int64_t nvec = (end - a0) >> 2;
  for (int64_t i = begin + threadIdx.x; i < a0; i += nthreads) f(rowp[i], i);
  const float4* body = reinterpret_cast<const float4*>(rowp + a0);
  for (int64_t v = threadIdx.x; v < nvec; v += nthreads) {
    const float4 x = body[v];
    const int64_t i = a0 + 4 * v;
    f(x.x, i); f(x.y, i + 1); f(x.z, i + 2); f(x.w, i + 3);
  }
  for (int64_t i = a0 + 4 * nvec + threadIdx.x; i < end; i += nthreads) f(rowp[i], i);
}

// ---- radix select ------------------------------------------------------------------------------
// Workspace state is zeroed by one memset per call: SelectState (prefix 0, need 0 = "k", done 0, count 0) and the
// per-pass histograms.  Each pass is a histogram launch plus a one-workgroup-per-row pick launch.  (Folding the pick into
// the histogram kernel through an arrival ticket + agent-scope release/acquire was measured: the 2048 release fences
// cost 35-100 us per pass, against 8 us for the separate launch.)

// one wave per row: walk the histogram from the top bin down to the bin holding the need-th key.
// Lane l owns the nb/64 bins [top - l*per - per + 1, top - l*per]; a wave prefix sum finds the owning lane, which then
// walks its own bins.
__global__ __launch_bounds__(64) void pick_bin_kernel(SelectState* __restrict__ st, const unsigned int* __restrict__ hist,
                                                      int pass, int rows, int k, const int32_t* __restrict__ run_if, int cap) {
  MOL_RUN_IF(run_if);
  const int row = blockIdx.x;
  if (st[row].done) return;
  const int shift = kPassShift[pass], bits = kPassBits[pass];
  const int nb = 1 << bits, per = nb / 64;  // 32 or 16 bins per lane, top bins first
  const unsigned int* gh = hist + ((int64_t)pass * rows + row) * kBins;
  const unsigned int need = pass == 0 ? (unsigned int)k : st[row].need;
  const int lane = threadIdx.x;
  const int top = nb - 1 - lane * per;
  unsigned int mine = 0;
  for (int j = 0; j < per; ++j) mine += gh[top - j];
  unsigned int incl = mine;  // inclusive prefix over lanes (lane 0 = top bins)
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const unsigned int excl = incl - mine;
  if (excl < need && need <= incl) {   // exactly one lane (need <= total by construction: k <= n)
    unsigned int cum = excl;
    int bin = top;
    for (int j = 0; j < per; ++j, --bin) { const unsigned int c = gh[bin]; if (cum + c >= need) break; cum += c; }
    const unsigned int c = gh[bin];
    const unsigned int still = need - cum;
    st[row].prefix |= ((unsigned long long)(unsigned int)bin) << shift;
    st[row].need = still;
    // all keys of this bin are wanted -> the threshold is the bin's lower edge; nothing left to resolve.
    // Otherwise the next pass narrows it, and after the last pass tie_resolve_kernel finishes it.
    // Also final when everything at or above this bin's lower edge fits the sort's `cap` slots: the compaction keeps those
    // (k - still) + c >= k keys and the sort cuts them to k -- the remaining passes and the tie scan exit at once.  (With
    // 22 bits resolved after the second pass a bin of amzn-books-sized logit rows holds a handful of keys.)
    if (c == still || (unsigned int)k - still + c <= (unsigned int)cap) st[row].done = 1u;
  }
}

__global__ __launch_bounds__(kHistThreads) void hist_kernel(const float* __restrict__ scores, int64_t ld, int64_t n,
                                                           const SelectState* __restrict__ st,
                                                           unsigned int* __restrict__ hist, int pass, int64_t chunk, const int32_t* __restrict__ run_if) {
  MOL_RUN_IF(run_if);
  __shared__ unsigned int h[kBins];
  const int row = blockIdx.y, rows = gridDim.y;
  if (st[row].done) return;
  const int shift = kPassShift[pass], bits = kPassBits[pass];
  const unsigned long long prefix = st[row].prefix;
  const int above = shift + bits;  // bits [above, 64) are resolved
  for (int i = threadIdx.x; i < kBins; i += kHistThreads) h[i] = 0u;
  __syncthreads();
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = (begin + chunk < n) ? begin + chunk : n;
  const float* rowp = scores + (int64_t)row * ld;
  const unsigned int mask = (1u << bits) - 1u;
  // (Measured and not kept: peeling the wave's most common bins into one LDS atomic each in the first pass, where logit rows
  // share a handful of bins: 32 -> 44 us.)
  for_each_in_chunk(rowp, begin, end, kHistThreads, [&](float sc, int64_t i) {
    const unsigned long long key = make_key(sc, (unsigned int)i);
    if ((above >= 64) || ((key >> above) == (prefix >> above))) atomicAdd(&h[(unsigned int)(key >> shift) & mask], 1u);
  });
  __syncthreads();
  unsigned int* gh = hist + ((int64_t)pass * rows + row) * kBins;
  for (int i = threadIdx.x; i < kBins; i += kHistThreads)
    if (h[i]) atomicAdd(&gh[i], h[i]);
}

// The k-th score is tied: `need` of the elements whose score equals it are wanted, lowest positions first.
// One workgroup per row scans the row in position order and finds the position of the need-th such element;
// that completes the threshold key (score bits | ~position).  Rows whose threshold is already final exit at once.
constexpr int kTieThreads = 1024;
__global__ __launch_bounds__(kTieThreads) void tie_resolve_kernel(const float* __restrict__ scores, int64_t ld, int64_t n,
                                                                 SelectState* __restrict__ st, const int32_t* __restrict__ run_if) {
  MOL_RUN_IF(run_if);
  __shared__ unsigned int wave_cnt[kTieThreads / 64];
  const int row = blockIdx.x;
  if (st[row].done) return;
  const unsigned int target = (unsigned int)(st[row].prefix >> 32);
  const unsigned int need = st[row].need;
  const float* rowp = scores + (int64_t)row * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int running = 0;
  for (int64_t base = 0; base < n; base += kTieThreads) {
    const int64_t i = base + threadIdx.x;
    const bool match = i < n && orderable(rowp[i]) == target;
    const unsigned long long bal = __ballot(match);
    if (lane == 0) wave_cnt[wave] = (unsigned int)__popcll(bal);
    __syncthreads();
    unsigned int before = running, total = 0;
    for (int w = 0; w < kTieThreads / 64; ++w) {
      if (w < wave) before += wave_cnt[w];
      total += wave_cnt[w];
    }
    const unsigned int rank = before + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull));  // 0-based among matches
    if (match && rank == need - 1) {
      st[row].prefix = ((unsigned long long)target << 32) | (unsigned int)(~(unsigned int)i);
      st[row].done = 1u;
    }
    running += total;
    __syncthreads();
    if (running >= need) break;
  }
}

__global__ __launch_bounds__(kHistThreads) void compact_kernel(const float* __restrict__ scores, int64_t ld, int64_t n,
                                                              SelectState* __restrict__ st,
                                                              unsigned long long* __restrict__ cand, int64_t cand_ld,
                                                              int k, int64_t chunk, const int32_t* __restrict__ run_if) {
  MOL_RUN_IF(run_if);
  const int row = blockIdx.y;
  const unsigned long long thr = st[row].prefix;
  const int lane = threadIdx.x & 63;
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = (begin + chunk < n) ? begin + chunk : n;
  const float* rowp = scores + (int64_t)row * ld;
  // One GLOBAL atomic per workgroup.  The selected keys of the chunk are staged in LDS (one LDS atomic per wave that holds one),
  // their range of the row's list is reserved with a single atomicAdd on the row cursor, and the staged keys are copied out.
  // A chunk with more selected keys than the stage holds (all of a row's winners in one chunk) is walked a second time and
  // placed directly (the second walk hits L2).  (One global atomic per WAVE that held a selected key was 222 us at k = 2561,
  // n = 695 762, 32 rows: ~2 500 dependent same-address atomics per row, each a round trip to L2; this is 64 per row.)
  constexpr int kStage = 2048;
  __shared__ unsigned long long stage[kStage];
  __shared__ unsigned int wg_base, wg_cursor;
  if (threadIdx.x == 0) wg_cursor = 0u;
  __syncthreads();
  for_each_in_chunk(rowp, begin, end, kHistThreads, [&](float sc, int64_t i) {
    const unsigned long long key = make_key(sc, (unsigned int)i);
    const bool sel = key >= thr;
    const unsigned long long m = __ballot(sel);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      unsigned int base = 0;
      if (lane == leader) base = atomicAdd(&wg_cursor, (unsigned int)__popcll(m));
      base = (unsigned int)__shfl((int)base, leader, 64);
      if (sel) {
        const unsigned int slot = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        if (slot < (unsigned int)kStage) stage[slot] = key;
      }
    }
  });
  __syncthreads();
  const unsigned int total = wg_cursor;
  if (total == 0u) return;   // uniform: every thread read the same value after the barrier above
  __syncthreads();           // every thread has read `total` before thread 0 resets the cursor for the overflow re-walk
  if (threadIdx.x == 0) { wg_base = atomicAdd(&st[row].count, total); wg_cursor = 0u; }
  __syncthreads();
  const unsigned int wbase = wg_base;
  if (total <= (unsigned int)kStage) {
    for (unsigned int j = threadIdx.x; j < total; j += kHistThreads)
      if (wbase + j < (unsigned int)cand_ld) cand[row * cand_ld + wbase + j] = stage[j];
    return;
  }
  for_each_in_chunk(rowp, begin, end, kHistThreads, [&](float sc, int64_t i) {
    const unsigned long long key = make_key(sc, (unsigned int)i);
    const bool sel = key >= thr;
    const unsigned long long m = __ballot(sel);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      unsigned int base = 0;
      if (lane == leader) base = atomicAdd(&wg_cursor, (unsigned int)__popcll(m));
      base = (unsigned int)__shfl((int)base, leader, 64);
      if (sel) {
        const unsigned int slot = wbase + base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        if (slot < (unsigned int)cand_ld) cand[row * cand_ld + slot] = key;
      }
    }
  });
}

// ---- register-resident select: one workgroup per (row, chunk of <= 49152 elements) ---------------------------------
// A workgroup holds its elements in registers (<= 48 per thread) and selects the k largest 64-bit keys.
//   source  SCORES: a chunk of a score row, loaded as float4 (thread t owns elements (jv*1024 + t)*4 .. +3)
//           KEYS:   a list of 64-bit keys (the per-chunk winners of a first level)
//   output  final (score, id) pairs, or the k sorted keys into the workspace (first level of the two-level path)
//
// Fast path (k <= 512).  The k-th largest of any subset of the row is <= the row's k-th largest, so any
// L <= (k-th largest per-thread maximum) bounds the answer from below and every wanted key is among {v >= L}.  L is
// resolved to its top 16 bits, four bits per step: 15 ballots give the 15 threshold counts, lane c keeps count c, one LDS
// atomic per wave accumulates them in a rotating counter set, one barrier, one LDS read per lane and a ballot pick the
// digit.  Elements are dealt to threads in small runs, so for untied data only about -1024 ln(1 - k/1024) of them pass
// (~220 for k = 200); they are compacted (per-thread counts, one wave scan and one LDS atomic per wave) and sorted as
// 64-bit keys, which settles ties by position.  Measured phases for n = 27278, k = 200 (tools/row_select_phases.hip):
// see DESIGN.md section 3.3.
// General path (k > 512, heavy ties, adversarial layouts that overflow the candidate buffer): radix bisection of the
// k-th largest key, two bits per step, first over the 32 score bits and -- only if the k-th score is tied -- on over the
// 32 position bits among the tied elements; exactly k keys are then >= the threshold key.
constexpr int kRowThreads = 1024;
constexpr int kRowMaxK = 4096;
constexpr int kRowMaxN = 48 * kRowThreads;
constexpr int kRowFastK = 512;
constexpr int kRowCandCap = 4096;   // candidate keys the fast path may hand to the sort

#ifdef RAILS_TOPK_PHASES   // tools/row_select_phases.hip: wall-clock stamps (100 MHz) of workgroup 0's phases
__device__ long long g_phase[16];
#define RAILS_PHASE(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_phase[i] = (long long)wall_clock64(); } while (0)
#else
#define RAILS_PHASE(i)
#endif

constexpr int kFuseMaxK = 512;    // candidates per row the fused seen-id filter stages in LDS
constexpr int kFuseMaxW = 256;    // seen ids per row

// The seen-id filter of the candidate index (reference indexing/candidate_index.py:149-175) over a row's k' selected candidates
// held in LDS, best first: keep the first k NOT-seen ones in order; if fewer than k exist, back-fill with the earliest dropped
// ones (seen, or beyond the first k) -- exactly what filter_seen_kernel does, one thread per candidate.  All threads of the
// workgroup call it (NT threads, NT >= kp); `scratch` = NT / 64 + 2 ints.
template <int NT>
__device__ __forceinline__ void filter_from_lds(const int64_t* id_s, const float* sc_s, int kp, const int64_t* inv_s, int width, int k,
                                                int64_t* __restrict__ out_ids, float* __restrict__ out_scores, int* scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  auto block_scan = [&](int v, int& total) -> int {   // inclusive scan over the NT threads
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    __syncthreads();
    if (lane == 63) scratch[wv] = inc;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int t = 0; t < NT / 64; ++t) { const int tv = scratch[t]; if (t < wv) base += tv; total += tv; }
    return base + inc;
  };
  int ok = 0;
  int64_t id = 0;
  if (tid < kp) {
    id = id_s[tid];
    bool seen = false;
    for (int w = 0; w < width; ++w) seen |= (inv_s[w] == id);
    ok = seen ? 0 : 1;
  }
  int total_ok;
  const int okc = block_scan(ok, total_ok);                 // not-seen candidates up to and including this one
  const bool valid = ok && okc <= k;
  const int gap = k - (total_ok < k ? total_ok : k);
  int total_bad;
  const int badc = block_scan(tid < kp && !valid ? 1 : 0, total_bad) - (tid < kp && !valid ? 1 : 0);   // dropped candidates before this one
  if (tid < kp) {
    int out_pos = -1;
    if (valid) out_pos = (okc - 1) + (badc < gap ? badc : gap);
    else if (badc + 1 <= gap) out_pos = (okc < k ? okc : k) + badc;
    if (out_pos >= 0 && out_pos < k) {
      out_ids[out_pos] = id;
      out_scores[out_pos] = sc_s[tid];
    }
  }
}

// ---- MSD radix selection of the want-th largest 64-bit key of a workgroup's register-resident keys ---------------------------------
// key_of(j), j < VPT: the thread's keys, 0 = no key (real keys are never 0 and all distinct).  Eight bits per pass: a 256-bin LDS
// histogram of the keys that match the resolved prefix (a wave adds its most common digit with one atomic), one wave turns it into
// the digit.  Bytes in which all keys agree -- bits of OR ^ AND over the keys -- take no pass (bf16 scores: the two low bytes of the
// score word; the top byte of the positions), and the passes stop as soon as every key that still matches is wanted.  On return
// exactly min(want, #keys) keys satisfy (key & fixed) >= prefix.  Two barriers per pass (the two-bits-per-step bisection this
// replaced took 17 steps on the scores + 16 on the positions whenever the k-th score was tied).
// `sh` must be initialised (radix_init by every thread, then a barrier) before the call; all threads of the workgroup call.
struct RadixShared {
  __attribute__((aligned(16))) unsigned int hist[2][256];
  unsigned long long red[2];      // OR, AND over the keys
  unsigned int pick[3];           // digit, keys still wanted among the matching ones, matching keys
  unsigned int count;             // keys
};
template <int NT>
__device__ __forceinline__ void radix_init(RadixShared& sh) {
  for (int i = threadIdx.x; i < 512; i += NT) (&sh.hist[0][0])[i] = 0u;
  if (threadIdx.x == 0) { sh.red[0] = 0ull; sh.red[1] = ~0ull; sh.count = 0u; }
}
// Keys come as two 32-bit words: hi_of(j) (the score word, 0 = no key: orderable() never returns 0) and lo_of(j, z) (the position word;
// z is an opaque zero the callers add into what they compute, so that the compiler does not hoist VPT position words out of the pass
// loop -- with 48 scores per thread that cost 436-968 B of scratch per lane).  Selected afterwards: radix_selected(hi, lo, sel).
struct RadixSel { unsigned int ph, fh, pl, fl; };   // prefix / fixed mask of the score word and of the position word
__device__ __forceinline__ bool radix_selected(unsigned int hi, unsigned int lo, const RadixSel& r) {
  const unsigned int mh = hi & r.fh;
  return hi != 0u && (mh > r.ph || (mh == r.ph && (lo & r.fl) >= r.pl));
}
template <int VPT, class KH, class KL>
__device__ __forceinline__ void radix_select(KH hi_of, KL lo_of, unsigned int want, RadixShared& sh, RadixSel& sel) {
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned int oh = 0u, ah = ~0u, ol = 0u, al = ~0u, mine = 0u;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const unsigned int h = hi_of(j), l = lo_of(j, 0u);
    if (h) { oh |= h; ah &= h; ol |= l; al &= l; ++mine; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    oh |= __shfl_xor(oh, o, 64); ah &= __shfl_xor(ah, o, 64); ol |= __shfl_xor(ol, o, 64); al &= __shfl_xor(al, o, 64);
    mine += __shfl_xor(mine, o, 64);
  }
  if (lane == 0) {
    atomicOr(&sh.red[0], ((unsigned long long)oh << 32) | ol);
    atomicAnd(&sh.red[1], ((unsigned long long)ah << 32) | al);
    atomicAdd(&sh.count, mine);
  }
  __syncthreads();
  const unsigned long long vor = sh.red[0];
  if (want > sh.count) want = sh.count;
  const unsigned long long varying = vor ^ sh.red[1];
  sel = RadixSel{0u, 0u, 0u, 0u};
  if (!want) { sel = RadixSel{~0u, ~0u, ~0u, ~0u}; return; }   // nothing is selected
  unsigned int need = want;
  int pass = 0;
#pragma unroll 1
  for (int byte = 7; byte >= 0; --byte) {   // rolled: one copy of the pass in the callers' code
    const int shift = 8 * (byte & 3);
    const bool score_word = byte >= 4;
    const unsigned int bmask = 0xFFu << shift;
    const unsigned int vary_w = score_word ? (unsigned int)(varying >> 32) : (unsigned int)varying;
    const unsigned int or_w = score_word ? (unsigned int)(vor >> 32) : (unsigned int)vor;
    if ((vary_w & bmask) == 0u) {   // all keys agree on this byte
      if (score_word) { sel.ph |= or_w & bmask; sel.fh |= bmask; } else { sel.pl |= or_w & bmask; sel.fl |= bmask; }
      continue;
    }
    unsigned int* h = sh.hist[pass & 1];
    unsigned int z = 0u;
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const unsigned int kh = hi_of(j), kl = lo_of(j, z);
      const bool act = kh != 0u && (kh & sel.fh) == sel.ph && (kl & sel.fl) == sel.pl;
      const unsigned int digit = ((score_word ? kh : kl) >> shift) & 255u;
      const unsigned long long m = __ballot(act);
      if (m) {   // wave-uniform
        const int leader = __ffsll((long long)m) - 1;
        const unsigned int d0 = (unsigned int)__shfl((int)digit, leader, 64);
        const unsigned long long same = __ballot(act && digit == d0);
        if (lane == leader) atomicAdd(&h[d0], (unsigned int)__popcll(same));
        if (act && digit != d0) atomicAdd(&h[digit], 1u);
      }
    }
    if (tid >= 256 && tid < 512) sh.hist[(pass + 1) & 1][tid - 256] = 0u;   // the next pass' bins (last read before the previous barrier)
    __syncthreads();
    if (tid < 64) {
      const uint4 c4 = reinterpret_cast<const uint4*>(h)[lane];      // bins 4 lane .. 4 lane + 3
      const unsigned int sum4 = c4.x + c4.y + c4.z + c4.w;
      unsigned int incl = sum4;                                       // keys in the bins of lanes >= lane
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_down(incl, o, 64); if (lane + o < 64) incl += t; }
      unsigned int cum = incl - sum4;                                 // keys in higher bins
      const unsigned int c[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
      for (int b = 3; b >= 0; --b) {
        if (cum < need && need <= cum + c[b]) { sh.pick[0] = 4u * lane + b; sh.pick[1] = need - cum; sh.pick[2] = c[b]; }
        cum += c[b];
      }
    }
    __syncthreads();
    if (score_word) { sel.ph |= sh.pick[0] << shift; sel.fh |= bmask; } else { sel.pl |= sh.pick[0] << shift; sel.fl |= bmask; }
    need = sh.pick[1];
    ++pass;
    if (sh.pick[2] == need) break;      // every key that still matches the prefix is wanted
  }
}

struct RowSelectArgs {
  const float* scores; int64_t ld; int64_t n; int64_t chunk;     // SCORES source: row r, chunk c = [c*chunk, min(n, (c+1)*chunk))
  const unsigned short* scores16;                                // SCORES source held as bf16 bit patterns (then `scores` is unused)
  const unsigned long long* keys_in; int keys_per_row;           // KEYS source
  int k; int lds_keys;
  const int64_t* ids; int64_t ids_row_stride;                    // final output (out_scores != NULL)
  const int64_t* ids_index; int64_t ids_index_ld;                // map_id
  float* out_scores; int64_t* out_ids;
  unsigned long long* keys_out;                                  // else: keys_out[(row*gridDim.y + c)*k + j]
  const int32_t* run_if;                                         // launch predicate (mol_kernels.h)
  // fused seen-id filter (final output only; k <= kFuseMaxK, width <= kFuseMaxW): the k selected (score, id) pairs go through the
  // filter of filter_seen_kernel inside this launch and f_k of them are written (out_scores / out_ids then hold f_k per row)
  const int64_t* f_invalid; int f_width; int f_k;
};

template <int VPT, bool KEYS, bool IDX = false>   // IDX: ids through ids_index (map_id); built for VPT = 4 / 8 (candidate rows of <= 8 192)
__global__ __launch_bounds__(kRowThreads) void row_select_kernel(const RowSelectArgs a) {
  static_assert(VPT % 4 == 0, "float4 loads");
  MOL_RUN_IF(a.run_if);
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // lds_keys candidates + 2048 exchange
  __shared__ unsigned int ctr[3][16];
  __shared__ unsigned int cursor;
  __shared__ RadixShared rsh;
  __shared__ int64_t f_id[kFuseMaxK], f_inv[kFuseMaxW];
  __shared__ float f_sc[kFuseMaxK];
  __shared__ int f_scratch[kRowThreads / 64 + 2];
  const int row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int k = a.k, lds_keys = a.lds_keys;
  const bool fuse = a.f_invalid != nullptr;   // wave-uniform
  if (fuse)
    for (int i = tid; i < a.f_width; i += kRowThreads) f_inv[i] = a.f_invalid[(int64_t)row * a.f_width + i];   // visible after the barriers below
  unsigned int v[VPT];
  unsigned int lo[KEYS ? VPT : 1];
  int cnt;
  unsigned int begin = 0u;
  RAILS_PHASE(0);
  if constexpr (KEYS) {
    cnt = a.keys_per_row;
    const unsigned long long* src = a.keys_in + (int64_t)row * cnt;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int i = j * kRowThreads + tid;
      const unsigned long long kv = i < cnt ? src[i] : 0ull;
      v[j] = (unsigned int)(kv >> 32);
      lo[j] = (unsigned int)kv;
    }
  } else {
    const int64_t b64 = (int64_t)blockIdx.y * a.chunk;
    begin = (unsigned int)b64;
    cnt = (int)((b64 + a.chunk < a.n ? b64 + a.chunk : a.n) - b64);
    if (a.scores16) {   // bf16 bit patterns (the coarse sample): the same keys as their fp32 values give, half the bytes
      const unsigned short* rowp = a.scores16 + (int64_t)row * a.ld + b64;
      const bool aligned = (reinterpret_cast<uintptr_t>(rowp) & 7) == 0;
#pragma unroll
      for (int jv = 0; jv < VPT / 4; ++jv) {
        const int base = (jv * kRowThreads + tid) * 4;
        if (aligned && base + 3 < cnt) {
          const uint2 x = *reinterpret_cast<const uint2*>(rowp + base);
          v[4 * jv] = orderable(__uint_as_float(x.x << 16)); v[4 * jv + 1] = orderable(__uint_as_float(x.x & 0xFFFF0000u));
          v[4 * jv + 2] = orderable(__uint_as_float(x.y << 16)); v[4 * jv + 3] = orderable(__uint_as_float(x.y & 0xFFFF0000u));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * jv + e] = base + e < cnt ? orderable(__uint_as_float((unsigned int)rowp[base + e] << 16)) : 0u;
        }
      }
    } else {
      const float* rowp = a.scores + (int64_t)row * a.ld + b64;
      const bool aligned = (reinterpret_cast<uintptr_t>(rowp) & 15) == 0;
#pragma unroll
      for (int jv = 0; jv < VPT / 4; ++jv) {
        const int base = (jv * kRowThreads + tid) * 4;
        if (aligned && base + 3 < cnt) {
          const float4 x = *reinterpret_cast<const float4*>(rowp + base);
          v[4 * jv] = orderable(x.x); v[4 * jv + 1] = orderable(x.y); v[4 * jv + 2] = orderable(x.z); v[4 * jv + 3] = orderable(x.w);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * jv + e] = base + e < cnt ? orderable(rowp[base + e]) : 0u;
        }
      }
    }
    lo[0] = 0u;
  }
  auto local_index = [&](int j) { return KEYS ? j * kRowThreads + tid : ((j >> 2) * kRowThreads + tid) * 4 + (j & 3); };
  auto low_of = [&](int j) -> unsigned int {
    if constexpr (KEYS) return lo[j];
    else return ~(begin + (unsigned int)local_index(j));
  };
  unsigned int tmax = 0u;
#pragma unroll
  for (int j = 0; j < VPT; ++j) tmax = v[j] > tmax ? v[j] : tmax;
  if (tid < 48) (&ctr[0][0])[tid] = 0u;
  if (tid == 0) cursor = 0u;
  radix_init<kRowThreads>(rsh);
  for (int i = tid; i < lds_keys; i += kRowThreads) keys[i] = 0ull;
  __syncthreads();
  RAILS_PHASE(1);

  int it = 0;   // rotating counter set: step `it` accumulates into ctr[it % 3] and clears ctr[(it + 1) % 3] before its barrier
  auto emit = [&](unsigned long long kv, int j) {
    if (a.out_scores) {
      const unsigned int pos = ~(unsigned int)(kv & 0xFFFFFFFFull);
      const float sc = unorderable((unsigned int)(kv >> 32));
      int64_t id;
      if constexpr (IDX) id = map_id(a.ids, a.ids_row_stride, a.ids_index, a.ids_index_ld, row, pos);
      else id = a.ids ? a.ids[a.ids_row_stride * row + pos] : (int64_t)pos;
      if (fuse) { f_sc[j] = sc; f_id[j] = id; }
      else { a.out_scores[(int64_t)row * k + j] = sc; a.out_ids[(int64_t)row * k + j] = id; }
    } else {
      a.keys_out[((int64_t)row * gridDim.y + blockIdx.y) * k + j] = kv;
    }
  };
  auto emit_sorted = [&](int npad) {   // sort keys[0, npad) descending, write the first k
    __syncthreads();
    if (npad >= 64 && npad <= 512) {
      unsigned long long kv; unsigned int rank; bool owner;
      block_rank_desc(keys, npad, reinterpret_cast<unsigned int*>(keys + lds_keys), kv, rank, owner);
      if (owner && kv != 0ull && rank < (unsigned int)k) emit(kv, (int)rank);
      return;
    }
    if (npad <= kRowThreads) {
      unsigned long long kv = tid < npad ? keys[tid] : 0ull;
      kv = block_sort_desc(kv, npad, keys + lds_keys);
      if (tid < k) emit(kv, tid);
      return;
    }
    for (int size = 2; size <= npad; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (npad >> 1); t += kRowThreads) {
          const int l = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
          const int h = l | stride;
          const bool desc = ((l & size) == 0);
          const unsigned long long x = keys[l], y = keys[h];
          if ((x < y) == desc) { keys[l] = y; keys[h] = x; }
        }
        __syncthreads();
      }
    }
    for (int j = tid; j < k; j += kRowThreads) emit(keys[j], j);
  };
  auto finish = [&]() {   // fused seen-id filter over the k staged candidates
    if (!fuse) return;
    __syncthreads();
    filter_from_lds<kRowThreads>(f_id, f_sc, k, f_inv, a.f_width, a.f_k, a.out_ids + (int64_t)row * a.f_k, a.out_scores + (int64_t)row * a.f_k, f_scratch);
  };
  auto compact = [&](auto pred) {      // append the keys of the selected elements (any order); overflow is dropped
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < VPT; ++j) c += (local_index(j) < cnt && pred(j)) ? 1u : 0u;
    unsigned int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int x = __shfl_up(incl, o, 64);
      if (lane >= o) incl += x;
    }
    unsigned int base = 0;
    if (lane == 63 && incl) base = atomicAdd(&cursor, incl);
    base = (unsigned int)__shfl((int)base, 63, 64);
    unsigned int at = base + incl - c;
    if (c) {
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        if (local_index(j) < cnt && pred(j)) {
          if (at < (unsigned int)lds_keys) keys[at] = ((unsigned long long)v[j] << 32) | low_of(j);
          ++at;
        }
      }
    }
  };

  if (k <= kRowFastK) {
    unsigned int L = 0u;
    for (int shift = 28; shift >= 16; shift -= 4) {
      unsigned int mine = 0u;
#pragma unroll
      for (int d = 1; d < 16; ++d) {
        const unsigned int cd = (unsigned int)__popcll(__ballot(tmax >= (L | ((unsigned int)d << shift))));
        mine = lane == d ? cd : mine;
      }
      const int buf = it % 3;
      if (lane >= 1 && lane < 16 && mine) atomicAdd(&ctr[buf][lane], mine);
      if (tid < 16) ctr[(it + 1) % 3][tid] = 0u;
      __syncthreads();
      const unsigned int tot = lane < 16 ? ctr[buf][lane] : 0u;
      const unsigned long long ok = __ballot(lane >= 1 && lane < 16 && tot >= (unsigned int)k);   // monotone in the digit
      L |= (unsigned int)__popcll(ok) << shift;
      ++it;
    }
    RAILS_PHASE(2);
    compact([&](int j) { return v[j] >= L; });
    __syncthreads();
    RAILS_PHASE(3);
    const unsigned int m_ge = cursor;
    if (m_ge <= (unsigned int)lds_keys) {
      int npad = 2;
      while (npad < (int)m_ge) npad <<= 1;
      emit_sorted(npad);
      finish();
      RAILS_PHASE(4);
#ifdef RAILS_TOPK_PHASES
      if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) g_phase[5] = m_ge;
#endif
      return;
    }
    __syncthreads();                       // everyone has read the cursor: start over on the general path
    if (tid == 0) cursor = 0u;
    for (int i = tid; i < lds_keys; i += kRowThreads) keys[i] = 0ull;
  }

  // the k-th largest 64-bit key (score, then lowest position first) by MSD radix selection over the registers (radix_select above;
  // the two-bits-per-step bisection it replaced: 32 rows, k = 1 000 of 3 200 40 -> 28 us, 2 000 of 16 000 66 -> 49 us; docs/HISTORY.md R4.5)
  auto hi_of = [&](int j) -> unsigned int { return v[j]; };               // elements past the end and empty key slots were loaded as 0
  auto lo_of = [&](int j, unsigned int z) -> unsigned int {
    if constexpr (KEYS) return lo[j];
    else return ~(begin + (unsigned int)local_index(j) + z);
  };
  RadixSel sel;
  radix_select<VPT>(hi_of, lo_of, (unsigned int)k, rsh, sel);
  compact([&](int j) { return radix_selected(v[j], low_of(j), sel); });
  int npad = 2;
  while (npad < k) npad <<= 1;
  emit_sorted(npad);
  finish();
}

template <int VPT, bool KEYS, bool IDX = false>
static int launch_row_select_t(const RowSelectArgs& a, int rows, int chunks, hipStream_t stream) {
  hipLaunchKernelGGL((row_select_kernel<VPT, KEYS, IDX>), dim3(rows, chunks), dim3(kRowThreads),
                     (a.lds_keys + 2 * kRowThreads) * sizeof(unsigned long long), stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// elements = per-workgroup element count (chunk size, or keys per row)
template <bool KEYS>
static int launch_row_select(RowSelectArgs a, int rows, int chunks, int elements, hipStream_t stream) {
  int lds_keys = 2;
  while (lds_keys < a.k) lds_keys <<= 1;
  if (a.k <= kRowFastK) lds_keys = kRowCandCap;     // room for the pre-filtered candidates
  a.lds_keys = lds_keys;
  if (elements <= 4 * kRowThreads) return launch_row_select_t<4, KEYS>(a, rows, chunks, stream);
  if (elements <= 8 * kRowThreads) return launch_row_select_t<8, KEYS>(a, rows, chunks, stream);
  if (elements <= 16 * kRowThreads) return launch_row_select_t<16, KEYS>(a, rows, chunks, stream);
  if constexpr (!KEYS) {
    if (elements <= 28 * kRowThreads) return launch_row_select_t<28, KEYS>(a, rows, chunks, stream);
    if (elements <= 40 * kRowThreads) return launch_row_select_t<40, KEYS>(a, rows, chunks, stream);
    if (elements <= 48 * kRowThreads) return launch_row_select_t<48, KEYS>(a, rows, chunks, stream);
  } else {
    if (elements <= 24 * kRowThreads) return launch_row_select_t<24, KEYS>(a, rows, chunks, stream);
  }
  set_error("row select: %d elements per workgroup", elements);
  return kErrUnsupported;
}

// two-level plan for n > kRowMaxN: chunks of <= kRowMaxN elements (multiples of 4, so float4 loads stay aligned when the
// row is), chunks * k keys for the second level
static bool two_level_plan_at(int64_t n, int k, int64_t max_chunk, int* chunks, int64_t* chunk) {
  const int64_t c = (n + max_chunk - 1) / max_chunk;
  if (c * k > 24 * kRowThreads) return false;   // the second level holds 24 576 keys per row
  int64_t len = (n + c - 1) / c;
  len = (len + 3) / 4 * 4;
  if (len > kRowMaxN || len < k) return false;
  *chunks = (int)((n + len - 1) / len);
  *chunk = len;
  // the last chunk must still hold k elements (its k keys are all real); otherwise leave it to the radix path
  if (n - (int64_t)(*chunks - 1) * len < k) return false;
  return true;
}
// max_k: kRowFastK for ordinary calls -- beyond it the multi-launch radix path is faster (32 x 695 762, k = 1 000: 111 vs 186 us) --
// and kRowMaxK for calls under a launch predicate: those are the fallbacks behind a device-side verdict, no-ops in the common case,
// where what counts is how many launches they are (2 instead of 10: ~40 us of every MoLAvgTopK call with K' >= 1 000 at amzn-books size)
static bool two_level_plan(int64_t n, int k, int* chunks, int64_t* chunk, int rows = 1 << 30, int max_k = 512) {
  if (k > max_k) return false;
  // first-level chunk size: RAILS_ROW_CHUNK (measurement override), else the largest a workgroup holds in registers -- half of that
  // for up to eight rows, where 49 152-element chunks leave most of the chip idle (695 762 elements: 1 row 29 -> 27 us, 4 rows
  // 37 -> 31, 8 rows 37 -> 32; from 32 rows on smaller chunks only add second-level work: 63 -> 75 us).  A plan that does not work
  // out with the smaller chunks is retried with the full ones, so feasibility does not depend on the row count.
  static const int64_t forced = [] { const char* e = getenv("RAILS_ROW_CHUNK"); const int64_t v = e ? atoll(e) : 0; return v >= 4096 && v <= kRowMaxN ? v / 4 * 4 : (int64_t)0; }();
  const int64_t first = forced ? forced : (rows <= 8 ? (int64_t)kRowMaxN / 2 : (int64_t)kRowMaxN);
  if (two_level_plan_at(n, k, first, chunks, chunk)) return true;
  return first != kRowMaxN && two_level_plan_at(n, k, kRowMaxN, chunks, chunk);
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t topk_workspace_bytes(int rows, int64_t n, int k) {
  if (n <= kSortCap || (n <= 48 * 1024 && k <= 4096)) return 256;   // single-launch paths need no workspace
  // sized for the radix path; the two-level path's rows * chunks * k keys (<= rows * 16384) are checked against it too
  size_t b = align_up(sizeof(SelectState) * (size_t)rows, 256);
  b += align_up(sizeof(unsigned int) * (size_t)kRadixPasses * rows * kBins, 256);
  b += align_up(sizeof(unsigned long long) * (size_t)rows * next_pow2(k < 2 ? 2 : k), 256);   // npad candidate slots per row
  const size_t two_level = sizeof(unsigned long long) * (size_t)rows * 24 * 1024;
  return b > two_level ? b : two_level;
}

static int ensure_sort_lds() {
  static DynLdsOnce once;
  return ensure_dyn_lds(once, reinterpret_cast<const void*>(&sort_emit_kernel), kSortCap * (int)sizeof(unsigned long long));
}

bool topk_can_fuse_filter(int64_t n, int k, int width, int k_out) {
  return n > 1024 && n < (1ll << 32) && k <= kFuseMaxK && k <= kRowMaxK && width >= 0 && width <= kFuseMaxW && k_out > 0 && k_out <= k &&
         (n <= kRowMaxN || [&] { int c; int64_t l; return two_level_plan(n, k, &c, &l); }());
}

// bf16 sources run the row-select launches only (one launch, or per-chunk winners + winners' winners)
bool topk_bf16_source_ok(int rows, int64_t n, int k) {
  return n > 1024 && n < (1ll << 32) && k <= kRowMaxK && (n <= kRowMaxN || [&] { int c; int64_t l; return two_level_plan(n, k, &c, &l, rows); }());
}

int topk(const float* scores, int64_t ld, int rows, int64_t n, int k, const int64_t* ids, int64_t ids_row_stride,
         float* out_scores, int64_t* out_ids, void* ws, size_t ws_bytes, int n_cu, hipStream_t stream,
         const int64_t* f_invalid, int f_width, int f_k, const unsigned short* scores16, const int32_t* run_if,
         const int64_t* ids_index, int64_t ids_index_ld) {
  if (rows <= 0 || k <= 0) return kOk;
  if (scores16 && !topk_bf16_source_ok(rows, n, k)) { set_error("topk: no bf16-source path at n = %lld, k = %d", (long long)n, k); return kErrUnsupported; }
  if (f_invalid && !topk_can_fuse_filter(n, k, f_width, f_k)) { set_error("topk: the seen-id filter cannot be fused at n = %lld, k = %d, width = %d", (long long)n, k, f_width); return kErrUnsupported; }
  if (k > kSortCap) { set_error("k = %d exceeds the in-LDS sort capacity (%d)", k, kSortCap); return kErrUnsupported; }
  if (n >= (1ll << 32)) { set_error("n = %lld does not fit 32-bit positions; shard the corpus", (long long)n); return kErrUnsupported; }
  if (ensure_sort_lds() != kOk) return kErrLaunch;
  const int32_t* pred = run_if;   // the caller's launch predicate (NULL for the internal selections of the fused scans)
  // 512 < n <= 1024 (the K' = 1 000 candidates of a two-pass rerank, k = 120): the register-resident selection's fast path instead of a
  // full sort of 1 024 keys (13.6 -> 7 us per 32 rows)
  if (ids_index && n > kSortCap) { set_error("topk: ids_index needs n <= %d", kSortCap); return kErrUnsupported; }
  // k > 512 of a short row: when the k winners fill the same power of two of sort slots as the whole row would (k = 3 200 of 3 200,
  // 2 561 of 4 096), selecting first buys nothing -- the row is sorted whole (n = 3 200 = k: 52 -> 41 us per 32 rows; k = 1 000 of
  // 3 200 stays on the selection: 28 vs 41 us; tools/r04_topk_bigk_ab.sh)
  const bool sort_whole = k > kRowFastK && n <= kSortCap && next_pow2((int)n) <= next_pow2(k) && !scores16 && !f_invalid;
  // candidate rows (ids through ids_index): the register-resident selection up to 8 192 candidates (16 per thread spills), the seen-id filter fused where asked
  // (round 6: the union of a Naive / Comb rerank, 6 400-7 400 candidates of which get_top_k_outputs wants k + |seen| <= 512)
  if (!sort_whole && ids_index && n > 512 && n <= 8 * kRowThreads && k <= kRowMaxK && !scores16) {
    RowSelectArgs a{};
    a.run_if = pred;
    a.scores = scores; a.ld = ld; a.n = n; a.k = k; a.chunk = n; a.ids = ids; a.ids_row_stride = ids_row_stride; a.out_scores = out_scores; a.out_ids = out_ids;
    a.ids_index = ids_index; a.ids_index_ld = ids_index_ld;
    a.f_invalid = f_invalid; a.f_width = f_width; a.f_k = f_k;
    int lds_keys = 2;                                   // as launch_row_select sets it
    while (lds_keys < a.k) lds_keys <<= 1;
    a.lds_keys = a.k <= kRowFastK ? kRowCandCap : lds_keys;
    if (n <= 4 * kRowThreads) return launch_row_select_t<4, false, true>(a, rows, 1, stream);
    return launch_row_select_t<8, false, true>(a, rows, 1, stream);
  }
  if (ids_index && f_invalid) { set_error("topk: the seen-id filter over candidate rows needs 1024 < n <= %d", 8 * kRowThreads); return kErrUnsupported; }
  if (!sort_whole && (n > 1024 || (n > 512 && k <= kRowFastK && !scores16 && !f_invalid)) && k <= kRowMaxK && !ids_index) {
    RowSelectArgs a{};
    a.run_if = pred;
    a.scores = scores; a.scores16 = scores16; a.ld = ld; a.n = n; a.k = k;
    if (n <= kRowMaxN) {                 // one launch
      a.chunk = n; a.ids = ids; a.ids_row_stride = ids_row_stride; a.out_scores = out_scores; a.out_ids = out_ids;
      a.f_invalid = f_invalid; a.f_width = f_width; a.f_k = f_k;
      return launch_row_select<false>(a, rows, 1, (int)n, stream);
    }
    int chunks; int64_t chunk;
    if (two_level_plan(n, k, &chunks, &chunk, rows, pred ? kRowMaxK : kRowFastK)) {   // two launches: per-chunk winners, then the winners' winners
      if (ws_bytes < topk_workspace_bytes(rows, n, k)) { set_error("top-k workspace too small"); return kErrNoMem; }
      unsigned long long* lvl1 = static_cast<unsigned long long*>(ws);
      a.chunk = chunk; a.keys_out = lvl1;
      const int rc = launch_row_select<false>(a, rows, chunks, (int)chunk, stream);
      if (rc != kOk) return rc;
      RowSelectArgs b{};
      b.run_if = pred;
      b.keys_in = lvl1; b.keys_per_row = chunks * k; b.k = k;
      b.ids = ids; b.ids_row_stride = ids_row_stride; b.out_scores = out_scores; b.out_ids = out_ids;
      b.f_invalid = f_invalid; b.f_width = f_width; b.f_k = f_k;
      return launch_row_select<true>(b, rows, 1, chunks * k, stream);
    }
  }
  if (n <= kSortCap) {
    const int npad = next_pow2((int)n < 2 ? 2 : (int)n);
    hipLaunchKernelGGL(sort_emit_kernel, dim3(rows), dim3(kSortThreads), (npad <= kSortThreads ? 3 * npad : npad) * sizeof(unsigned long long), stream,
                       scores, ld, n, n, (const unsigned long long*)nullptr, (int64_t)0, 0, k, npad, ids, ids_row_stride,
                       out_scores, out_ids, (unsigned long long*)nullptr, (int64_t)0, pred, (const SelectState*)nullptr, ids_index, ids_index_ld);
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
  }
  if (ws_bytes < topk_workspace_bytes(rows, n, k)) { set_error("top-k workspace too small"); return kErrNoMem; }
  char* base = static_cast<char*>(ws);
  SelectState* st = reinterpret_cast<SelectState*>(base);
  base += align_up(sizeof(SelectState) * (size_t)rows, 256);
  unsigned int* hist = reinterpret_cast<unsigned int*>(base);
  const size_t hist_bytes = sizeof(unsigned int) * (size_t)kRadixPasses * rows * kBins;
  base += align_up(hist_bytes, 256);
  const size_t state_bytes = (size_t)(base - static_cast<char*>(ws));
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(base);

  if (hipMemsetAsync(ws, 0, state_bytes, stream) != hipSuccess) return kErrLaunch;   // state + histograms
  // enough workgroups to fill the chip (four per CU), at least 8K elements each
  static const int radix_wgs = [] { const char* e = getenv("RAILS_RADIX_WGS"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 16384 ? v : 1024; }();   // workgroups per launch (override for measurements): 512 / 1024 / 2048 / 4096 -> 135 / 131 / 140 / 147 us at k' = 2561, 32 x 695 762 (fewer per-workgroup histogram merges)
  int64_t chunks = (radix_wgs + rows - 1) / rows;
  const int64_t max_chunks = (n + 8191) / 8192;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  const int64_t chunk = (n + chunks - 1) / chunks;
  const int npad = next_pow2(k < 2 ? 2 : k);
  for (int pass = 0; pass < kRadixPasses; ++pass) {
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)chunks, rows), dim3(kHistThreads), 0, stream, scores, ld, n, st, hist, pass,
                       chunk, pred);
    hipLaunchKernelGGL(pick_bin_kernel, dim3(rows), dim3(64), 0, stream, st, hist, pass, rows, k, pred, npad);
  }
  hipLaunchKernelGGL(tie_resolve_kernel, dim3(rows), dim3(kTieThreads), 0, stream, scores, ld, n, st, pred);
  hipLaunchKernelGGL(compact_kernel, dim3((unsigned)chunks, rows), dim3(kHistThreads), 0, stream, scores, ld, n, st, cand,
                     (int64_t)npad, k, chunk, pred);
  hipLaunchKernelGGL(sort_emit_kernel, dim3(rows), dim3(kSortThreads), (npad <= kSortThreads ? 3 * npad : npad) * sizeof(unsigned long long), stream, scores, ld, n, n,
                     cand, (int64_t)npad, npad, k, npad, ids, ids_row_stride, out_scores, out_ids, (unsigned long long*)nullptr,
                     (int64_t)0, pred, (const SelectState*)st, ids_index, ids_index_ld);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// Top-k of rows of 64-bit keys (score bits << 32 | ~position), zero-padded: out (score, position), descending.
// Used by the fused coarse top-K' (mol_coarse.hip) on its candidate lists.
int select_keys(const unsigned long long* keys, int rows, int keys_per_row, int k, float* out_scores, int64_t* out_pos,
                hipStream_t stream) {
  if (rows <= 0 || k <= 0) return kOk;
  if (k > kRowMaxK || keys_per_row > 24 * kRowThreads || k > keys_per_row) {
    set_error("select_keys: k = %d of %d keys per row is out of range", k, keys_per_row);
    return kErrUnsupported;
  }
  RowSelectArgs b{};
  b.keys_in = keys; b.keys_per_row = keys_per_row; b.k = k;
  b.out_scores = out_scores; b.out_ids = out_pos;
  return launch_row_select<true>(b, rows, 1, keys_per_row, stream);
}

// ---- selection over a row's candidate SUB-LISTS (the fused coarse top-K' of mol_coarse.hip) -------------------------------------
// The select scan leaves, per row, n_sub sub-lists of cap / n_sub slots with their fill counts.  One workgroup per row:
//   * reads the counts, writes the row's candidate count (cap + 1 when a sub-list overflowed) and raises *out_flag when the row is
//     not exact (overflow, or fewer than k candidates) -- the counts kernel and the range check of the caller, in this launch;
//   * loads the FILLED slots only (the lists need no zeroing between calls) into registers as 64-bit keys;
//   * finds the k-th largest key by MSD radix selection, eight bits per pass: a 256-bin LDS histogram of the keys that match the
//     resolved prefix (a wave adds its most common digit with one atomic), one wave turns it into the digit.  Bytes in which all
//     keys agree -- bits of OR ^ AND over the row -- take no pass (bf16 scores: the two low bytes of the score word; the top byte
//     of the positions), and the passes stop as soon as every key that still matches is wanted: 3-5 passes of two barriers for a
//     row of ~4 000 candidates, where the two-bits-per-step bisection of row_select_kernel took 17 (+ 16 on the positions when
//     the k-th score is tied, which bf16 scores almost always are);
//   * compacts the k winners into LDS, sorts them (block_sort_desc / block_sort_desc_multi) and writes (score, position).
// K' = 1 000 of ~4 100 candidates per row, 32 rows: 56.7 us (row_select_kernel<8, true>) + 5.1 (counts) + 5.1 (memset of the
// lists) -> 24.6 us (docs/HISTORY.md R4.5).
struct SubSelArgs {
  const unsigned long long* keys; const unsigned int* counts; int cap, n_sub, k, npad;
  float* out_scores; int64_t* out_pos; int32_t* out_counts; int32_t* out_flag;
};

template <int VPT>
__global__ __launch_bounds__(kRowThreads) void sublist_select_kernel(const SubSelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];   // npad sort slots (+ 2 npad exchange slots when npad <= 1024)
  __shared__ unsigned int cnt_s[64];
  __shared__ RadixShared rsh;
  __shared__ unsigned int cursor;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int k = a.k, subcap = a.cap / a.n_sub, npad = a.npad;
  if (tid < a.n_sub) cnt_s[tid] = a.counts[(int64_t)row * a.n_sub + tid];
  radix_init<kRowThreads>(rsh);
  if (tid == 0) cursor = 0u;
  for (int i = tid; i < npad; i += kRowThreads) skeys[i] = 0ull;
  // every slot is requested now, next to the counts (one round trip to memory instead of two); the slots beyond a sub-list's count
  // hold leftovers of earlier calls and are masked below
  unsigned long long raw[VPT];
  {
    const unsigned long long* src0 = a.keys + (int64_t)row * a.cap;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int i = j * kRowThreads + tid;
      raw[j] = i < a.cap ? src0[i] : 0ull;
    }
  }
  __syncthreads();
  unsigned int total = 0u, held = 0u;
  bool over = false;
  for (int sub = 0; sub < a.n_sub; ++sub) {
    const unsigned int c = cnt_s[sub];
    total += c;
    held += c < (unsigned int)subcap ? c : (unsigned int)subcap;
    over |= c > (unsigned int)subcap;
  }
  if (tid == 0) {
    a.out_counts[row] = over ? a.cap + 1 : (int32_t)total;
    if ((over || total < (unsigned int)k) && a.out_flag) *a.out_flag = 1;   // every writer stores the same value
  }
  const unsigned int want = held < (unsigned int)k ? held : (unsigned int)k;   // < k only on rows the caller redoes
  unsigned int khi[VPT], klo[VPT];
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = j * kRowThreads + tid;
    const int sub = i / subcap;
    const bool filled = i < a.cap && (unsigned int)(i - sub * subcap) < cnt_s[sub < a.n_sub ? sub : 0];
    const unsigned long long kv = filled ? raw[j] : 0ull;
    khi[j] = (unsigned int)(kv >> 32);
    klo[j] = (unsigned int)kv;
  }
  RadixSel sel;
  radix_select<VPT>([&](int j) { return khi[j]; }, [&](int j, unsigned int) { return klo[j]; }, want, rsh, sel);
  // exactly `want` keys are selected
  {
    unsigned int c = 0u;
#pragma unroll
    for (int j = 0; j < VPT; ++j) c += radix_selected(khi[j], klo[j], sel) ? 1u : 0u;
    unsigned int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int x = __shfl_up(incl, o, 64); if (lane >= o) incl += x; }
    unsigned int base = 0u;
    if (lane == 63 && incl) base = atomicAdd(&cursor, incl);
    base = (unsigned int)__shfl((int)base, 63, 64);
    unsigned int at = base + incl - c;
    if (c) {
#pragma unroll
      for (int j = 0; j < VPT; ++j)
        if (radix_selected(khi[j], klo[j], sel)) { if (at < (unsigned int)npad) skeys[at] = ((unsigned long long)khi[j] << 32) | klo[j]; ++at; }
    }
  }
  __syncthreads();
  auto emit = [&](unsigned long long kv, int j) {
    a.out_scores[(int64_t)row * k + j] = kv ? unorderable((unsigned int)(kv >> 32)) : -INFINITY;
    a.out_pos[(int64_t)row * k + j] = kv ? (int64_t)(~(unsigned int)(kv & 0xFFFFFFFFull)) : 0;   // an unfilled slot names item 0 (such a row is redone by the caller)
  };
  if (npad <= kRowThreads) {
    unsigned long long kv = tid < npad ? skeys[tid] : 0ull;
    kv = block_sort_desc(kv, npad, skeys + npad);
    if (tid < k) emit(kv, tid);
  } else {
    if (npad == 2 * kRowThreads) block_sort_desc_multi<2>(skeys);
    else block_sort_desc_multi<4>(skeys);
    for (int j = tid; j < k; j += kRowThreads) emit(skeys[j], j);
  }
}

// ---- the same for MANY SHORT rows (the component scans of MoLNaiveTopK / MoLCombTopK: 2 048 rows of ~50-500 candidates) ----------------
// One 256-thread workgroup per row -- eight resident per CU, where the 1 024-thread kernel above runs its 2 048 workgroups four deep and
// requests all `cap` slots of every row (33 MB) before it knows the counts: 58 us for 2 048 rows, 41 of them with no candidate at all.
// Here the FILLED slots are loaded into LDS and every key's descending rank is counted against the others (broadcast LDS reads; keys are
// distinct); keys of rank < k go straight to their output slot.  cap <= kSmallCap, k <= cap.
constexpr int kSmallThreads = 256, kSmallCap = 2048;
__global__ __launch_bounds__(kSmallThreads) void sublist_rank_kernel(const SubSelArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned long long skeys[kSmallCap];
  __shared__ unsigned int cnt_s[64], base_s[65];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int k = a.k, subcap = a.cap / a.n_sub;
  if (tid < a.n_sub) cnt_s[tid] = a.counts[(int64_t)row * a.n_sub + tid];
  __syncthreads();
  if (tid == 0) {
    unsigned int total = 0u, held = 0u;
    bool over = false;
    for (int sub = 0; sub < a.n_sub; ++sub) {
      const unsigned int c = cnt_s[sub];
      base_s[sub] = held;
      total += c;
      held += c < (unsigned int)subcap ? c : (unsigned int)subcap;
      over |= c > (unsigned int)subcap;
    }
    base_s[a.n_sub] = held;
    a.out_counts[row] = over ? a.cap + 1 : (int32_t)total;
    if ((over || total < (unsigned int)k) && a.out_flag) *a.out_flag = 1;
  }
  __syncthreads();
  const int held = (int)base_s[a.n_sub];
  const unsigned long long* src = a.keys + (int64_t)row * a.cap;
  for (int i = tid; i < a.cap; i += kSmallThreads) {      // slot i of the row: sub-list i / subcap, filled iff its index is below the count
    const int sub = i / subcap, j = i - sub * subcap;
    if ((unsigned int)j < cnt_s[sub] ) skeys[base_s[sub] + j] = src[i];
  }
  __syncthreads();
  for (int i = tid; i < held; i += kSmallThreads) {
    const unsigned long long mine = skeys[i];
    int rank = 0;
    const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(skeys);
    for (int j = 0; j < held / 2; ++j) { const ulonglong2 x = p2[j]; rank += x.x > mine ? 1 : 0; rank += x.y > mine ? 1 : 0; }
    if (held & 1) rank += skeys[held - 1] > mine ? 1 : 0;
    if (rank < k) {
      a.out_scores[(int64_t)row * k + rank] = unorderable((unsigned int)(mine >> 32));
      a.out_pos[(int64_t)row * k + rank] = (int64_t)(~(unsigned int)(mine & 0xFFFFFFFFull));
    }
  }
  for (int j = held + tid; j < k; j += kSmallThreads) {     // a row with fewer than k candidates (redone by the caller): defined filler
    a.out_scores[(int64_t)row * k + j] = -INFINITY;
    a.out_pos[(int64_t)row * k + j] = 0;
  }
}

// The r-th largest of every row of bf16 bit patterns (the threshold of a fused scan from its block of running maxima), one WAVE per row:
// the row's 16-bit orderable keys in registers; a 256-bin LDS histogram of (key - row minimum) >> shift, shift such that the row's RANGE
// spreads over the bins (the maxima of a row share their high byte: a histogram of the high byte itself sent every lane to one bin, 41 us
// for 2 048 rows; bisecting the sixteen key bits cost 4 000 VALU instructions per row, 38 us), then -- shift > 0 -- one more histogram
// inside the bin that holds the r-th largest.  thr[row] = that value as fp32: row_select_kernel's r-th output on the same row.
constexpr int kKthRegs = 64;      // rows of up to 64 * 2 * 64 = 8 192 values
__global__ __launch_bounds__(256) void bf16_rows_kth_kernel(const unsigned short* __restrict__ rows16, int64_t ld, int n_rows, int n, int r,
                                                            float* __restrict__ thr) {
  __shared__ __attribute__((aligned(16))) unsigned int hist[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n_rows) return;          // (whole waves: no barrier below)
  unsigned int* h = hist[wave];
  const unsigned int* src = reinterpret_cast<const unsigned int*>(rows16 + (int64_t)row * ld);      // (ld and the row starts are even: pairs of values)
  const int n2 = n / 2, regs = (n2 + 63) / 64;
  unsigned int kv[kKthRegs];
  auto key16 = [](unsigned int b) -> unsigned int { return (b & 0x8000u) ? (~b & 0xFFFFu) : (b | 0x8000u); };
  unsigned int mn = 0xFFFFu, mx = 0u;
#pragma unroll
  for (int j = 0; j < kKthRegs; ++j) {
    kv[j] = 0xFFFFFFFFu;                       // no pair here
    if (j < regs) {
      const int i = j * 64 + lane;
      if (i < n2) {
        const unsigned int p = src[i], k0 = key16(p & 0xFFFFu), k1 = key16(p >> 16);
        kv[j] = k0 | (k1 << 16);
        mn = min(mn, min(k0, k1));
        mx = max(mx, max(k0, k1));
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (unsigned int)__shfl_xor((int)mn, o, 64)); mx = max(mx, (unsigned int)__shfl_xor((int)mx, o, 64)); }
  const unsigned int range = mx - mn;
  const int shift = range < 256u ? 0 : 32 - __builtin_clz(range) - 8;      // (range >> shift) <= 255
  auto pick = [&](unsigned int want, unsigned int& left) -> unsigned int {      // the bin holding the want-th largest counted key; left = its rank inside the bin
    const uint4 c4 = reinterpret_cast<const uint4*>(h)[lane];
    const unsigned int sum4 = c4.x + c4.y + c4.z + c4.w;
    unsigned int incl = sum4;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_down(incl, o, 64); if (lane + o < 64) incl += t; }
    unsigned int cum = incl - sum4;
    const unsigned int c[4] = {c4.x, c4.y, c4.z, c4.w};
    unsigned int bin = 0xFFFFFFFFu, lf = 0u;
#pragma unroll
    for (int b = 3; b >= 0; --b) {
      if (cum < want && want <= cum + c[b]) { bin = 4u * lane + b; lf = want - cum; }
      cum += c[b];
    }
    const unsigned long long has = __ballot(bin != 0xFFFFFFFFu);
    const int src_lane = has ? __ffsll((long long)has) - 1 : 0;
    left = (unsigned int)__shfl((int)lf, src_lane, 64);
    return (unsigned int)__shfl((int)bin, src_lane, 64);
  };
  for (int i = lane; i < 256; i += 64) h[i] = 0u;
#pragma unroll
  for (int j = 0; j < kKthRegs; ++j)
    if (j < regs && kv[j] != 0xFFFFFFFFu) {
      atomicAdd(&h[((kv[j] & 0xFFFFu) - mn) >> shift], 1u);
      atomicAdd(&h[((kv[j] >> 16) - mn) >> shift], 1u);
    }
  unsigned int left = 0u;
  const unsigned int b1 = pick((unsigned int)r, left);
  unsigned int key = mn + (b1 << shift);
  if (shift > 0) {
    const unsigned int low = (1u << shift) - 1u;
    for (int i = lane; i < 256; i += 64) h[i] = 0u;
#pragma unroll
    for (int j = 0; j < kKthRegs; ++j)
      if (j < regs && kv[j] != 0xFFFFFFFFu) {
        const unsigned int d0 = (kv[j] & 0xFFFFu) - mn, d1 = (kv[j] >> 16) - mn;
        if ((d0 >> shift) == b1) atomicAdd(&h[d0 & low], 1u);
        if ((d1 >> shift) == b1) atomicAdd(&h[d1 & low], 1u);
      }
    unsigned int left2 = 0u;
    key += pick(left, left2);
  }
  const unsigned int bits = (key & 0x8000u) ? (key & 0x7FFFu) : (~key & 0xFFFFu);
  if (lane == 0) thr[row] = __uint_as_float(bits << 16);
}
int bf16_rows_kth(const unsigned short* rows16, int64_t ld, int n_rows, int n, int r, float* thr, hipStream_t stream) {
  if (n_rows <= 0) return kOk;
  if (r < 1 || r > n || n > kKthRegs * 128 || (n & 1) || (ld & 1)) { set_error("bf16_rows_kth: r = %d of n = %d (even, <= %d), ld = %lld", r, n, kKthRegs * 128, (long long)ld); return kErrInvalid; }
  hipLaunchKernelGGL(bf16_rows_kth_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, rows16, ld, n_rows, n, r, thr);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// keys[row][sub][slot], counts[row][sub]: the k largest keys of every row as (score, position), descending; out_counts[row] = its
// candidates (cap + 1: a sub-list overflowed); *out_flag (optional, zeroed by the caller) = 1 if some row has not k <= count <= cap.
int select_sublists(const unsigned long long* keys, const unsigned int* counts, int rows, int cap, int n_sub, int k, float* out_scores,
                    int64_t* out_pos, int32_t* out_counts, int32_t* out_flag, hipStream_t stream) {
  static_assert(kSortThreads == kRowThreads, "block_sort_desc_multi sorts KPT * kSortThreads keys with the workgroup of this kernel");
  if (rows <= 0 || k <= 0) return kOk;
  if (k > kRowMaxK || cap > 24 * kRowThreads || k > cap || n_sub < 1 || n_sub > 64 || cap % n_sub) {
    set_error("select_sublists: k = %d of %d slots in %d sub-lists is out of range", k, cap, n_sub);
    return kErrUnsupported;
  }
  SubSelArgs a{};
  a.keys = keys; a.counts = counts; a.cap = cap; a.n_sub = n_sub; a.k = k;
  a.npad = next_pow2(k < 64 ? 64 : k);
  a.out_scores = out_scores; a.out_pos = out_pos; a.out_counts = out_counts; a.out_flag = out_flag;
  if (rows >= 512 && cap <= kSmallCap) {      // many short rows (the component scans)
    hipLaunchKernelGGL(sublist_rank_kernel, dim3(rows), dim3(kSmallThreads), 0, stream, a);
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
  }
  const size_t lds = (size_t)(a.npad <= kRowThreads ? 3 * a.npad : a.npad) * sizeof(unsigned long long);
  if (cap <= 4 * kRowThreads) hipLaunchKernelGGL((sublist_select_kernel<4>), dim3(rows), dim3(kRowThreads), lds, stream, a);
  else if (cap <= 8 * kRowThreads) hipLaunchKernelGGL((sublist_select_kernel<8>), dim3(rows), dim3(kRowThreads), lds, stream, a);
  else if (cap <= 16 * kRowThreads) hipLaunchKernelGGL((sublist_select_kernel<16>), dim3(rows), dim3(kRowThreads), lds, stream, a);
  else hipLaunchKernelGGL((sublist_select_kernel<24>), dim3(rows), dim3(kRowThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- seen-id filter ----------------------------------------------------------------------------
// One workgroup per row.  Literal restatement of indexing/candidate_index.py:156-175:
//   valid  = not seen, and among the first k such
//   if fewer than k are valid, back-fill with the first (k - #valid) of the others, in position order
//   output = the selected positions in ascending position order (exactly k of them)
template <int kFilterThreads>      // 256; 1 024 for long candidate rows (MoLNaiveTopK100's 6 400 per query: 45 -> ~15 us)
__global__ __launch_bounds__(kFilterThreads) void filter_seen_kernel(const int64_t* __restrict__ top_ids,
                                                                    const float* __restrict__ top_scores, int k_prime,
                                                                    const int64_t* __restrict__ invalid, int width,
                                                                    int k, int64_t* __restrict__ out_ids,
                                                                    float* __restrict__ out_scores) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  int64_t* inv = reinterpret_cast<int64_t*>(fsm);                        // [width]
  unsigned char* ok = reinterpret_cast<unsigned char*>(inv + width);     // [k_prime]: 1 = not seen
  __shared__ int seg_ok[kFilterThreads], seg_bad[kFilterThreads];
  __shared__ int total_ok;
  const int row = blockIdx.x;
  const int64_t* ids = top_ids + (int64_t)row * k_prime;
  for (int i = threadIdx.x; i < width; i += kFilterThreads) inv[i] = invalid[(int64_t)row * width + i];
  __syncthreads();
  for (int j = threadIdx.x; j < k_prime; j += kFilterThreads) {
    const int64_t id = ids[j];
    bool seen = false;
    for (int w = 0; w < width; ++w) seen |= (inv[w] == id);
    ok[j] = seen ? 0 : 1;
  }
  __syncthreads();
  // contiguous segment per thread; exclusive prefix of not-seen counts over segments
  const int seg = (k_prime + kFilterThreads - 1) / kFilterThreads;
  const int s0 = threadIdx.x * seg, s1 = (s0 + seg < k_prime) ? s0 + seg : k_prime;
  int c = 0;
  for (int j = s0; j < s1; ++j) c += ok[j];
  // block-wide exclusive scan of the per-thread counts: shuffle scan inside each wave, wave totals through LDS (a serial scan by
  // thread 0 cost ~7 us of this kernel's 14)
  auto block_exclusive_scan = [&](int v, int* totals, int& total) -> int {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) totals[wv] = inc;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int t = 0; t < kFilterThreads / 64; ++t) { const int tv = totals[t]; if (t < wv) base += tv; total += tv; }
    __syncthreads();
    return base + inc - v;
  };
  int tot;
  seg_ok[threadIdx.x] = block_exclusive_scan(c, seg_bad, tot);
  if (threadIdx.x == 0) total_ok = tot;
  __syncthreads();
  const int n_valid = total_ok < k ? total_ok : k;   // valid = not seen and cumsum <= k
  const int gap = k - n_valid;
  // "invalid" = everything that is not valid: seen ids AND not-seen ids beyond the first k
  int okc = seg_ok[threadIdx.x];
  c = 0;
  for (int j = s0; j < s1; ++j) { const bool valid = ok[j] && (okc + 1 <= k); okc += ok[j]; c += valid ? 0 : 1; }
  __syncthreads();   // seg_bad served as the first scan's scratch
  {
    __shared__ int wave_tot[kFilterThreads / 64];
    const int ex = block_exclusive_scan(c, wave_tot, tot);
    seg_bad[threadIdx.x] = ex;
  }
  __syncthreads();
  okc = seg_ok[threadIdx.x];
  int badc = seg_bad[threadIdx.x];
  for (int j = s0; j < s1; ++j) {
    const bool valid = ok[j] && (okc + 1 <= k);
    okc += ok[j];
    int out_pos = -1;
    if (valid) {
      // selected positions before j: valid ones (= okc_before, all <= k) + back-filled ones (min(badc, gap))
      out_pos = (okc - 1) + (badc < gap ? badc : gap);
    } else {
      if (badc + 1 <= gap) out_pos = (okc < k ? okc : k) + badc;
      badc += 1;
    }
    if (out_pos >= 0 && out_pos < k) {
      out_ids[(int64_t)row * k + out_pos] = ids[j];
      out_scores[(int64_t)row * k + out_pos] = top_scores[(int64_t)row * k_prime + j];
    }
  }
}

// ---- the call's verdict from its rows' verdicts (rails_candidates_finish, rails_merge_candidates_verdict) ---------------------------
// Every row's workgroup stores the row's (error, margin, guard magnitude, flags) as ONE 16-byte word of the workspace, fences, and bumps
// the arrival counter; the LAST workgroup to arrive reduces the rows with its first wave, folds the call into `state`
// (rails_rescore_verdict's layout) and mirrors it into pinned host memory -- the words first, the call counter (which the host polls)
// last.  `call`: [0] arrivals (zero between calls), [8 ...) the rows' words.  All threads of the workgroup call (one barrier).
struct RowVerdict { float err, gap, grd; unsigned int flags; };       // flags: 1 = the row failed, 2 = the row was bad (NaN, guard)
__device__ __forceinline__ unsigned int gap_key(float g) { return ~orderable(g); }
__device__ __forceinline__ void verdict_commit(unsigned int* call, int row, int rows, int fail, int bad, float err, float gap, float grd, float default_eps,
                                               float safety, float* st, float* state_host, int* s_last) {
  float4* slots = reinterpret_cast<float4*>(call + 8);
  if (threadIdx.x == 0) {
    slots[row] = float4{bad ? 0.0f : err, gap == gap ? gap : -INFINITY, grd, __uint_as_float((fail ? 1u : 0u) | (bad ? 2u : 0u))};
    __threadfence();
    *s_last = atomicAdd(&call[0], 1u) == (unsigned int)rows - 1u ? 1 : 0;
  }
  __syncthreads();
  if (!*s_last || threadIdx.x >= 64) return;
  __threadfence();
  float e = 0.0f, g = INFINITY, gd = 0.0f;
  unsigned int fl = 0u;
  for (int r = threadIdx.x; r < rows; r += 64) {
    const volatile float* vp = reinterpret_cast<const volatile float*>(&slots[r]);
    e = fmaxf(e, vp[0]); g = fminf(g, vp[1]); gd = fmaxf(gd, vp[2]); fl |= __float_as_uint(vp[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    e = fmaxf(e, __shfl_xor(e, o, 64)); g = fminf(g, __shfl_xor(g, o, 64)); gd = fmaxf(gd, __shfl_xor(gd, o, 64)); fl |= (unsigned int)__shfl_xor((int)fl, o, 64);
  }
  if (threadIdx.x != 0) return;
  const bool any_bad = (fl & 2u) != 0u;
  float seen = st[0];
  if (!any_bad) seen = fmaxf(seen, e);
  const int redo = (fl & 1u) ? 1 : 0;
  const float calls = st[5] + 1.0f, redone = st[6] + (redo ? 1.0f : 0.0f), grd_all = fmaxf(st[7], gd), eps_used = fmaxf(default_eps, safety * seen);
  st[0] = seen;
  reinterpret_cast<int32_t*>(st)[1] = redo;
  st[2] = eps_used;
  st[3] = any_bad ? INFINITY : e;
  st[4] = g;
  st[5] = calls;
  st[6] = redone;
  st[7] = grd_all;
  call[0] = 0u;
  if (state_host) {
    // two 16-byte stores into one 32-byte block of pinned host memory, the half with the call counter second: posted writes of one source to
    // one destination arrive in order, so no system-scope fence (a round trip over the link) stands between them; the host still reads the
    // counter before and after its snapshot
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v* sh = reinterpret_cast<f32x4v*>(state_host);
    __builtin_nontemporal_store(f32x4v{seen, __int_as_float(redo), eps_used, any_bad ? INFINITY : e}, &sh[0]);
    __builtin_nontemporal_store(f32x4v{g, calls, redone, grd_all}, &sh[1]);
  }
}

// ---- item-sharded top-k: message pack + merge (rails_amd/sharded.py) -----------------------------------------
// msg[row] = [k score words (fp32 bits in the low half of an int64) | k ids]; rows shorter than k are padded with
// (-inf, -1) so every rank contributes the same size to the single all-gather.
__global__ void pack_candidates_kernel(const float* __restrict__ scores, const int64_t* __restrict__ ids, int rows, int k_local,
                                       int k, int64_t* __restrict__ msg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * k) return;
  const int row = i / k, j = i - row * k;
  const float sc = j < k_local ? scores[(int64_t)row * k_local + j] : -INFINITY;
  msg[(int64_t)row * 2 * k + j] = (int64_t)(unsigned long long)__float_as_uint(sc);
  msg[(int64_t)row * 2 * k + k + j] = j < k_local ? ids[(int64_t)row * k_local + j] : -1;
}

// gathered: (R, rows, 2k) messages in rank order.  Candidate (r, j) gets position r*k + j: shard-major order is global
// position order for contiguous shards, so the tie rule (score desc, position asc) carries over and the merged result
// is bit-identical to the unsharded one.  One workgroup per row, R*k <= 16384 keys in LDS.
//
// Each rank's list arrives sorted (it is that rank's top-k), so no sort is needed: a key's rank in the merged order is
// the number of larger keys, = its index in its own list + one binary search in each other list (keys are distinct, so
// ranks are a permutation); keys of rank < k_out are written straight to their slot.  One barrier instead of the
// ~70 of a 2048-key bitonic sort (21 us -> 12 us at R = 8, k = 200).  Unsorted input (not produced by this library, but
// legal for the C entry point) is detected and takes the bitonic sort.
// f_invalid != NULL: the seen-id filter of the candidate index runs inside this launch over the k_out merged winners (staged in LDS,
// filter_from_lds) and f_k results per row are written -- the sharded counterpart of rails_topk_filtered.
// v.state != NULL (rails_merge_candidates_verdict): the messages are 2k + 2 wide -- [k score words | k ids | m | err] of rails_candidates_finish's
// sharded form -- and the row's verdict (merged k_out-th score - max over ranks of m > eps, no bad rank, guard) is folded into the call's.
struct MergeVerdictArgs {
  float default_eps, safety; const float* guard; int guard_per_row; float guard_limit;
  float* state; float* state_host; unsigned int* call;
};
__global__ __launch_bounds__(kSortThreads) void merge_candidates_kernel(const int64_t* __restrict__ gathered, int R, int rows,
                                                                       int k, int k_out, int npad,
                                                                       float* __restrict__ out_scores,
                                                                       int64_t* __restrict__ out_ids,
                                                                       const int64_t* __restrict__ f_invalid, int f_width, int f_k, int msg_ld,
                                                                       const MergeVerdictArgs v) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ int unsorted;
  __shared__ unsigned int s_kth, s_gbad, s_grd;
  __shared__ int s_last;
  __shared__ int64_t f_id[kFuseMaxK], f_inv[kFuseMaxW];
  __shared__ float f_sc[kFuseMaxK];
  __shared__ int f_scratch[kSortThreads / 64 + 2];
  const int row = blockIdx.x;
  const int count = R * k;
  const bool fuse = f_invalid != nullptr;
  if (fuse)
    for (int i = threadIdx.x; i < f_width; i += kSortThreads) f_inv[i] = f_invalid[(int64_t)row * f_width + i];   // visible after the barriers below
  if (threadIdx.x == 0) { unsorted = 0; s_kth = orderable(-INFINITY); s_gbad = 0u; s_grd = 0u; }
  for (int i = threadIdx.x; i < npad; i += kSortThreads) {
    unsigned long long kv = 0ull;
    if (i < count) {
      const int r = i / k, j = i - r * k;
      const unsigned int bits = (unsigned int)(unsigned long long)gathered[((int64_t)r * rows + row) * msg_ld + j];
      kv = ((unsigned long long)orderable(__uint_as_float(bits)) << 32) | (unsigned int)(~(unsigned int)i);
    }
    keys[i] = kv;
  }
  __syncthreads();
  auto emit = [&](unsigned long long kv, int slot) {
    const unsigned int pos = ~(unsigned int)(kv & 0xFFFFFFFFull);
    const int r = (int)(pos / (unsigned int)k), jj = (int)(pos - (unsigned int)r * (unsigned int)k);
    const float sc = unorderable((unsigned int)(kv >> 32));
    const int64_t id = gathered[((int64_t)r * rows + row) * msg_ld + k + jj];
    if (slot == k_out - 1) s_kth = (unsigned int)(kv >> 32);
    if (fuse) { f_sc[slot] = sc; f_id[slot] = id; }
    else { out_scores[(int64_t)row * k_out + slot] = sc; out_ids[(int64_t)row * k_out + slot] = id; }
  };
  auto finish = [&]() {
    if (v.state) {
      if (v.guard)
        for (int i = threadIdx.x; i < v.guard_per_row; i += kSortThreads) {
          const float g = fabsf(v.guard[(int64_t)row * v.guard_per_row + i]);
          if (!(g <= v.guard_limit)) atomicOr(&s_gbad, 1u);
          atomicMax(&s_grd, __float_as_uint(g == g ? g : INFINITY));
        }
      __syncthreads();
      float m = -INFINITY, err = 0.0f;
      int bad = s_gbad ? 1 : 0;
      for (int r = 0; r < R; ++r) {       // (every thread: a few broadcast loads)
        const int64_t* msg = gathered + ((int64_t)r * rows + row) * msg_ld;
        const float mr = __uint_as_float((unsigned int)(unsigned long long)msg[2 * k]), er = __uint_as_float((unsigned int)(unsigned long long)msg[2 * k + 1]);
        bad |= !(mr == mr) || !(er < INFINITY);
        m = fmaxf(m, mr == mr ? mr : INFINITY);
        err = fmaxf(err, er == er ? er : INFINITY);
      }
      const float kth = unorderable(s_kth);              // -inf when fewer than k_out real entries were merged
      const float gap = kth - m;
      const float eps = fmaxf(v.default_eps, v.safety * fmaxf(v.state[0], bad ? 0.0f : err));
      const int fail = bad || !(gap > eps);
      verdict_commit(v.call, row, rows, fail, bad, err, gap, __uint_as_float(s_grd), v.default_eps, v.safety, v.state, v.state_host, &s_last);
    }
    if (!fuse) return;
    __syncthreads();
    filter_from_lds<kSortThreads>(f_id, f_sc, k_out, f_inv, f_width, f_k, out_ids + (int64_t)row * f_k, out_scores + (int64_t)row * f_k, f_scratch);
  };
  bool bad = false;
  for (int i = threadIdx.x; i + 1 < count; i += kSortThreads)
    if ((i + 1) % k != 0 && keys[i] < keys[i + 1]) bad = true;
  if (bad) unsorted = 1;
  __syncthreads();
  if (!unsorted) {
    for (int i = threadIdx.x; i < count; i += kSortThreads) {
      const unsigned long long kv = keys[i];
      const int r = i / k, j = i - r * k;
      int rank = j;
      for (int o = 0; o < R && rank < k_out; ++o) {
        if (o == r) continue;
        const unsigned long long* list = keys + o * k;   // descending
        int lo = 0, hi = k;
        while (lo < hi) {                                // first index whose key is < kv  = #keys > kv
          const int mid = (lo + hi) >> 1;
          if (list[mid] > kv) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
      if (rank < k_out) emit(kv, rank);
    }
    finish();
    return;
  }
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (npad >> 1); t += kSortThreads) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
        const int hi2 = lo | stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = keys[lo], b = keys[hi2];
        if ((a < b) == desc) { keys[lo] = b; keys[hi2] = a; }
      }
      __syncthreads();
    }
  }
  for (int j = threadIdx.x; j < k_out; j += kSortThreads) emit(keys[j], j);
  finish();
}

int pack_candidates(const float* scores, const int64_t* ids, int rows, int k_local, int k, int64_t* msg, hipStream_t stream) {
  const int total = rows * k;
  if (total <= 0) return kOk;
  hipLaunchKernelGGL(pack_candidates_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, scores, ids, rows, k_local, k, msg);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int merge_candidates(const int64_t* gathered, int R, int rows, int k, int k_out, float* out_scores, int64_t* out_ids,
                     hipStream_t stream, const int64_t* f_invalid, int f_width, int f_k, const MergeVerdict* verdict) {
  if (rows <= 0 || k_out <= 0) return kOk;
  if (f_invalid && !(k_out <= kFuseMaxK && f_width >= 0 && f_width <= kFuseMaxW && f_k > 0 && f_k <= k_out)) {
    set_error("merge_candidates: the seen-id filter cannot be fused at k = %d, width = %d", k_out, f_width);
    return kErrUnsupported;
  }
  const int64_t count = (int64_t)R * k;
  if (count > kSortCap) { set_error("merge_candidates: R*k = %lld exceeds the in-LDS sort capacity (%d)", (long long)count, kSortCap); return kErrUnsupported; }
  if (ensure_sort_lds() != kOk) return kErrLaunch;
  static DynLdsOnce once;
  if (ensure_dyn_lds(once, reinterpret_cast<const void*>(&merge_candidates_kernel), kSortCap * (int)sizeof(unsigned long long)) != kOk)
    return kErrLaunch;
  const int npad = next_pow2((int)count < 2 ? 2 : (int)count);
  MergeVerdictArgs v{};
  int msg_ld = 2 * k;
  if (verdict) {
    v.default_eps = verdict->default_eps; v.safety = verdict->safety; v.guard = verdict->guard; v.guard_per_row = verdict->guard ? verdict->guard_per_row : 0;
    v.guard_limit = verdict->guard_limit; v.state = verdict->state; v.state_host = verdict->state_host; v.call = verdict->call;
    msg_ld = 2 * k + 2;
  }
  hipLaunchKernelGGL(merge_candidates_kernel, dim3(rows), dim3(kSortThreads), npad * sizeof(unsigned long long), stream, gathered, R,
                     rows, k, k_out, npad, out_scores, out_ids, f_invalid, f_width, f_k, msg_ld, v);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- precision "f16x3-exact": the verified finish of a speculative brute-force top-k -------------------------------------
// Per row: kc candidates with their exact fp32 logits `exact`, the f16x3 logits `approx` that selected them (same order) and
// their corpus positions.  One workgroup sorts the row by (exact desc, position asc) -- the dense fp32 path's total order -- in
// LDS, emits the top k (scores, ids[position]) and the row's verdict:
//   ok[row] = (k-th exact score > min approx + margin_eps)  and  (max |exact - approx| <= check_eps)
// The first clause means no item outside the candidates (approx <= min approx, exact <= approx + eps) can reach the k-th place;
// the second monitors the error bound eps on the candidates themselves.  NaNs fail both.  (topk_modules._forward_rescored)
// one_sided: approx is an upper bound of the exact score (rails_mol_score_dense_upper); the monitored quantity is exact - approx (<= 0 when
// the bound holds; the stat is max(0, .)) and margin_eps = 0 proves the row.
// Entries [n_ranked, kc) are PROBES: items drawn at random from the whole corpus, re-scored with the candidates.  They only feed
// the error monitor (their approximate logit is read from the dense matrix at their position); they take no part in the selection.
__global__ __launch_bounds__(kSortThreads) void rescore_select_kernel(const float* __restrict__ exact, int64_t ld,
                                                                      const float* __restrict__ approx,
                                                                      const float* __restrict__ approx_dense, int64_t ld_dense,
                                                                      const int64_t* __restrict__ positions,
                                                                      const int64_t* __restrict__ ids, int n_ranked, int kc, int k, int npad,
                                                                      float margin_eps, float check_eps, int one_sided, float* __restrict__ out_scores,
                                                                      int64_t* __restrict__ out_ids, int* __restrict__ ok,
                                                                      float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ float red_min[kSortThreads / 64], red_err[kSortThreads / 64];
  __shared__ int red_nan[kSortThreads / 64];
  const int row = blockIdx.x;
  float mn = INFINITY, err = 0.0f;
  int bad = 0;
  for (int i = threadIdx.x; i < npad; i += kSortThreads) {
    unsigned long long kv = 0ull;
    if (i < kc) {
      const float e = exact[(int64_t)row * ld + i];
      const int64_t pos = positions[(int64_t)row * kc + i];
      float a;
      if (i < n_ranked) {
        a = approx[(int64_t)row * n_ranked + i];
        kv = ((unsigned long long)orderable(e) << 32) | (unsigned int)(~(unsigned int)pos);
        mn = fminf(mn, a);
      } else {
        a = approx_dense[(int64_t)row * ld_dense + pos];
      }
      const float dd = one_sided ? e - a : fabsf(e - a);   // one_sided: `approx` is an UPPER BOUND of the exact score; only exact > approx is an error
      bad |= !(dd <= check_eps);       // catches NaN as well
      err = fmaxf(err, dd == dd ? dd : INFINITY);
    }
    keys[i] = kv;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    err = fmaxf(err, __shfl_xor(err, o, 64));
    bad |= __shfl_xor(bad, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { red_min[threadIdx.x >> 6] = mn; red_err[threadIdx.x >> 6] = err; red_nan[threadIdx.x >> 6] = bad; }
  __syncthreads();
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (npad >> 1); t += kSortThreads) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
        const int hi2 = lo | stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = keys[lo], b = keys[hi2];
        if ((a < b) == desc) { keys[lo] = b; keys[hi2] = a; }
      }
      __syncthreads();
    }
  }
  for (int j = threadIdx.x; j < k; j += kSortThreads) {
    const unsigned long long kv = keys[j];
    const int64_t pos = (int64_t)(~(unsigned int)(kv & 0xFFFFFFFFull));
    out_scores[(int64_t)row * k + j] = unorderable((unsigned int)(kv >> 32));
    out_ids[(int64_t)row * k + j] = ids ? ids[pos] : pos;
  }
  if (threadIdx.x == 0) {
    float m = INFINITY, er = 0.0f;
    int b = 0;
    for (int w = 0; w < kSortThreads / 64; ++w) { m = fminf(m, red_min[w]); er = fmaxf(er, red_err[w]); b |= red_nan[w]; }
    const float kth = unorderable((unsigned int)(keys[k - 1] >> 32));
    if (ok) ok[row] = (!b && kth > m + margin_eps) ? 1 : 0;
    if (stats) { stats[2 * row] = er; stats[2 * row + 1] = kth - m; }   // largest |exact - approx| seen, and the margin the row has
  }
}

// Verdict of a speculative call on the device (the host used to read the per-row stats and decide: ~100 us of GPU idle per call).
// state[0] largest |first pass - fp32| ever seen (in/out)   state[1] REDO flag as int32 (out; the fallback's launch predicate)
// state[2] eps used   state[3] this call's largest error   state[4] this call's smallest margin   state[5] calls   state[6] redone calls
// state[7] largest |guard value| ever seen
// guard (optional): `guard_count` floats whose magnitudes must stay <= guard_limit for the caller's bound on |first pass - fp32| to
// hold (the proved mode passes the batch's gq' rows: rails_amd/f16x3_bound.py); a larger one, or a NaN, raises REDO like a failed margin.
__global__ __launch_bounds__(256) void rescore_verdict_kernel(const float* __restrict__ row_stats, int rows, float default_eps, float safety,
                                                              const float* __restrict__ guard, int64_t guard_count, float guard_limit,
                                                              float* __restrict__ state) {
  __shared__ float s_err[256], s_gap[256], s_grd[256];
  __shared__ int s_bad[256];
  float err = 0.0f, gap = INFINITY, grd = 0.0f;
  int bad = 0;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float e = row_stats[2 * r], g = row_stats[2 * r + 1];
    if (!(e == e) || !(g == g)) bad = 1;        // NaN anywhere: the call is redone
    err = fmaxf(err, e);
    gap = fminf(gap, g);
  }
  for (int64_t i = threadIdx.x; i < guard_count; i += 256) {
    const float v = fabsf(guard[i]);
    if (!(v <= guard_limit)) bad = 1;           // too large, or NaN
    grd = fmaxf(grd, v == v ? v : INFINITY);
  }
  s_err[threadIdx.x] = err; s_gap[threadIdx.x] = gap; s_bad[threadIdx.x] = bad; s_grd[threadIdx.x] = grd;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      s_err[threadIdx.x] = fmaxf(s_err[threadIdx.x], s_err[threadIdx.x + w]);
      s_gap[threadIdx.x] = fminf(s_gap[threadIdx.x], s_gap[threadIdx.x + w]);
      s_grd[threadIdx.x] = fmaxf(s_grd[threadIdx.x], s_grd[threadIdx.x + w]);
      s_bad[threadIdx.x] |= s_bad[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    err = s_err[0]; gap = s_gap[0]; bad = s_bad[0] || !(err < INFINITY);
    float seen = state[0];
    if (!bad) seen = fmaxf(seen, err);          // never forgotten
    const float eps = fmaxf(default_eps, safety * seen);
    const int redo = bad || !(gap > eps);
    state[0] = seen;
    reinterpret_cast<int32_t*>(state)[1] = redo;
    state[2] = eps;
    state[3] = bad ? INFINITY : err;
    state[4] = gap;
    state[5] += 1.0f;
    if (redo) state[6] += 1.0f;
    state[7] = fmaxf(state[7], s_grd[0]);
  }
}

int rescore_verdict(const float* row_stats, int rows, float default_eps, float safety, const float* guard, int64_t guard_count, float guard_limit,
                    float* state, hipStream_t stream) {
  hipLaunchKernelGGL(rescore_verdict_kernel, dim3(1), dim3(256), 0, stream, row_stats, rows, default_eps, safety, guard, guard ? guard_count : 0,
                     guard_limit, state);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// row_stats of a GLOBAL verdict (item-sharded proved top-k, rails_amd/sharded.py): row_stats[row] = [err_max[0], kth[row * ld + col] - m_max[row]]
// -- the largest |first pass - fp32| any rank saw on its candidates, and the margin between the merged k-th fp32 score and the best
// first-pass score any rank left outside its candidates (both all-reduced by the caller).  -inf margins (m = +inf) and NaNs pass through.
__global__ void margin_stats_kernel(const float* __restrict__ kth, int64_t ld, int col, const float* __restrict__ m_max, const float* __restrict__ err_max,
                                    int rows, float* __restrict__ row_stats) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  row_stats[2 * r] = err_max[0];
  row_stats[2 * r + 1] = kth[(int64_t)r * ld + col] - m_max[r];
}
int margin_stats(const float* kth, int64_t ld, int col, const float* m_max, const float* err_max, int rows, float* row_stats, hipStream_t stream) {
  if (rows <= 0) return kOk;
  hipLaunchKernelGGL(margin_stats_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, kth, ld, col, m_max, err_max, rows, row_stats);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int rescore_select(const float* exact, int64_t ld, const float* approx, const float* approx_dense, int64_t ld_dense, const int64_t* positions,
                   const int64_t* ids, int rows, int n_ranked, int kc, int k, float margin_eps, float check_eps, int one_sided, float* out_scores,
                   int64_t* out_ids, int* ok, float* stats, hipStream_t stream) {
  if (rows <= 0) return kOk;
  if (kc > kSortCap) { set_error("rescore_select: %d candidates exceed the in-LDS sort capacity (%d)", kc, kSortCap); return kErrUnsupported; }
  static DynLdsOnce once;
  if (ensure_dyn_lds(once, reinterpret_cast<const void*>(&rescore_select_kernel), kSortCap * (int)sizeof(unsigned long long)) != kOk)
    return kErrLaunch;
  const int npad = next_pow2(kc < 2 ? 2 : kc);
  hipLaunchKernelGGL(rescore_select_kernel, dim3(rows), dim3(kSortThreads), npad * sizeof(unsigned long long), stream, exact, ld, approx,
                     approx_dense, ld_dense, positions, ids, n_ranked, kc, k, npad, margin_eps, check_eps, one_sided, out_scores, out_ids, ok, stats);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- candidate-union helpers of MoLNaiveTopK / MoLCombTopK ------------------------------------------------
// torch.sort(cat(all_indices), dim=1) (mol_top_k.py:257, :515): ascending LDS bitonic sort of each row of int64.
__global__ __launch_bounds__(kSortThreads) void sort_rows_i64_kernel(const int64_t* __restrict__ in, int n, int npad,
                                                                    int64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const int row = blockIdx.x;
  if (npad >= 2 * kSortThreads && npad <= 16 * kSortThreads) {
    // the register / shuffle sort (descending) on complemented keys = ascending order of the values: 73 -> ~35 us for 32 rows of 6 400
    for (int i = threadIdx.x; i < npad; i += kSortThreads)
      keys[i] = i < n ? ~((unsigned long long)in[(int64_t)row * n + i] ^ 0x8000000000000000ull) : 0ull;      // padding sorts last
    __syncthreads();
    if (npad == 2 * kSortThreads) block_sort_desc_multi<2>(keys);
    else if (npad == 4 * kSortThreads) block_sort_desc_multi<4>(keys);
    else if (npad == 8 * kSortThreads) block_sort_desc_multi<8>(keys);
    else block_sort_desc_multi<16>(keys);
    for (int i = threadIdx.x; i < n; i += kSortThreads) out[(int64_t)row * n + i] = (int64_t)(~keys[i] ^ 0x8000000000000000ull);
    return;
  }
  for (int i = threadIdx.x; i < npad; i += kSortThreads)
    keys[i] = i < n ? ((unsigned long long)in[(int64_t)row * n + i] ^ 0x8000000000000000ull) : ~0ull;  // signed order
  __syncthreads();
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (npad >> 1); t += kSortThreads) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
        const int hi2 = lo | stride;
        const bool asc = ((lo & size) == 0);
        const unsigned long long a = keys[lo], b = keys[hi2];
        if ((a > b) == asc) { keys[lo] = b; keys[hi2] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += kSortThreads) out[(int64_t)row * n + i] = (int64_t)(keys[i] ^ 0x8000000000000000ull);
}

// candidate_scores = where(idx[j] != idx[j-1] or j == 0, scores, fill)  (mol_top_k.py:277-284, :535-542)
__global__ void mask_sorted_duplicates_kernel(const int64_t* __restrict__ idx, float* __restrict__ scores, int64_t ld,
                                              int rows, int n, float fill) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * n) return;
  const int row = (int)(i / n), j = (int)(i - (int64_t)row * n);
  if (j > 0 && idx[i] == idx[i - 1]) scores[(int64_t)row * ld + j] = fill;
}

int sort_rows_i64(const int64_t* in, int rows, int n, int64_t* out, hipStream_t stream) {
  if (rows <= 0 || n <= 0) return kOk;
  if (n > kSortCap) { set_error("sort_rows_i64: n = %d exceeds the in-LDS sort capacity (%d)", n, kSortCap); return kErrUnsupported; }
  static DynLdsOnce once;
  if (ensure_dyn_lds(once, reinterpret_cast<const void*>(&sort_rows_i64_kernel), kSortCap * (int)sizeof(unsigned long long)) != kOk)
    return kErrLaunch;
  const int npad = next_pow2(n < 2 ? 2 : n);
  hipLaunchKernelGGL(sort_rows_i64_kernel, dim3(rows), dim3(kSortThreads), npad * sizeof(unsigned long long), stream, in, n, npad, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int mask_sorted_duplicates(const int64_t* idx, float* scores, int64_t ld, int rows, int n, float fill, hipStream_t stream) {
  const int64_t total = (int64_t)rows * n;
  if (total <= 0) return kOk;
  hipLaunchKernelGGL(mask_sorted_duplicates_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, idx, scores, ld, rows, n, fill);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- the tail of a candidate rerank without the integer sort (round 6) ------------------------------------------------------------------
// get_top_k_outputs over MoLNaiveTopK / MoLCombTopK wants the first k unseen of the candidates ranked by (score desc, position asc) with
// duplicates dropped.  The module's own forward() sorts the candidate positions (so that duplicates are neighbours and the column order IS the
// position order: 72 us per 32 rows of 6 400), scores, masks, ranks.  Here the candidates stay in the order the scans left them:
//   rerank_keys_kernel  one workgroup per row: a 16 384-slot LDS hash set of the row's positions decides which copy of a position is its
//                       first; that copy's key is (score, ~position) -- the order of the sorted form, whatever the column -- the others' 0
//                       (an empty key, below everything); rows with fewer than `min_unique` distinct positions raise *flag: an empty key
//                       could then reach the result, and the caller redoes the call on the sorted form
//   row_select_kernel<., KEYS>  top-k' of the keys, ids by position, the seen-id filter inside
// Same (ids, scores) as sort -> score -> mask -> rails_topk_candidates_filtered whenever the flag stays 0.
constexpr int kRerankHash = 16384;
constexpr int kRerankMax = 8192;

__global__ __launch_bounds__(1024) void rerank_keys_kernel(const float* __restrict__ scores, int64_t ld, const int64_t* __restrict__ positions, int n,
                                                          int min_unique, unsigned long long* __restrict__ keys, int32_t* __restrict__ flag) {
  extern __shared__ unsigned int rr_tab[];      // kRerankHash slots + the row's distinct count
  const int row = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i <= kRerankHash; i += 1024) rr_tab[i] = i < kRerankHash ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  unsigned int uniq = 0u;
  for (int c = tid; c < n; c += 1024) {
    const unsigned int p = (unsigned int)positions[(int64_t)row * n + c];
    unsigned int h = (p * 2654435761u) >> 18;
    bool first;
    for (;;) {
      const unsigned int old = atomicCAS(&rr_tab[h], 0xFFFFFFFFu, p);
      if (old == 0xFFFFFFFFu) { first = true; break; }
      if (old == p) { first = false; break; }
      h = (h + 1u) & (unsigned int)(kRerankHash - 1);
    }
    keys[(int64_t)row * n + c] = first ? make_key(scores[(int64_t)row * ld + c], p) : 0ull;
    uniq += first ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) uniq += (unsigned int)__shfl_xor((int)uniq, o, 64);
  if ((tid & 63) == 0 && uniq) atomicAdd(&rr_tab[kRerankHash], uniq);
  __syncthreads();
  if (tid == 0 && rr_tab[kRerankHash] < (unsigned int)min_unique) *flag = 1;     // every writer stores the same value
}

size_t rerank_workspace_bytes(int rows, int n_cand) { return sizeof(unsigned long long) * (size_t)rows * (size_t)n_cand; }

int rerank_topk_filtered(const float* scores, int64_t ld, int rows, int n_cand, int k_prime, const int64_t* positions, const int64_t* ids,
                         const int64_t* invalid, int width, int k, void* ws, size_t ws_bytes, int64_t* out_ids, float* out_scores, int32_t* flag,
                         hipStream_t stream) {
  if (rows <= 0 || k <= 0) return kOk;
  if (n_cand <= 1024 || n_cand > kRerankMax || !topk_can_fuse_filter(n_cand, k_prime, width, k) || k_prime > n_cand) {
    set_error("rerank_topk_filtered: unsupported size (n_cand = %d, k' = %d, width = %d, k = %d)", n_cand, k_prime, width, k);
    return kErrUnsupported;
  }
  if (ws_bytes < rerank_workspace_bytes(rows, n_cand)) { set_error("rerank_topk_filtered: workspace too small"); return kErrNoMem; }
  if (ensure_sort_lds() != kOk) return kErrLaunch;
  static DynLdsOnce once;
  const int lds = (kRerankHash + 16) * (int)sizeof(unsigned int);
  if (ensure_dyn_lds(once, reinterpret_cast<const void*>(&rerank_keys_kernel), lds) != kOk) return kErrLaunch;
  unsigned long long* keys = static_cast<unsigned long long*>(ws);
  hipLaunchKernelGGL(rerank_keys_kernel, dim3(rows), dim3(1024), lds, stream, scores, ld, positions, n_cand, k_prime, keys, flag);
  if (hipGetLastError() != hipSuccess) return kErrLaunch;
  RowSelectArgs b{};
  b.keys_in = keys; b.keys_per_row = n_cand; b.k = k_prime;
  b.ids = ids; b.ids_row_stride = 0; b.out_scores = out_scores; b.out_ids = out_ids;
  b.f_invalid = invalid; b.f_width = width; b.f_k = k;
  return launch_row_select<true>(b, rows, 1, n_cand, stream);
}

int filter_seen(const int64_t* top_ids, const float* top_scores, int rows, int k_prime, const int64_t* invalid,
                int width, int k, int64_t* out_ids, float* out_scores, hipStream_t stream) {
  if (rows <= 0) return kOk;
  if (k > k_prime) { set_error("seen-id filter: k (%d) > k' (%d)", k, k_prime); return kErrInvalid; }
  const size_t lds = sizeof(int64_t) * (size_t)width + (size_t)k_prime + 16;
  if (lds > 60000) { set_error("seen-id filter: k' or width too large for LDS"); return kErrUnsupported; }
  if (k_prime > 1024)
    hipLaunchKernelGGL(filter_seen_kernel<1024>, dim3(rows), dim3(1024), lds, stream, top_ids, top_scores, k_prime, invalid, width, k, out_ids, out_scores);
  else
    hipLaunchKernelGGL(filter_seen_kernel<256>, dim3(rows), dim3(256), lds, stream, top_ids, top_scores, k_prime, invalid, width, k, out_ids, out_scores);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- candidates of the proved exact top-k: threshold selection + fused verification (round 6) --------------------------------------
// The proved flow (rails_amd/topk_modules.py) does not need the EXACT kc best first-pass scores of a row: it needs a
// candidate set C and a value m with "every item outside C has a first-pass score <= m".  Any threshold does: with a monotone bin
// function b(s), C = {x : b(s_x) >= b_t} and m = min over C of s (an item outside has b(s) < b_t <= b(s_c), hence s < s_c for every
// candidate c).  So the selection is ONE histogram pass (4 096 linear bins over the a-priori range of the logits, |s| <= 1/tau + the
// per-pair bound) that finds, per row, the lowest bin b_t with #{b(s) >= b_t} <= cap, and ONE compaction pass -- two launches that read
// the (B, N) first-pass matrix twice, where the exact radix selection took a memset + 3 histogram + 3 pick + tie + compact + sort
// launches and five reads (85 us -> ~25 us at 32 x 695 762).  Rows of up to kCandSingleMax scores take both passes in one launch
// (one workgroup per row, LDS histogram, the second walk hits L2).
//   rails_candidates_select   -> positions (rows, cap) int64, first-pass scores (rows, cap), counts[row] <= cap (in the workspace)
//   rails_mol_score_indexed_rows(..., counts)   fp32 logits of the first counts[row] candidates of every row
//   rails_candidates_finish   one workgroup per row: sort by (fp32 score desc, position asc), top-k (+ the seen-id filter of the candidate
//                             index), the row's verdict, and -- last workgroup done -- the call's verdict / calibration state, written to the
//                             device state AND straight into pinned host memory (no copy launch); leaves the workspace zeroed for the next call.
// Workspace (rails_candidates_workspace_bytes, zeroed ONCE by the caller; every call restores the zeros):
//   counts[rows] | flags[rows] | call[8 + 4 rows] (arrival counter, the rows' verdict words) | coarse[rows][64] | fine[rows][4096]
constexpr int kCandBins = 4096, kCandCoarse = 64, kCandPer = kCandBins / kCandCoarse;
constexpr int kCandThreads = 512;
constexpr int kCandSingleMax = 65536;      // rows up to this many scores: one launch, one workgroup per row
constexpr int kCandStage = 2048;

struct CandWs { unsigned int* counts; unsigned int* flags; unsigned int* call; unsigned int* coarse; unsigned int* fine; };
static size_t cand_rows_pad(int rows) { return ((size_t)rows + 3) / 4 * 4; }     // keeps the 16-byte words of the call block aligned
static size_t cand_ws_words(int rows) { return cand_rows_pad(rows) * 2 + 8 + (size_t)rows * 4 + (size_t)rows * kCandCoarse + (size_t)rows * kCandBins; }
static CandWs cand_ws(void* ws, int rows) {
  unsigned int* w = static_cast<unsigned int*>(ws);
  const size_t rp = cand_rows_pad(rows);
  CandWs c;
  c.counts = w; c.flags = w + rp; c.call = w + 2 * rp; c.coarse = c.call + 8 + 4 * (size_t)rows; c.fine = c.coarse + (size_t)rows * kCandCoarse;
  return c;
}
size_t candidates_workspace_bytes(int rows) { return cand_ws_words(rows < 1 ? 1 : rows) * sizeof(unsigned int); }

// monotone non-decreasing in s (round-to-nearest subtraction and multiplication by a positive scale, clamp, truncation of a non-negative
// value); a NaN lands in bin 0 and raises the row's flag
__device__ __forceinline__ int cand_bin(float s, float lo, float scale) {
  float x = (s - lo) * scale;
  x = fminf(fmaxf(x, 0.0f), (float)(kCandBins - 1));
  return (int)x;
}

// Walk [begin, end) of a row, four 16-byte loads in flight per thread: f4(x, i) once per aligned group of four scores (positions i .. i + 3),
// f1(score, i) for the unaligned head and tail.
template <int NT, class F4, class F1>
__device__ __forceinline__ void walk_chunk4(const float* __restrict__ rowp, int64_t begin, int64_t end, F4 f4, F1 f1) {
  int64_t a0 = begin + ((4 - (int64_t)((reinterpret_cast<uintptr_t>(rowp + begin) >> 2) & 3)) & 3);
  if (a0 > end) a0 = end;
  const int64_t nvec = (end - a0) >> 2;
  for (int64_t i = begin + threadIdx.x; i < a0; i += NT) f1(rowp[i], i);
  const float4* body = reinterpret_cast<const float4*>(rowp + a0);
  int64_t v = threadIdx.x;
  for (; v + 3 * NT < nvec; v += 4 * NT) {
    const float4 x0 = body[v], x1 = body[v + NT], x2 = body[v + 2 * NT], x3 = body[v + 3 * NT];
    f4(x0, a0 + 4 * v); f4(x1, a0 + 4 * (v + NT)); f4(x2, a0 + 4 * (v + 2 * NT)); f4(x3, a0 + 4 * (v + 3 * NT));
  }
  for (; v < nvec; v += NT) f4(body[v], a0 + 4 * v);
  for (int64_t i = a0 + 4 * nvec + threadIdx.x; i < end; i += NT) f1(rowp[i], i);
}

// the smallest float whose bin is >= bt (cand_bin is monotone): a score is selected iff it is >= this value -- one compare per score in the
// compaction walk instead of the bin arithmetic.  bt <= 0: -inf (everything); no such float: NaN (nothing compares >= NaN).
__device__ __forceinline__ float cand_threshold_value(int bt, float lo, float scale) {
  if (bt <= 0) return -INFINITY;
  if (cand_bin(INFINITY, lo, scale) < bt) return __uint_as_float(0x7FC00000u);
  if (cand_bin(-INFINITY, lo, scale) >= bt) return -INFINITY;
  unsigned int a = orderable(-INFINITY), b = orderable(INFINITY);       // orderable keys of finite floats lie between: bin(a) < bt <= bin(b)
  while (b - a > 1u) {
    const unsigned int mid = a + ((b - a) >> 1);
    if (cand_bin(unorderable(mid), lo, scale) >= bt) b = mid; else a = mid;
  }
  return unorderable(b);
}

// LDS histogram h[kCandBins] of a workgroup -> coarse sums hc[kCandCoarse] (every thread sums a run of fine bins, runs of one coarse bin sit in
// neighbouring lanes).  NT * run = kCandBins.
template <int NT>
__device__ __forceinline__ void cand_coarse_sums(const unsigned int* h, unsigned int* hc) {
  constexpr int run = kCandBins / NT;              // 8 (512 threads) or 4 (1024)
  constexpr int per = kCandPer / run;              // threads per coarse bin: 8 or 16
  unsigned int s = 0u;
#pragma unroll
  for (int j = 0; j < run; ++j) s += h[threadIdx.x * run + j];
#pragma unroll
  for (int o = 1; o < per; o <<= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & (per - 1)) == 0) hc[threadIdx.x / per] = s;
}

// One wave: the threshold bin of a row from its coarse / fine histograms (global or LDS): the LOWEST fine bin b_t whose count of scores in
// bins >= b_t is <= cap (0 when the whole row fits; kCandBins -- nothing selected -- when the top fine bin alone holds more than cap).
__device__ __forceinline__ int cand_pick(const unsigned int* coarse, const unsigned int* fine, unsigned int cap, int lane) {
  const unsigned int cc = coarse[kCandCoarse - 1 - lane];
  unsigned int incl = cc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
  const unsigned long long over = __ballot(incl > cap);
  if (over == 0ull) return 0;
  const int first = __ffsll((long long)over) - 1;
  const int cg = kCandCoarse - 1 - first;
  const unsigned int base = (unsigned int)__shfl((int)(incl - cc), first, 64);     // scores in the coarse bins above cg
  const unsigned int f = fine[cg * kCandPer + kCandPer - 1 - lane];
  unsigned int incl2 = f;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned int v = __shfl_up(incl2, o, 64); if (lane >= o) incl2 += v; }
  const int fit = __popcll(__ballot(base + incl2 <= cap));                          // monotone in the lane: the top `fit` fine bins of cg fit
  return cg * kCandPer + kCandPer - fit;                                            // fit == 0: the first bin above cg
}

// the selected scores of [begin, end) of a row -> the row's candidate list: staged in LDS (one LDS atomic per wave that holds one), ONE
// global atomic per workgroup reserves their range of the list (compact_kernel's scheme); a chunk with more selected scores than the
// stage holds is walked again and placed directly.  A score is selected iff it is >= thr (cand_threshold_value of the row's threshold bin).
template <int NT>
__device__ __forceinline__ void cand_append(const float* __restrict__ rowp, int64_t begin, int64_t end, float thr,
                                            unsigned int* count, int64_t* __restrict__ out_pos, float* __restrict__ out_a, unsigned int cap,
                                            unsigned long long* stage, unsigned int* wg_base, unsigned int* wg_cursor) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) *wg_cursor = 0u;
  __syncthreads();
  auto put = [&](bool sel, float sc, int64_t i, unsigned int base_all, bool direct) {   // wave-uniform call; appends the lanes with sel
    const unsigned long long m = __ballot(sel);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      unsigned int base = 0;
      if (lane == leader) base = atomicAdd(wg_cursor, (unsigned int)__popcll(m));
      base = (unsigned int)__shfl((int)base, leader, 64);
      if (sel) {
        const unsigned int slot = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        if (!direct) { if (slot < (unsigned int)kCandStage) stage[slot] = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned int)i; }
        else if (base_all + slot < cap) { out_pos[base_all + slot] = i; out_a[base_all + slot] = sc; }
      }
    }
  };
  auto walk = [&](unsigned int base_all, bool direct) {
    walk_chunk4<NT>(rowp, begin, end,
        [&](const float4 x, int64_t i) {
          const bool s0 = x.x >= thr, s1 = x.y >= thr, s2 = x.z >= thr, s3 = x.w >= thr;
          if (__ballot(s0 | s1 | s2 | s3) != 0ull) {       // rare: ~cap of the row's scores are selected
            put(s0, x.x, i, base_all, direct); put(s1, x.y, i + 1, base_all, direct); put(s2, x.z, i + 2, base_all, direct); put(s3, x.w, i + 3, base_all, direct);
          }
        },
        [&](float sc, int64_t i) { put(sc >= thr, sc, i, base_all, direct); });
  };
  walk(0u, false);
  __syncthreads();
  const unsigned int total = *wg_cursor;
  if (total == 0u) return;
  __syncthreads();
  if (threadIdx.x == 0) { *wg_base = atomicAdd(count, total); *wg_cursor = 0u; }
  __syncthreads();
  const unsigned int wbase = *wg_base;
  if (total <= (unsigned int)kCandStage) {
    for (unsigned int j = threadIdx.x; j < total; j += NT) {
      const unsigned long long e = stage[j];
      if (wbase + j < cap) { out_pos[wbase + j] = (int64_t)(unsigned int)e; out_a[wbase + j] = __uint_as_float((unsigned int)(e >> 32)); }
    }
    return;
  }
  walk(wbase, true);
}

__global__ __launch_bounds__(kCandThreads) void cand_hist_kernel(const float* __restrict__ scores, int64_t ld, int64_t n, int64_t chunk, float lo,
                                                                float scale, unsigned int cap, CandWs w) {
  __shared__ __attribute__((aligned(16))) unsigned int h[kCandBins];
  __shared__ unsigned int hc[kCandCoarse];
  __shared__ int s_cut;
  const int row = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < kCandBins; i += kCandThreads) h[i] = 0u;
  __syncthreads();
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = (begin + chunk < n) ? begin + chunk : n;
  const float* rowp = scores + (int64_t)row * ld;
  bool nan = false;
  auto one = [&](float sc, int64_t) { nan |= !(sc == sc); atomicAdd(&h[cand_bin(sc, lo, scale)], 1u); };
  walk_chunk4<kCandThreads>(rowp, begin, end, [&](const float4 x, int64_t i) { one(x.x, i); one(x.y, i); one(x.z, i); one(x.w, i); }, one);
  if (__ballot(nan) != 0ull && lane == 0) atomicOr(&w.flags[row], 1u);
  __syncthreads();
  cand_coarse_sums<kCandThreads>(h, hc);
  __syncthreads();
  // Only the bins that can hold the row's threshold are merged: with c the coarse bin in which THIS chunk's count from the top passes cap, the
  // row's count passes cap in c or above, so its threshold bin lies in a coarse bin >= c -- and every workgroup merges all of its bins >= its c.
  if (tid < 64) {
    const unsigned int cc = hc[kCandCoarse - 1 - lane];
    unsigned int incl = cc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    const unsigned long long over = __ballot(incl > cap);
    if (lane == 0) s_cut = over ? kCandCoarse - 1 - (__ffsll((long long)over) - 1) : 0;
  }
  __syncthreads();
  const int cut = s_cut;
  unsigned int* gf = w.fine + (size_t)row * kCandBins;
  unsigned int* gc = w.coarse + (size_t)row * kCandCoarse;
  for (int i = cut * kCandPer + tid; i < kCandBins; i += kCandThreads)
    if (h[i]) atomicAdd(&gf[i], h[i]);
  if (tid < kCandCoarse && tid >= cut && hc[tid]) atomicAdd(&gc[tid], hc[tid]);
}

__global__ __launch_bounds__(kCandThreads) void cand_compact_kernel(const float* __restrict__ scores, int64_t ld, int64_t n, int64_t chunk, float lo,
                                                                   float scale, unsigned int cap, CandWs w, int64_t* __restrict__ out_pos,
                                                                   float* __restrict__ out_a, int64_t cand_ld) {
  __shared__ unsigned long long stage[kCandStage];
  __shared__ unsigned int wg_base, wg_cursor;
  __shared__ float s_thr;
  const int row = blockIdx.y, tid = threadIdx.x;
  if (tid < 64) {
    const int bt = cand_pick(w.coarse + (size_t)row * kCandCoarse, w.fine + (size_t)row * kCandBins, cap, tid);
    if (tid == 0) s_thr = cand_threshold_value(bt, lo, scale);
  }
  __syncthreads();
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = (begin + chunk < n) ? begin + chunk : n;
  cand_append<kCandThreads>(scores + (int64_t)row * ld, begin, end, s_thr, &w.counts[row], out_pos + (int64_t)row * cand_ld,
                            out_a + (int64_t)row * cand_ld, cap, stage, &wg_base, &wg_cursor);
}

// rows of <= kCandSingleMax scores: histogram, threshold and compaction in one launch (the second walk of the row hits L2)
__global__ __launch_bounds__(kRowThreads) void cand_single_kernel(const float* __restrict__ scores, int64_t ld, int64_t n, float lo, float scale,
                                                                 unsigned int cap, CandWs w, int64_t* __restrict__ out_pos, float* __restrict__ out_a,
                                                                 int64_t cand_ld) {
  __shared__ __attribute__((aligned(16))) unsigned int h[kCandBins];
  __shared__ unsigned int hc[kCandCoarse];
  __shared__ unsigned long long stage[kCandStage];
  __shared__ unsigned int wg_base, wg_cursor;
  __shared__ float s_thr;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < kCandBins; i += kRowThreads) h[i] = 0u;
  __syncthreads();
  const float* rowp = scores + (int64_t)row * ld;
  bool nan = false;
  auto one = [&](float sc, int64_t) { nan |= !(sc == sc); atomicAdd(&h[cand_bin(sc, lo, scale)], 1u); };
  walk_chunk4<kRowThreads>(rowp, 0, n, [&](const float4 x, int64_t i) { one(x.x, i); one(x.y, i); one(x.z, i); one(x.w, i); }, one);
  if (__ballot(nan) != 0ull && lane == 0) atomicOr(&w.flags[row], 1u);
  __syncthreads();
  cand_coarse_sums<kRowThreads>(h, hc);
  __syncthreads();
  if (tid < 64) {
    const int bt = cand_pick(hc, h, cap, tid);
    if (tid == 0) s_thr = cand_threshold_value(bt, lo, scale);
  }
  __syncthreads();
  cand_append<kRowThreads>(rowp, 0, n, s_thr, &w.counts[row], out_pos + (int64_t)row * cand_ld, out_a + (int64_t)row * cand_ld, cap, stage,
                           &wg_base, &wg_cursor);
}

static int cand_single_max() {
  static const int v = [] { const char* e = getenv("RAILS_CAND_SINGLE_MAX"); const int x = e ? atoi(e) : -1; return x >= 0 ? x : kCandSingleMax; }();
  return v;
}

int candidates_select(const float* scores, int64_t ld, int rows, int64_t n, int cap, float lo, float hi, void* ws, int64_t* out_pos, float* out_approx,
                      int64_t cand_ld, int n_cu, hipStream_t stream) {
  if (rows <= 0 || n <= 0) return kOk;
  if (cap < 1 || cap > kSortCap || cand_ld < cap) { set_error("candidates_select: cap = %d out of range (1 .. %d, row stride %lld)", cap, kSortCap, (long long)cand_ld); return kErrInvalid; }
  if (n >= (1ll << 32)) { set_error("candidates_select: n = %lld does not fit 32-bit positions", (long long)n); return kErrUnsupported; }
  if (!(hi > lo)) { set_error("candidates_select: empty score range"); return kErrInvalid; }
  const float scale = (float)kCandBins / (hi - lo);
  const CandWs w = cand_ws(ws, rows);
  if (n <= cand_single_max()) {
    hipLaunchKernelGGL(cand_single_kernel, dim3(rows), dim3(kRowThreads), 0, stream, scores, ld, n, lo, scale, (unsigned int)cap, w, out_pos, out_approx, cand_ld);
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
  }
  static const int cand_wgs = [] { const char* e = getenv("RAILS_CAND_WGS"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 16384 ? v : 0; }();
  const int target = cand_wgs ? cand_wgs : 2 * (n_cu > 0 ? n_cu : 256);      // 128 / 256 / 384 / 512 / 1024 / 2048 workgroups at 32 x 695 762: hist 26.9 / 19.3 / - / 18.2 / 20.9 / 31.5 us
  int64_t chunks = (target + rows - 1) / rows;
  const int64_t max_chunks = (n + 8191) / 8192;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  int64_t chunk = (n + chunks - 1) / chunks;
  chunk = (chunk + 3) / 4 * 4;
  chunks = (n + chunk - 1) / chunk;
  hipLaunchKernelGGL(cand_hist_kernel, dim3((unsigned)chunks, rows), dim3(kCandThreads), 0, stream, scores, ld, n, chunk, lo, scale, (unsigned int)cap, w);
  hipLaunchKernelGGL(cand_compact_kernel, dim3((unsigned)chunks, rows), dim3(kCandThreads), 0, stream, scores, ld, n, chunk, lo, scale, (unsigned int)cap, w,
                     out_pos, out_approx, cand_ld);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- the fused finish -------------------------------------------------------------------------------------------------------------
struct CandFinishArgs {
  const float* exact; int64_t ld;            // fp32 logits of the candidates (row stride ld)
  const float* approx; const int64_t* pos; int64_t cand_ld;   // first-pass scores and corpus positions (row stride cand_ld)
  CandWs w; int cap, npad, k;
  const int64_t* ids; int64_t n_items;
  float default_eps, safety; int one_sided;
  const float* guard; int guard_per_row; float guard_limit;
  float* out_scores; int64_t* out_ids;       // (rows, k)
  const int64_t* f_invalid; int f_width, f_k; int64_t* f_out_ids; float* f_out_scores;   // optional seen-id filter over the k winners
  float* state; float* state_host;           // verdict state (rails_rescore_verdict's layout); state_host: optional mirror in pinned host memory
  int clean_hist;                            // the selection was the two-launch one: its global histograms are zeroed here
  int debug;                                 // RAILS_FINISH_DEBUG (measurements): 1 = no verdict commit, 2 = no host mirror, 4 = no sort
  int64_t* msg;                              // sharded form: (rows, 2k + 2) message [k score words | k ids | m | err], no verdict here
};

__global__ __launch_bounds__(kSortThreads) void cand_finish_kernel(const CandFinishArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ float red_min[kSortThreads / 64], red_err[kSortThreads / 64], red_grd[kSortThreads / 64];
  __shared__ int red_bad[kSortThreads / 64];
  __shared__ int64_t f_id[kFuseMaxK], f_inv[kFuseMaxW];
  __shared__ float f_sc[kFuseMaxK];
  __shared__ int f_scratch[kSortThreads / 64 + 2];
  __shared__ int s_last;
  const int row = blockIdx.x, rows = gridDim.x, tid = threadIdx.x, lane = tid & 63;
  const int k = a.k, npad = a.npad;
  const bool fuse = a.f_invalid != nullptr;
  const unsigned int flags = a.w.flags[row];
  const float seen_before = a.state ? a.state[0] : 0.0f;
  if (fuse)
    for (int i = tid; i < a.f_width; i += kSortThreads) f_inv[i] = a.f_invalid[(int64_t)row * a.f_width + i];
  float mn = INFINITY, err = 0.0f, grd = 0.0f;
  int bad = (flags & 1u) ? 1 : 0;
  unsigned int c = a.w.counts[row];
  if (c > (unsigned int)a.cap) c = (unsigned int)a.cap;
  for (int i = tid; i < npad; i += kSortThreads) {
    unsigned long long kv = 0ull;
    float e = 0.0f, ap = 0.0f;
    int64_t p = 0;
    if (i < a.cap) {      // requested before the row's count is known: one round trip to memory, not two (slots past the count hold stale values, masked below)
      e = a.exact[(int64_t)row * a.ld + i];
      ap = a.approx[(int64_t)row * a.cand_ld + i];
      p = a.pos[(int64_t)row * a.cand_ld + i];
    }
    if (i < (int)c) {
      kv = ((unsigned long long)orderable(e) << 32) | (unsigned int)(~(unsigned int)p);
      mn = fminf(mn, ap);
      const float dd = a.one_sided ? fmaxf(e - ap, 0.0f) : fabsf(e - ap);
      bad |= !(dd == dd) || !(e == e) || !(ap == ap);
      err = fmaxf(err, dd == dd ? dd : INFINITY);
    }
    keys[i] = kv;
  }
  if (a.guard)
    for (int i = tid; i < a.guard_per_row; i += kSortThreads) {
      const float v = fabsf(a.guard[(int64_t)row * a.guard_per_row + i]);
      if (!(v <= a.guard_limit)) bad = 1;
      grd = fmaxf(grd, v == v ? v : INFINITY);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    err = fmaxf(err, __shfl_xor(err, o, 64));
    grd = fmaxf(grd, __shfl_xor(grd, o, 64));
    bad |= __shfl_xor(bad, o, 64);
  }
  if (lane == 0) { red_min[tid >> 6] = mn; red_err[tid >> 6] = err; red_grd[tid >> 6] = grd; red_bad[tid >> 6] = bad; }
  // leave the workspace as the next call expects it (nothing below reads it)
  {
    if (a.clean_hist) {
      unsigned int* gf = a.w.fine + (size_t)row * kCandBins;
      for (int i = tid; i < kCandBins; i += kSortThreads) gf[i] = 0u;
      if (tid < kCandCoarse) a.w.coarse[(size_t)row * kCandCoarse + tid] = 0u;
    }
    if (tid == 0) { a.w.counts[row] = 0u; a.w.flags[row] = 0u; }
  }
  __syncthreads();
  if (a.debug & 4) {
  } else if (npad <= kSortThreads) {
    unsigned long long kv = tid < npad ? keys[tid] : 0ull;
    kv = block_sort_desc(kv, npad, keys + npad);
    __syncthreads();
    if (tid < npad) keys[tid] = kv;
    __syncthreads();
  } else if (npad == 2 * kSortThreads) block_sort_desc_multi<2>(keys);
  else if (npad == 4 * kSortThreads) block_sort_desc_multi<4>(keys);
  else if (npad == 8 * kSortThreads) block_sort_desc_multi<8>(keys);
  else block_sort_desc_multi<16>(keys);
  mn = INFINITY; err = 0.0f; grd = 0.0f; bad = 0;
  for (int wv = 0; wv < kSortThreads / 64; ++wv) { mn = fminf(mn, red_min[wv]); err = fmaxf(err, red_err[wv]); grd = fmaxf(grd, red_grd[wv]); bad |= red_bad[wv]; }
  const bool whole_row = (int64_t)c >= a.n_items;          // every item of the row is a candidate: nothing is left outside
  if (whole_row) mn = -INFINITY;
  else if (c == 0u) mn = INFINITY;                          // no candidate of a non-empty row: nothing is known about it
  if (a.msg) {
    // item-sharded form: this rank's part of the global proof travels with its top-k (rails_amd/sharded.py)
    int64_t* m = a.msg + (int64_t)row * (2 * k + 2);
    for (int j = tid; j < k; j += kSortThreads) {
      const unsigned long long kv = j < (int)c ? keys[j] : 0ull;
      const int64_t p = (int64_t)(~(unsigned int)(kv & 0xFFFFFFFFull));
      const float sc = kv ? unorderable((unsigned int)(kv >> 32)) : -INFINITY;
      m[j] = (int64_t)(unsigned long long)__float_as_uint(sc);
      m[k + j] = kv ? (a.ids ? a.ids[p] : p) : -1;
    }
    if (tid == 0) {
      m[2 * k] = (int64_t)(unsigned long long)__float_as_uint(mn);
      m[2 * k + 1] = (int64_t)(unsigned long long)__float_as_uint(bad ? INFINITY : err);
    }
    return;
  }
  for (int j = tid; j < k; j += kSortThreads) {
    const unsigned long long kv = keys[j];
    const int64_t p = (int64_t)(~(unsigned int)(kv & 0xFFFFFFFFull));
    const float sc = unorderable((unsigned int)(kv >> 32));
    const int64_t id = a.ids ? a.ids[p < a.n_items ? p : 0] : p;
    a.out_scores[(int64_t)row * k + j] = sc;
    a.out_ids[(int64_t)row * k + j] = id;
    if (fuse && j < kFuseMaxK) { f_sc[j] = sc; f_id[j] = id; }
  }
  // the row's verdict: its k-th fp32 score must clear the best first-pass score left outside the candidates by eps
  const float kth = (int)c >= k ? unorderable((unsigned int)(keys[k - 1] >> 32)) : -INFINITY;
  const float gap = kth - mn;
  const float eps = fmaxf(a.default_eps, a.safety * fmaxf(seen_before, bad ? 0.0f : err));
  const int fail = bad || (int)c < k || !(gap > eps);
  if (fuse) {
    __syncthreads();
    filter_from_lds<kSortThreads>(f_id, f_sc, k, f_inv, a.f_width, a.f_k, a.f_out_ids + (int64_t)row * a.f_k, a.f_out_scores + (int64_t)row * a.f_k, f_scratch);
  }
  if (!(a.debug & 1)) verdict_commit(a.w.call, row, rows, fail, bad, err, gap, grd, a.default_eps, a.safety, a.state, (a.debug & 2) ? nullptr : a.state_host, &s_last);
}

int candidates_finish(const float* exact, int64_t ld, const float* approx, const int64_t* pos, int64_t cand_ld, int cap, void* ws, const int64_t* ids,
                      int64_t n_items, int rows, int k, float default_eps, float safety, int one_sided, const float* guard, int guard_per_row,
                      float guard_limit, float* out_scores, int64_t* out_ids, const int64_t* f_invalid, int f_width, int f_k, int64_t* f_out_ids,
                      float* f_out_scores, float* state, float* state_host, int64_t* msg, hipStream_t stream) {
  if (rows <= 0) return kOk;
  if (cap < 1 || cap > kSortCap || k < 1 || (k > cap && !msg)) { set_error("candidates_finish: k = %d of cap = %d out of range", k, cap); return kErrInvalid; }
  if (f_invalid && !(k <= kFuseMaxK && f_width >= 0 && f_width <= kFuseMaxW && f_k > 0 && f_k <= k)) {
    set_error("candidates_finish: the seen-id filter cannot be fused at k = %d, width = %d", k, f_width);
    return kErrUnsupported;
  }
  static DynLdsOnce once;
  if (ensure_dyn_lds(once, reinterpret_cast<const void*>(&cand_finish_kernel), kSortCap * (int)sizeof(unsigned long long)) != kOk) return kErrLaunch;
  CandFinishArgs a{};
  a.exact = exact; a.ld = ld; a.approx = approx; a.pos = pos; a.cand_ld = cand_ld; a.w = cand_ws(ws, rows); a.cap = cap; a.k = k;
  int npad = next_pow2(cap < 64 ? 64 : cap);
  a.npad = npad;
  a.ids = ids; a.n_items = n_items; a.default_eps = default_eps; a.safety = safety; a.one_sided = one_sided;
  a.guard = guard; a.guard_per_row = guard ? guard_per_row : 0; a.guard_limit = guard_limit;
  a.out_scores = out_scores; a.out_ids = out_ids;
  a.f_invalid = f_invalid; a.f_width = f_width; a.f_k = f_k; a.f_out_ids = f_out_ids; a.f_out_scores = f_out_scores;
  a.state = state; a.state_host = state_host; a.msg = msg;
  static const int debug = [] { const char* e = getenv("RAILS_FINISH_DEBUG"); return e ? atoi(e) : 0; }();
  a.debug = debug;
  a.clean_hist = n_items > cand_single_max() ? 1 : 0;
  const size_t lds = (size_t)(npad <= kSortThreads ? 3 * npad : npad) * sizeof(unsigned long long);
  hipLaunchKernelGGL(cand_finish_kernel, dim3(rows), dim3(kSortThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace mol
