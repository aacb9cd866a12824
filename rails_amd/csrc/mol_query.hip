// Query-side prologue.
//
//   Eq = l2norm(cat[Linear(GLU(q W + b)), uid_emb[(user_id % hash) + 1]])   (B, P_Q, d)
//        reference: rails/similarities/mol/query_embeddings_fns.py:175-254, GLU rails/similarities/layers.py:19-74
//   gq = Linear_nobias(silu(Linear(q)))                                     (B, L), from the RAW q
//        reference: modeling/similarity_utils.py:153-168, applied rails/similarities/mol/similarity_fn.py:166-169
//
// B rows of a few hundred KFLOP each, but it is on the critical path of every step: what
// matters is latency.  The kernels write both the plain tensors (for the module's accessors) and the
// MFMA-fragment-ordered copies the scoring kernel reads (mol_layout.h).  fp32, precise expf/erff, true divisions.
//
// Two implementations:
//   batched (default)  a tile of 32 queries sits on the MFMA row axis and the weight columns are spread over
//                      workgroups, so the 1-3 MB of query-side weights are read once chip-wide instead of once per
//                      query by a single CU (which bounded the per-query kernel at ~150 GB/s of L2 per CU: 56 us for
//                      ML-20M).  Three short dependent launches: P1 = GLU + first gate layer, P2 = projection + second
//                      gate layer (raw), P3 = uid embeddings, l2norm, fragment packing.  In P1/P2 a workgroup owns 32
//                      output columns and splits K over its waves, one load batch + 16 MFMAs per wave.
//   per-query          one workgroup per query, wave-per-column dot products; kept for hidden sizes that are not
//                      multiples of 32.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "mol_kernels.h"
#include "mol_layout.h"

#ifndef RAILS_BATCHED_PROLOGUE_BYTES
#define RAILS_BATCHED_PROLOGUE_BYTES (1300 * 1024)   // query-side weights per query beyond which the batched (MFMA) kernels take over
#endif

namespace mol {

constexpr int kQueryThreads = 1024;  // 16 waves: the prologue is a chain of small dense layers, latency-bound

struct QueryArgs {
  const float* q;
  const int64_t* user_ids;
  Weights w;
  float* eqfrag;
  float* gqfrag;
  float* eq_out;
  float* gq_out;
  int B;
  int D, PQ, PX, d, QH, Hq, n_uid, glu, l2norm;
  float eps;
  float temperature;
  int split;   // precision f16x3: Eq fragments are written as f16 hi/lo (mol_layout.h)
  float* eqfrag2;   // optional second pack in the OTHER format (fp32 <-> f16 hi/lo), same values: the verified fast modes need both
  float* gqfrag2;
  int has_gate;   // 0: no query-only gate part -> gq = 0
};

// One element of a query group's Eq fragment: K index s of lane half hi, accumulator row `row` (= qj*P_Q + p).
//   fp32:   EqFrag[sc = s/4][lane = hi*32 + row][j = s%4]
//   f16x3:  the lane's chunk pair (2ks, 2ks+1) holds its 8 k-values of K=16 step ks as 8 hi halves, then 8 lo halves
__device__ __forceinline__ void eq_frag_store(float* eqf, int s, int hi, int row, float v, int split) {
  const int sc = s >> 2, lane = hi * 32 + row;
  if (!split) {
    eqf[(sc * 64 + lane) * 4 + (s & 3)] = v;
    return;
  }
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(v, 0.0f));   // round toward zero
  const _Float16 l = (_Float16)(v - (float)h.x);                              // remainder is exact in fp32
  unsigned short* e16 = reinterpret_cast<unsigned short*>(eqf);
  const int jj = 4 * (sc & 1) + (s & 3), base = ((sc & ~1) * 64 + lane) * 8;
  e16[base + jj] = __builtin_bit_cast(unsigned short, h.x);
  e16[base + 512 + jj] = __builtin_bit_cast(unsigned short, l);
}

// out[c] = bias[c] + sum_k W[c][k] in[k]; a wave owns four output columns at a time, lanes stride k, then a shuffle
// reduction per column.  KPL = ceil(K / 64) is a compile-time bound so that all 4 * KPL weight loads of a column group are
// issued before the first FMA (the layer is a chain of L2 latencies otherwise); accumulation order is k ascending per lane.
template <int KPL, int NC = 4>
__device__ __forceinline__ void wave_dense_t(const float* __restrict__ W, const float* __restrict__ bias, int ncols,
                                             int K, const float* __restrict__ in_s, float* __restrict__ out_s,
                                             bool silu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = kQueryThreads / 64;
  float xv[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) xv[i] = (lane + 64 * i < K) ? in_s[lane + 64 * i] : 0.0f;
  for (int c0 = wave * NC; c0 < ncols; c0 += nw * NC) {
    float wv[NC][KPL];
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
      for (int i = 0; i < KPL; ++i)
        wv[j][i] = (c0 + j < ncols && lane + 64 * i < K) ? W[(int64_t)(c0 + j) * K + lane + 64 * i] : 0.0f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < KPL; ++i) acc = __builtin_fmaf(wv[j][i], xv[i], acc);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (lane == 0 && c0 + j < ncols) {
        float v = acc + (bias ? bias[c0 + j] : 0.0f);
        if (silu) v = v / (1.0f + expf(-v));
        out_s[c0 + j] = v;
      }
    }
  }
}

// wave_dense for ONE pass of columns (ncols <= 16 NC) whose input vector still has to come from global memory: the weight loads of
// the wave's columns are issued FIRST, the input row is copied to LDS under them, then the barrier -- one L2 round trip instead of
// two in a row.  Same per-column arithmetic as wave_dense_t (k ascending per lane, the same shuffle tree): same bits.
template <int KPL, int NC>
__device__ __forceinline__ void wave_dense_prefetched(const float* __restrict__ W, const float* __restrict__ bias, int ncols, int K,
                                                      const float* __restrict__ src, float* __restrict__ in_s, float* __restrict__ out_s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = wave * NC;
  float wv[NC][KPL];
#pragma unroll
  for (int j = 0; j < NC; ++j)
#pragma unroll
    for (int i = 0; i < KPL; ++i)
      wv[j][i] = (c0 + j < ncols && lane + 64 * i < K) ? W[(int64_t)(c0 + j) * K + lane + 64 * i] : 0.0f;
  for (int i = threadIdx.x; i < K; i += kQueryThreads) in_s[i] = src[i];
  __syncthreads();
  float xv[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) xv[i] = (lane + 64 * i < K) ? in_s[lane + 64 * i] : 0.0f;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < KPL; ++i) acc = __builtin_fmaf(wv[j][i], xv[i], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0 && c0 + j < ncols) out_s[c0 + j] = acc + (bias ? bias[c0 + j] : 0.0f);
  }
}

__device__ __forceinline__ void wave_dense(const float* __restrict__ W, const float* __restrict__ bias, int ncols,
                                           int K, const float* __restrict__ in_s, float* __restrict__ out_s,
                                           bool silu) {
  const int kpl = (K + 63) / 64;
  // a wave owns NC output columns per pass and has all their weight loads in flight: a layer is ceil(ncols / (16 NC)) L2 round
  // trips.  Eight columns where four would need more than one pass (the per-column arithmetic does not depend on NC).
  const bool wide = ncols > 4 * (kQueryThreads / 64);
  if (kpl <= 1) { if (wide) wave_dense_t<1, 8>(W, bias, ncols, K, in_s, out_s, silu); else wave_dense_t<1>(W, bias, ncols, K, in_s, out_s, silu); }
  else if (kpl <= 2) { if (wide) wave_dense_t<2, 8>(W, bias, ncols, K, in_s, out_s, silu); else wave_dense_t<2>(W, bias, ncols, K, in_s, out_s, silu); }
  else if (kpl <= 4) { if (wide) wave_dense_t<4, 8>(W, bias, ncols, K, in_s, out_s, silu); else wave_dense_t<4>(W, bias, ncols, K, in_s, out_s, silu); }
  else if (kpl <= 8) { if (wide) wave_dense_t<8, 8>(W, bias, ncols, K, in_s, out_s, silu); else wave_dense_t<8>(W, bias, ncols, K, in_s, out_s, silu); }
  else {  // generic (query_hidden_dim > 512): rolled loop
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = kQueryThreads / 64;
    for (int c = wave; c < ncols; c += nw) {
      float acc = 0.0f;
      for (int k = lane; k < K; k += 64) acc = __builtin_fmaf(W[(int64_t)c * K + k], in_s[k], acc);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (lane == 0) {
        float v = acc + (bias ? bias[c] : 0.0f);
        if (silu) v = v / (1.0f + expf(-v));
        out_s[c] = v;
      }
    }
  }
}

// One pre-activation column of the GLU layer, h[c] = b[c] + sum_k q[k] W[k][c]: a thread owns the column, up to 64 loads in flight
// (one L2 round trip for D <= 64), FMAs in k order.  Shared by the per-query kernel and the split kernels: same bits.
__device__ __forceinline__ float glu_column(const QueryArgs& a, const float* qs, int c) {
  const int D = a.D, QH = a.QH;
  float acc = 0.0f;
  int k = 0;
  for (; k + 64 <= D; k += 64) {
    float wv[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) wv[i] = a.w.q_glu_w[(int64_t)(k + i) * 2 * QH + c];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc = __builtin_fmaf(qs[k + i], wv[i], acc);
  }
  for (; k + 32 <= D; k += 32) {
    float wv[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) wv[i] = a.w.q_glu_w[(int64_t)(k + i) * 2 * QH + c];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc = __builtin_fmaf(qs[k + i], wv[i], acc);
  }
  for (; k + 16 <= D; k += 16) {
    float wv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wv[i] = a.w.q_glu_w[(int64_t)(k + i) * 2 * QH + c];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_fmaf(qs[k + i], wv[i], acc);
  }
  for (; k < D; ++k) acc = __builtin_fmaf(qs[k], a.w.q_glu_w[(int64_t)k * 2 * QH + c], acc);
  return acc + a.w.q_glu_b[c];
}
__device__ __forceinline__ float glu_act(const QueryArgs& a, float l, float r) {
  const float act = a.glu == RAILS_GEGLU ? 0.5f * l * (1.0f + erff(l * 0.70710678118654752440f)) : l / (1.0f + expf(-l));
  return act * r;
}

// grid (padded queries, 2): blockIdx.y = 0 computes the query's sub-embeddings (GLU -> projection -> l2norm -> Eq fragments),
// blockIdx.y = 1 its query-only gate row (two small layers -> gq fragments).  The two chains share nothing but the input row, so
// they run as separate workgroups on different CUs instead of one after the other (the gate chain was ~4 us of a 29 us kernel).
__global__ __launch_bounds__(kQueryThreads) void query_prologue_kernel(QueryArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, d = a.d, PQ = a.PQ, L = a.PQ * a.PX, QH = a.QH;
  const int QT = 32 / PQ;
  float* qs = smem;             // [D]
  float* glu = qs + D;          // [2*QH] then [QH]
  float* eqs = glu + 2 * QH;    // [PQ*d]
  float* hq = eqs + PQ * d;     // [Hq]
  float* gqs = hq + a.Hq;       // [L]
  float* inv = gqs + L;         // [PQ]
  const int b = blockIdx.x;     // padded query index: [0, n_groups * QT)
  const int g = b / QT, qj = b % QT;
  float* eqf = a.eqfrag + (int64_t)g * 32 * d;
  const bool gate_role = blockIdx.y == 1;

  if (b >= a.B) {  // padding row of the last query group: zero operand rows
    if (gate_role) return;
    for (int i = threadIdx.x; i < PQ * d; i += kQueryThreads) {
      const int p = i / d, k = i - p * d;
      const int hi = k / (d / 2), s = k - hi * (d / 2);
      eq_frag_store(eqf, s, hi, qj * PQ + p, 0.0f, a.split);
      if (a.eqfrag2) eq_frag_store(a.eqfrag2 + (int64_t)g * 32 * d, s, hi, qj * PQ + p, 0.0f, !a.split);
    }
    return;
  }

  for (int i = threadIdx.x; i < D; i += kQueryThreads) qs[i] = a.q[(int64_t)b * D + i];
  __syncthreads();

  if (gate_role) {
    // query-only gate on the raw query; absent part (gating_query_fn = False): zeros
    if (a.has_gate) {
      wave_dense(a.w.gq_w1, a.w.gq_b1, a.Hq, D, qs, hq, true);
      __syncthreads();
      wave_dense(a.w.gq_w2, nullptr, L, a.Hq, hq, gqs, false);
    } else {
      for (int i = threadIdx.x; i < L; i += kQueryThreads) gqs[i] = 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += kQueryThreads) {
      if (a.gq_out) a.gq_out[(int64_t)b * L + i] = gqs[i];
      // gqfrag[b][hi][e] = gq[b][logit_of(e, hi)]
      const int hi = i / (L / 2), e = i - hi * (L / 2);
      a.gqfrag[(int64_t)b * L + i] = -kLog2e * gqs[logit_of(e, hi, PQ, a.PX)];  // fragment copy carries -log2e
      if (a.gqfrag2) a.gqfrag2[(int64_t)b * L + i] = -kLog2e * gqs[logit_of(e, hi, PQ, a.PX)];
    }
    return;
  }

  // GLU: h = q W + b (D x 2QH, row-major so lanes stride columns); act(lhs) * rhs.  QH = 0: the projection is a plain Linear
  for (int c = threadIdx.x; c < 2 * QH; c += kQueryThreads) glu[c] = glu_column(a, qs, c);
  __syncthreads();
  for (int c = threadIdx.x; c < QH; c += kQueryThreads) {
    glu[c] = glu_act(a, glu[c], glu[QH + c]);  // lhs slot is only read by its own thread
  }
  __syncthreads();

  const int proj_groups = PQ - a.n_uid;
  if (QH > 0) wave_dense(a.w.q_proj_w, a.w.q_proj_b, proj_groups * d, QH, glu, eqs, false);
  else wave_dense(a.w.q_proj_w, a.w.q_proj_b, proj_groups * d, D, qs, eqs, false);   // similarity_utils.py:108-116
  for (int t = 0; t < a.n_uid; ++t) {
    const int64_t hs = a.w.uid_hash_size[t];
    int64_t row = a.user_ids[b] % hs;
    if (row < 0) row += hs;  // python % is non-negative
    row += 1;
    for (int k = threadIdx.x; k < d; k += kQueryThreads) eqs[(proj_groups + t) * d + k] = a.w.uid_table[t][row * d + k];
  }
  __syncthreads();   // the projection's eqs (written by all waves above) are read for the norms below
  if (threadIdx.x < PQ) {
    float ss = 0.0f;
    for (int k = 0; k < d; ++k) {
      const float v = eqs[threadIdx.x * d + k];
      ss = __builtin_fmaf(v, v, ss);
    }
    inv[threadIdx.x] = a.l2norm ? fmaxf(sqrtf(ss), a.eps) : 1.0f;
  }
  __syncthreads();

  for (int i = threadIdx.x; i < PQ * d; i += kQueryThreads) {
    const int p = i / d, k = i - p * d;
    const float v = eqs[i] / inv[p];
    if (a.eq_out) a.eq_out[(int64_t)b * PQ * d + i] = v;
    // EqFrag[g][sc][lane][j] = Eq[g*QT + row/PQ][row%PQ][kdim_of(4sc + j, hi)], lane = hi*32 + row
    const int hi = k / (d / 2), s = k - hi * (d / 2);
    eq_frag_store(eqf, s, hi, qj * PQ + p, v / a.temperature, a.split);  // fragment copy carries 1/tau
    if (a.eqfrag2) eq_frag_store(a.eqfrag2 + (int64_t)g * 32 * d, s, hi, qj * PQ + p, v / a.temperature, !a.split);
  }
}

// ---------------------------------------------------------------------------------------------
// Split prologue (round 4): the per-query kernel's arithmetic -- the same bits -- spread over more workgroups in TWO short launches.
// The per-query kernel streams all of a query's weights (1.1 MB for ML-1M) through ONE CU at the ~150 GB/s a CU draws from L2: 22 us
// for 32 queries, a third of an ML-1M step.  Here
//   launch 1   workgroup = (128 GLU outputs of one query): 256 threads, a thread per pre-activation column (lhs and gate half), then
//              act(lhs) * rhs -> a scratch row in the query pack;
//   launch 2   workgroup = (one sub-embedding group p of one query): its d projection columns over the GLU row (or the uid
//              embedding), its l2 norm, its share of the Eq fragments; one more workgroup per query runs the query-only gate chain.
// A workgroup touches 25-260 KB of weights instead of 1-3 MB; the launch boundary replaces the per-query kernel's barrier between
// the two layers.
// ---------------------------------------------------------------------------------------------
constexpr int kGluSlice = 128, kGluThreads = 256;
__global__ __launch_bounds__(kGluThreads) void query_glu_slice_kernel(QueryArgs a, float* __restrict__ glu_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qs = smem;            // [D]
  float* pre = smem + a.D;     // [256]: lhs columns, then their gate columns
  const int b = blockIdx.y, c0 = blockIdx.x * kGluSlice, QH = a.QH;
  for (int i = threadIdx.x; i < a.D; i += kGluThreads) qs[i] = a.q[(int64_t)b * a.D + i];
  __syncthreads();
  const int half = threadIdx.x >> 7, cc = c0 + (threadIdx.x & (kGluSlice - 1));
  if (cc < QH) pre[threadIdx.x] = glu_column(a, qs, half * QH + cc);
  __syncthreads();
  if (threadIdx.x < kGluSlice && cc < QH) glu_out[(int64_t)b * QH + cc] = glu_act(a, pre[threadIdx.x], pre[kGluSlice + threadIdx.x]);
}

// grid (P_Q + 1, padded queries): x < P_Q: sub-embedding group x; x == P_Q: the query-only gate row
__global__ __launch_bounds__(kQueryThreads) void query_group_kernel(QueryArgs a, const float* __restrict__ glu_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, d = a.d, PQ = a.PQ, L = a.PQ * a.PX, QH = a.QH;
  const int QT = 32 / PQ;
  const int K = QH > 0 ? QH : D;
  float* in_s = smem;           // [max(K, D)]: the GLU row (projection groups) or the raw query (gate chain, plain-Linear projection)
  float* eqs = in_s + (K > D ? K : D);   // [d]
  float* hq = eqs + d;          // [Hq]
  float* gqs = hq + a.Hq;       // [L]
  __shared__ float inv;
  const int b = blockIdx.y, p = blockIdx.x;
  const int g = b / QT, qj = b % QT;
  float* eqf = a.eqfrag + (int64_t)g * 32 * d;
  if (p == PQ) {   // gate chain, as in query_prologue_kernel
    if (b >= a.B) return;
    for (int i = threadIdx.x; i < D; i += kQueryThreads) in_s[i] = a.q[(int64_t)b * D + i];
    __syncthreads();
    if (a.has_gate) {
      wave_dense(a.w.gq_w1, a.w.gq_b1, a.Hq, D, in_s, hq, true);
      __syncthreads();
      wave_dense(a.w.gq_w2, nullptr, L, a.Hq, hq, gqs, false);
    } else {
      for (int i = threadIdx.x; i < L; i += kQueryThreads) gqs[i] = 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += kQueryThreads) {
      if (a.gq_out) a.gq_out[(int64_t)b * L + i] = gqs[i];
      const int hi = i / (L / 2), e = i - hi * (L / 2);
      a.gqfrag[(int64_t)b * L + i] = -kLog2e * gqs[logit_of(e, hi, PQ, a.PX)];
      if (a.gqfrag2) a.gqfrag2[(int64_t)b * L + i] = -kLog2e * gqs[logit_of(e, hi, PQ, a.PX)];
    }
    return;
  }
  if (b >= a.B) {  // padding row of the last query group: zero operand rows
    for (int k = threadIdx.x; k < d; k += kQueryThreads) {
      const int hi = k / (d / 2), s = k - hi * (d / 2);
      eq_frag_store(eqf, s, hi, qj * PQ + p, 0.0f, a.split);
      if (a.eqfrag2) eq_frag_store(a.eqfrag2 + (int64_t)g * 32 * d, s, hi, qj * PQ + p, 0.0f, !a.split);
    }
    return;
  }
  const int proj_groups = PQ - a.n_uid;
  if (p < proj_groups) {
    const float* src = QH > 0 ? glu_in + (int64_t)b * QH : a.q + (int64_t)b * D;
    const float* Wp = a.w.q_proj_w + (int64_t)p * d * K;
    const float* bp = a.w.q_proj_b + p * d;
    const int kpl = (K + 63) / 64;
    if (d <= 64 && kpl <= 8) {          // the wave_dense dispatch of this size (four columns per wave), weights requested ahead
      if (kpl <= 1) wave_dense_prefetched<1, 4>(Wp, bp, d, K, src, in_s, eqs);
      else if (kpl <= 2) wave_dense_prefetched<2, 4>(Wp, bp, d, K, src, in_s, eqs);
      else if (kpl <= 4) wave_dense_prefetched<4, 4>(Wp, bp, d, K, src, in_s, eqs);
      else wave_dense_prefetched<8, 4>(Wp, bp, d, K, src, in_s, eqs);
    } else if (d <= 128 && kpl <= 8) {   // eight columns per wave
      if (kpl <= 1) wave_dense_prefetched<1, 8>(Wp, bp, d, K, src, in_s, eqs);
      else if (kpl <= 2) wave_dense_prefetched<2, 8>(Wp, bp, d, K, src, in_s, eqs);
      else if (kpl <= 4) wave_dense_prefetched<4, 8>(Wp, bp, d, K, src, in_s, eqs);
      else wave_dense_prefetched<8, 8>(Wp, bp, d, K, src, in_s, eqs);
    } else {
      for (int i = threadIdx.x; i < K; i += kQueryThreads) in_s[i] = src[i];
      __syncthreads();
      wave_dense(Wp, bp, d, K, in_s, eqs, false);
    }
  } else {
    const int t = p - proj_groups;
    const int64_t hs = a.w.uid_hash_size[t];
    int64_t row = a.user_ids[b] % hs;
    if (row < 0) row += hs;  // python % is non-negative
    row += 1;
    for (int k = threadIdx.x; k < d; k += kQueryThreads) eqs[k] = a.w.uid_table[t][row * d + k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ss = 0.0f;
    for (int k = 0; k < d; ++k) ss = __builtin_fmaf(eqs[k], eqs[k], ss);
    inv = a.l2norm ? fmaxf(sqrtf(ss), a.eps) : 1.0f;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < d; k += kQueryThreads) {
    const float v = eqs[k] / inv;
    if (a.eq_out) a.eq_out[((int64_t)b * PQ + p) * d + k] = v;
    const int hi = k / (d / 2), s = k - hi * (d / 2);
    eq_frag_store(eqf, s, hi, qj * PQ + p, v / a.temperature, a.split);
    if (a.eqfrag2) eq_frag_store(a.eqfrag2 + (int64_t)g * 32 * d, s, hi, qj * PQ + p, v / a.temperature, !a.split);
  }
}

// ---------------------------------------------------------------------------------------------
// Batched prologue (v_mfma_f32_32x32x2_f32: A = 32 queries x 2 k, B = 2 k x 32 output columns, exact fp32).
// A "macro step" is 32 consecutive k: MFMA step s of half h takes k = 32*ms + 16*h + s, so every lane reads 16
// consecutive floats of its query row / weight row.
// ---------------------------------------------------------------------------------------------
typedef float qf32x16 __attribute__((ext_vector_type(16)));

#ifdef RAILS_QUERY_PHASES   // tools/query_phases.sh: wall-clock stamps (100 MHz) of P1's workgroup (0, 0)
__device__ long long g_qphase[8];
#define RAILS_QPHASE(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_qphase[i] = (long long)wall_clock64(); } while (0)
#else
#define RAILS_QPHASE(i)
#endif
constexpr int kP1Threads = 512;    // 8 waves: K split eight ways (one macro step per wave up to D = 256)
constexpr int kP2Threads = 1024;   // 16 waves: K split sixteen ways (one macro step per wave at QH = 512)
constexpr int kP3Threads = 1024;

__device__ __forceinline__ float silu_precise(float v) { return v / (1.0f + expf(-v)); }

// Sum of the NW per-wave partial tiles, wave order (deterministic).
template <int NW>
__device__ __forceinline__ float sum_parts(const float (*part)[32][33], int rr, int cc) {
  float v = part[0][rr][cc];
#pragma unroll
  for (int w = 1; w < NW; ++w) v += part[w][rr][cc];
  return v;
}

// One (query tile, component p): bring the 32 x d raw values into LDS with one batch of independent loads (uid components
// gather their embedding rows instead), l2-normalise each row, write the plain copy and the EqFrag entries.
// val: 32*d floats of LDS, inv: 32 floats, urow_s: 32 int64.  NT = threads of the calling workgroup.
template <int NT>
__device__ __forceinline__ void finalize_component(const QueryArgs& a, int tile, int p, const float* __restrict__ eq_raw,
                                                   float* val, float* inv, long long* urow_s) {
  const int d = a.d, PQ = a.PQ;
  const int QT = 32 / PQ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int proj_groups = PQ - a.n_uid;
  const bool is_uid = p >= proj_groups;
  if (is_uid) {
    if (threadIdx.x < 32) {
      const int64_t bb = (int64_t)tile * 32 + threadIdx.x;
      long long urow = 0;
      if (bb < a.B) {
        const int64_t hs = a.w.uid_hash_size[p - proj_groups];
        urow = a.user_ids[bb] % hs;
        if (urow < 0) urow += hs;  // python % is non-negative
        urow += 1;
      }
      urow_s[threadIdx.x] = urow;
    }
    __syncthreads();
  }
  // (rr, k) = (e / d, e % d) for e = tid, tid + NT, ...: stepped incrementally -- with one to four waves per SIMD the
  // runtime integer divisions of a per-element index decode were most of this kernel's time
  const int step_r = NT / d, step_k = NT - step_r * d;
  const int rr0 = (int)threadIdx.x / d, k0 = (int)threadIdx.x - rr0 * d;
  {
    int rr = rr0, k = k0;
    for (int e = threadIdx.x; e < 32 * d; e += NT) {
      const int bb = tile * 32 + rr;
      float v = 0.0f;
      if (bb < a.B) v = is_uid ? a.w.uid_table[p - proj_groups][urow_s[rr] * d + k] : eq_raw[(int64_t)bb * (PQ * d) + p * d + k];
      val[e] = v;
      rr += step_r; k += step_k;
      if (k >= d) { k -= d; ++rr; }
    }
  }
  __syncthreads();
  for (int rr = wave; rr < 32; rr += NT / 64) {
    float ss = 0.0f;
    for (int k = lane; k < d; k += 64) ss = __builtin_fmaf(val[rr * d + k], val[rr * d + k], ss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) inv[rr] = a.l2norm ? fmaxf(sqrtf(ss), a.eps) : 1.0f;
  }
  __syncthreads();
  const int padded = (a.B + QT - 1) / QT * QT;   // rows of the last query group exist in EqFrag
  const int qt_shift = __ffs(QT) - 1;            // QT = 32 / P_Q is a power of two
  const int half = d / 2;
  {
    int rr = rr0, k = k0;
    for (int e = threadIdx.x; e < 32 * d; e += NT) {
      const int bb = tile * 32 + rr;
      if (bb < padded) {
        const float v = bb < a.B ? val[e] / inv[rr] : 0.0f;
        if (bb < a.B && a.eq_out) a.eq_out[(int64_t)bb * PQ * d + p * d + k] = v;
        // EqFrag[g][sc][lane][j] = Eq[g*QT + row/PQ][row%PQ][kdim_of(4sc + j, hi)], lane = hi*32 + row
        const int g = bb >> qt_shift, qj = bb & (QT - 1);
        const int hi = k >= half ? 1 : 0, s = k - hi * half;
        eq_frag_store(a.eqfrag + (int64_t)g * 32 * d, s, hi, qj * PQ + p, bb < a.B ? v / a.temperature : 0.0f, a.split);
        if (a.eqfrag2) eq_frag_store(a.eqfrag2 + (int64_t)g * 32 * d, s, hi, qj * PQ + p, bb < a.B ? v / a.temperature : 0.0f, !a.split);
      }
      rr += step_r; k += step_k;
      if (k >= d) { k -= d; ++rr; }
    }
  }
}

// GEMM body of P1 (kept out of the kernel so that the uid role is a plain if/else: an early `return` next to the MFMA
// loop crashed clang-22's SimplifyCFG).
__device__ __forceinline__ void p1_gemm_block(const QueryArgs& a, int tile, int block, float (*part)[kP1Threads / 64][32][33],
                                              float* __restrict__ glu_out, float* __restrict__ hq_out) {
  constexpr int NW = kP1Threads / 64;
  const int D = a.D, QH = a.QH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 31, h = lane >> 5, col = lane & 31;
  const int b = tile * 32 + row;
  const bool valid = b < a.B;
  const int nglu = QH / 32;
  const bool is_glu = block < nglu;
  const int c0 = (is_glu ? block : block - nglu) * 32;
  // biases of this thread's two output elements (same column): loaded now so that the round trip (a first touch of
  // that tensor since the scoring kernel streamed the corpus: TLB miss included) overlaps the operand loads
  RAILS_QPHASE(0);
  const int tcc = threadIdx.x & 31;
  const float bias_l = is_glu ? a.w.q_glu_b[c0 + tcc] : a.w.gq_b1[c0 + tcc];
  const float bias_r = is_glu ? a.w.q_glu_b[QH + c0 + tcc] : 0.0f;
  qf32x16 accl = {0}, accr = {0};
  const int MS = (D + 31) / 32;
  for (int ms = wave; ms < MS; ms += NW) {   // all loads of the step are issued before its first MFMA
    const int kb = ms * 32 + h * 16;
    float av[16], bl[16], br[16];
    // unconditional loads from clamped (always valid) addresses, zeroed afterwards: guarded loads (`k < D ? load : 0`)
    // compiled to a branch per load and the batch took 6 us instead of one round trip
    const int bc = valid ? b : a.B - 1;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int k = kb + s < D ? kb + s : D - 1;
      av[s] = a.q[(int64_t)bc * D + k];
      if (is_glu) {
        bl[s] = a.w.q_glu_w[(int64_t)k * 2 * QH + c0 + col];
        br[s] = a.w.q_glu_w[(int64_t)k * 2 * QH + QH + c0 + col];
      } else {
        bl[s] = a.w.gq_w1[(int64_t)(c0 + col) * D + k];
        br[s] = 0.0f;
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const bool in = kb + s < D;
      av[s] = (valid && in) ? av[s] : 0.0f;
      bl[s] = in ? bl[s] : 0.0f;
      br[s] = in ? br[s] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      accl = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bl[s], accl, 0, 0, 0);
      if (is_glu) accr = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], br[s], accr, 0, 0, 0);
    }
  }
  RAILS_QPHASE(1);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    part[0][wave][acc_row(r, h)][col] = accl[r];
    part[1][wave][acc_row(r, h)][col] = accr[r];
  }
  __syncthreads();
  RAILS_QPHASE(2);
  for (int e = threadIdx.x; e < 1024; e += kP1Threads) {
    const int rr = e >> 5, cc = e & 31;
    const int64_t bb = (int64_t)tile * 32 + rr;
    float l = sum_parts<NW>(part[0], rr, cc) + bias_l;   // cc == tcc: kP1Threads is a multiple of 32
    if (is_glu) {
      const float r = sum_parts<NW>(part[1], rr, cc) + bias_r;
      const float act = a.glu == RAILS_GEGLU ? 0.5f * l * (1.0f + erff(l * 0.70710678118654752440f)) : silu_precise(l);
      glu_out[bb * QH + c0 + cc] = act * r;
    } else {
      hq_out[bb * a.Hq + c0 + cc] = silu_precise(l);
    }
  }
  RAILS_QPHASE(3);
}

__global__ __launch_bounds__(kP1Threads) void query_p1_kernel(QueryArgs a, float* __restrict__ glu_out, float* __restrict__ hq_out) {
  __shared__ float part[2][kP1Threads / 64][32][33];
  __shared__ float inv[32];
  __shared__ long long urow_s[32];
  const int n_gemm = a.QH / 32 + a.Hq / 32;
  if ((int)blockIdx.x < n_gemm) {
    p1_gemm_block(a, blockIdx.y, blockIdx.x, part, glu_out, hq_out);
  } else {   // uid component: 32*d <= 8192 floats of `part` serve as its value buffer
    finalize_component<kP1Threads>(a, blockIdx.y, a.PQ - a.n_uid + ((int)blockIdx.x - n_gemm), nullptr, &part[0][0][0][0], inv, urow_s);
  }
}

// P2: grid.x = 32-column blocks of the projection ((P_Q - n_uid) * d / 32), then L/32 blocks of the second gate layer;
// grid.y = query tiles.  Raw outputs:  eq_raw[b][c] = Wp[c] . glu[b] + bp[c]   gq_raw[b][l] = W2[l] . hq[b]
__global__ __launch_bounds__(kP2Threads) void query_p2_kernel(QueryArgs a, const float* __restrict__ glu, const float* __restrict__ hq,
                                                             float* __restrict__ eq_raw, float* __restrict__ gq_raw) {
  constexpr int NW = kP2Threads / 64;
  __shared__ float part[NW][32][33];
  const int L = a.PQ * a.PX;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 31, h = lane >> 5, col = lane & 31;
  const int tile = blockIdx.y;
  const int nproj = (a.PQ - a.n_uid) * a.d / 32;
  const bool is_proj = (int)blockIdx.x < nproj;
  const int c0 = (is_proj ? blockIdx.x : blockIdx.x - nproj) * 32;
  const int K = is_proj ? a.QH : a.Hq;
  const float* A = is_proj ? glu : hq;
  const float* W = is_proj ? a.w.q_proj_w : a.w.gq_w2;
  const int64_t bb = (int64_t)tile * 32 + row;
  const float bias = is_proj ? a.w.q_proj_b[c0 + (threadIdx.x & 31)] : 0.0f;   // early: see P1
  qf32x16 acc = {0};
  for (int ms = wave; ms < K / 32; ms += NW) {
    const int kb = ms * 32 + h * 16;
    float av[16], bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      av[s] = A[bb * K + kb + s];                                 // scratch rows exist for the whole tile
      bv[s] = W[(int64_t)(c0 + col) * K + kb + s];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][acc_row(r, h)][col] = acc[r];
  __syncthreads();
  {
    const int rr = threadIdx.x >> 5, cc = threadIdx.x & 31;   // 1024 threads = 32 x 32 outputs
    const float v = sum_parts<NW>(part, rr, cc);
    const int64_t b2 = (int64_t)tile * 32 + rr;
    if (is_proj) eq_raw[b2 * (a.PQ * a.d) + c0 + cc] = v + bias;
    else gq_raw[b2 * L + c0 + cc] = v;
  }
}

// P3: grid.x = the P_Q - n_uid projected components (l2norm + plain copy + EqFrag) and one more workgroup for the gate (plain copy + permuted, -log2e-scaled gqfrag); grid.y = query tiles.
__global__ __launch_bounds__(kP3Threads) void query_p3_kernel(QueryArgs a, const float* __restrict__ eq_raw, const float* __restrict__ gq_raw) {
  __shared__ float inv[32];
  const int d = a.d, PQ = a.PQ, L = a.PQ * a.PX;
  const int QT = 32 / PQ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.y;
  const int proj_groups = PQ - a.n_uid;
  if ((int)blockIdx.x == proj_groups) {
    for (int e = threadIdx.x; e < 32 * L; e += kP3Threads) {
      const int rr = e / L, i = e - rr * L;
      const int64_t bb = (int64_t)tile * 32 + rr;
      if (bb >= a.B) continue;
      if (a.gq_out) a.gq_out[bb * L + i] = gq_raw[bb * L + i];
      // gqfrag[b][hi][e] = gq[b][logit_of(e, hi)]
      const int hi = i / (L / 2), ee = i - hi * (L / 2);
      a.gqfrag[bb * L + i] = -kLog2e * gq_raw[bb * L + logit_of(ee, hi, PQ, a.PX)];  // fragment copy carries -log2e
      if (a.gqfrag2) a.gqfrag2[bb * L + i] = -kLog2e * gq_raw[bb * L + logit_of(ee, hi, PQ, a.PX)];
    }
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float val[];   // [32][d]
  __shared__ long long urow_s[32];
  finalize_component<kP3Threads>(a, tile, blockIdx.x, eq_raw, val, inv, urow_s);
}

size_t query_scratch_floats(const Shape& s, int B) {
  const size_t bt = (size_t)(B + 31) / 32 * 32;
  return bt * ((size_t)(s.query_hidden_dim > 0 ? s.query_hidden_dim : 0) + (size_t)(s.gating_query_hidden_dim > 0 ? s.gating_query_hidden_dim : 0) +
               (size_t)s.query_dot_product_groups * s.dot_product_dimension +
               (size_t)s.query_dot_product_groups * s.item_dot_product_groups);
}

int query_prologue(const Shape& s, const Weights& w, const float* q, const int64_t* user_ids, int B, float* qpack,
                   float* eq_out, float* gq_out, hipStream_t stream, float* qpack_other) {
  if (B <= 0) return kOk;
  QueryArgs a;
  a.q = q; a.user_ids = user_ids; a.w = w; a.B = B;
  a.D = s.query_embedding_dim; a.PQ = s.query_dot_product_groups; a.PX = s.item_dot_product_groups;
  a.d = s.dot_product_dimension; a.QH = s.query_hidden_dim > 0 ? s.query_hidden_dim : 0; a.Hq = s.gating_has_query ? s.gating_query_hidden_dim : 0;
  a.has_gate = s.gating_has_query;
  a.n_uid = s.num_uid_tables; a.glu = s.query_nonlinearity; a.l2norm = s.dot_product_l2_norm; a.eps = s.eps; a.temperature = s.temperature;
  a.split = is_split(s) ? 1 : 0;
  const int QT = queries_per_group(s);
  const int n_groups = (B + QT - 1) / QT;
  a.eqfrag = qpack;
  a.gqfrag = qpack + (int64_t)n_groups * 32 * a.d;
  a.eqfrag2 = qpack_other;
  a.gqfrag2 = qpack_other ? qpack_other + (int64_t)n_groups * 32 * a.d : nullptr;
  a.eq_out = eq_out; a.gq_out = gq_out;
  const int L = a.PQ * a.PX;
  // RAILS_PROLOGUE: 0 / unset = choose, 1 = per-query kernel, 2 = batched (MFMA) kernels, 3 = split per-query kernels (measurement override)
  const char* forced_env = getenv("RAILS_PROLOGUE");   // read per call: tests and A/B runs switch it in-process
  const int forced = forced_env ? atoi(forced_env) : 0;
  const bool batched_ok = a.QH > 0 && a.has_gate && a.QH % 32 == 0 && a.Hq % 32 == 0 && a.d % 32 == 0 && L % 32 == 0 && a.d <= 256;
  // The per-query kernel streams every weight matrix through one CU per query (~150 GB/s of L2 each); the batched
  // kernels read them once but pay three dependent launches (~8 us each).  Crossover ~1.3 MB of weights per query.
  const size_t weight_bytes = sizeof(float) * ((size_t)a.D * 2 * a.QH + (size_t)(a.PQ - a.n_uid) * a.d * a.QH + (size_t)a.Hq * a.D + (size_t)L * a.Hq);
  const bool use_batched = forced == 2 || (forced == 0 && weight_bytes > (size_t)RAILS_BATCHED_PROLOGUE_BYTES);
  if (batched_ok && use_batched && !(forced == 0 && a.QH > 0 && B <= 64)) {
    // scratch rows behind the fragment pack (rails_mol_query_pack_floats counts them)
    const int64_t bt = (int64_t)(B + 31) / 32 * 32;
    float* glu = a.gqfrag + (int64_t)B * L;
    float* hq = glu + bt * a.QH;
    float* eq_raw = hq + bt * a.Hq;
    float* gq_raw = eq_raw + bt * a.PQ * a.d;
    const int tiles = (B + 31) / 32;
    hipLaunchKernelGGL(query_p1_kernel, dim3(a.QH / 32 + a.Hq / 32 + a.n_uid, tiles), dim3(kP1Threads), 0, stream, a, glu, hq);
    hipLaunchKernelGGL(query_p2_kernel, dim3((a.PQ - a.n_uid) * a.d / 32 + L / 32, tiles), dim3(kP2Threads), 0, stream, a,
                       (const float*)glu, (const float*)hq, eq_raw, gq_raw);
    hipLaunchKernelGGL(query_p3_kernel, dim3(a.PQ - a.n_uid + 1, tiles), dim3(kP3Threads), 32 * a.d * sizeof(float), stream, a, (const float*)eq_raw,
                       (const float*)gq_raw);
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
  }
  // Measured (profiles/r04_prologue_kernel_times.txt, B = 32): ML-1M per-query 21.7 / batched 18.4 / split 15.0 us; ML-20M 39.6 / 23.7 /
  // 21.6 (B <= 8: 19.6 batched, 14.5 split); amzn-books 14.5 / 17.9 / 15.8.  Beyond ~64 queries the split kernels' 9 workgroups of 1 024
  // threads per query no longer fit one round (B = 128: 30-45 us).
  const bool use_split = forced == 3 || (forced == 0 && a.QH > 0 && B <= 64 && weight_bytes > (size_t)1000 * 1024);
  if (use_split) {
    // split prologue: two short launches, same bits as the per-query kernel (RAILS_PROLOGUE=1 keeps that one)
    float* glu = a.gqfrag + (int64_t)B * L;   // scratch rows behind the fragment pack (rails_mol_query_pack_floats counts them)
    if (a.QH > 0)
      hipLaunchKernelGGL(query_glu_slice_kernel, dim3((a.QH + kGluSlice - 1) / kGluSlice, B), dim3(kGluThreads), sizeof(float) * (size_t)(a.D + kGluThreads), stream, a, glu);
    const int K = a.QH > 0 ? a.QH : a.D;
    const size_t lds2 = sizeof(float) * (size_t)((K > a.D ? K : a.D) + a.d + a.Hq + L);
    hipLaunchKernelGGL(query_group_kernel, dim3(a.PQ + 1, n_groups * QT), dim3(kQueryThreads), lds2, stream, a, (const float*)glu);
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
  }
  const size_t lds = sizeof(float) * (size_t)(a.D + 2 * a.QH + a.PQ * a.d + a.Hq + a.PQ * a.PX + a.PQ);
  hipLaunchKernelGGL(query_prologue_kernel, dim3(n_groups * QT, 2), dim3(kQueryThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace mol

#ifdef RAILS_QUERY_PHASES
extern "C" int rails_debug_query_phases(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mol::g_qphase), sizeof(long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
