// Query-side prologue: one workgroup per query row.
//
//   Eq = l2norm(cat[Linear(GLU(q W + b)), uid_emb[(user_id % hash) + 1]])   (B, P_Q, d)
//        reference: rails/similarities/mol/query_embeddings_fns.py:175-254, GLU rails/similarities/layers.py:19-74
//   gq = Linear_nobias(silu(Linear(q)))                                     (B, L), from the RAW q
//        reference: modeling/similarity_utils.py:153-168, applied rails/similarities/mol/similarity_fn.py:166-169
//
// B rows of a few hundred KFLOP each: not a tuning target.  The kernel
// writes both the plain tensors (for the module's accessors) and the MFMA-fragment-ordered copies the
// scoring kernel reads (mol_layout.h).  fp32, precise expf/erff, true divisions.
#include <hip/hip_runtime.h>

#include "mol_kernels.h"
#include "mol_layout.h"

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
};

// out[c] = bias[c] + sum_k W[c][k] in[k]; a wave owns four output columns at a time, lanes stride k, then a shuffle
// reduction per column.  KPL = ceil(K / 64) is a compile-time bound so that all 4 * KPL weight loads of a column group are
// issued before the first FMA (the layer is a chain of L2 latencies otherwise); accumulation order is k ascending per lane.
template <int KPL>
__device__ __forceinline__ void wave_dense_t(const float* __restrict__ W, const float* __restrict__ bias, int ncols,
                                             int K, const float* __restrict__ in_s, float* __restrict__ out_s,
                                             bool silu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = kQueryThreads / 64;
  float xv[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) xv[i] = (lane + 64 * i < K) ? in_s[lane + 64 * i] : 0.0f;
  for (int c0 = wave * 4; c0 < ncols; c0 += nw * 4) {
    float wv[4][KPL];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < KPL; ++i)
        wv[j][i] = (c0 + j < ncols && lane + 64 * i < K) ? W[(int64_t)(c0 + j) * K + lane + 64 * i] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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

__device__ __forceinline__ void wave_dense(const float* __restrict__ W, const float* __restrict__ bias, int ncols,
                                           int K, const float* __restrict__ in_s, float* __restrict__ out_s,
                                           bool silu) {
  const int kpl = (K + 63) / 64;
  if (kpl <= 1) wave_dense_t<1>(W, bias, ncols, K, in_s, out_s, silu);
  else if (kpl <= 2) wave_dense_t<2>(W, bias, ncols, K, in_s, out_s, silu);
  else if (kpl <= 4) wave_dense_t<4>(W, bias, ncols, K, in_s, out_s, silu);
  else if (kpl <= 8) wave_dense_t<8>(W, bias, ncols, K, in_s, out_s, silu);
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

  if (b >= a.B) {  // padding row of the last query group: zero operand rows
    for (int i = threadIdx.x; i < PQ * d; i += kQueryThreads) {
      const int p = i / d, k = i - p * d;
      const int hi = k / (d / 2), s = k - hi * (d / 2);
      eqf[((s >> 2) * 64 + hi * 32 + qj * PQ + p) * 4 + (s & 3)] = 0.0f;
    }
    return;
  }

  for (int i = threadIdx.x; i < D; i += kQueryThreads) qs[i] = a.q[(int64_t)b * D + i];
  __syncthreads();

  // GLU: h = q W + b (D x 2QH, row-major so lanes stride columns); act(lhs) * rhs
  for (int c = threadIdx.x; c < 2 * QH; c += kQueryThreads) {
    float acc = 0.0f;
    int k = 0;
    for (; k + 16 <= D; k += 16) {   // 16 loads in flight, then 16 FMAs in k order
      float wv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) wv[i] = a.w.q_glu_w[(int64_t)(k + i) * 2 * QH + c];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_fmaf(qs[k + i], wv[i], acc);
    }
    for (; k < D; ++k) acc = __builtin_fmaf(qs[k], a.w.q_glu_w[(int64_t)k * 2 * QH + c], acc);
    glu[c] = acc + a.w.q_glu_b[c];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < QH; c += kQueryThreads) {
    const float l = glu[c], r = glu[QH + c];
    const float act = a.glu == RAILS_GEGLU ? 0.5f * l * (1.0f + erff(l * 0.70710678118654752440f))
                                           : l / (1.0f + expf(-l));
    glu[c] = act * r;  // lhs slot is only read by its own thread
  }
  __syncthreads();

  const int proj_groups = PQ - a.n_uid;
  wave_dense(a.w.q_proj_w, a.w.q_proj_b, proj_groups * d, QH, glu, eqs, false);
  for (int t = 0; t < a.n_uid; ++t) {
    const int64_t hs = a.w.uid_hash_size[t];
    int64_t row = a.user_ids[b] % hs;
    if (row < 0) row += hs;  // python % is non-negative
    row += 1;
    for (int k = threadIdx.x; k < d; k += kQueryThreads) eqs[(proj_groups + t) * d + k] = a.w.uid_table[t][row * d + k];
  }
  // query-only gate on the raw query
  wave_dense(a.w.gq_w1, a.w.gq_b1, a.Hq, D, qs, hq, true);
  __syncthreads();
  wave_dense(a.w.gq_w2, nullptr, L, a.Hq, hq, gqs, false);
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
    eqf[((s >> 2) * 64 + hi * 32 + qj * PQ + p) * 4 + (s & 3)] = v / a.temperature;  // fragment copy carries 1/tau
  }
  for (int i = threadIdx.x; i < L; i += kQueryThreads) {
    if (a.gq_out) a.gq_out[(int64_t)b * L + i] = gqs[i];
    // gqfrag[b][hi][e] = gq[b][logit_of(e, hi)]
    const int hi = i / (L / 2), e = i - hi * (L / 2);
    a.gqfrag[(int64_t)b * L + i] = -kLog2e * gqs[logit_of(e, hi, PQ, a.PX)];  // fragment copy carries -log2e
  }
}

int query_prologue(const Shape& s, const Weights& w, const float* q, const int64_t* user_ids, int B, float* qpack,
                   float* eq_out, float* gq_out, hipStream_t stream) {
  if (B <= 0) return kOk;
  QueryArgs a;
  a.q = q; a.user_ids = user_ids; a.w = w; a.B = B;
  a.D = s.query_embedding_dim; a.PQ = s.query_dot_product_groups; a.PX = s.item_dot_product_groups;
  a.d = s.dot_product_dimension; a.QH = s.query_hidden_dim; a.Hq = s.gating_query_hidden_dim;
  a.n_uid = s.num_uid_tables; a.glu = s.query_nonlinearity; a.l2norm = s.dot_product_l2_norm; a.eps = s.eps; a.temperature = s.temperature;
  const int QT = queries_per_group(s);
  const int n_groups = (B + QT - 1) / QT;
  a.eqfrag = qpack;
  a.gqfrag = qpack + (int64_t)n_groups * 32 * a.d;
  a.eq_out = eq_out; a.gq_out = gq_out;
  const size_t lds = sizeof(float) * (size_t)(a.D + 2 * a.QH + a.PQ * a.d + a.Hq + a.PQ * a.PX + a.PQ);
  hipLaunchKernelGGL(query_prologue_kernel, dim3(n_groups * QT), dim3(kQueryThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace mol
