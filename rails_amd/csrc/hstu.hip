// HSTU query encoder, eval path: the step upstream of the retrieval path, without fbgemm.
//
// Reference: modeling/sequential/hstu.py -- SequentialTransductionUnitJagged.forward (:268-437), the attention
// _hstu_attention_maybe_from_cache (:144-213), RelativeBucketedTimeAndPositionBasedBias (:82-138); the preprocessor
// modeling/sequential/input_features_preprocessors.py:75-92; the postprocessors output_postprocessors.py:38-85.
//
// The reference runs its layers on a jagged (sum of lengths, D) tensor through fbgemm's dense_to_jagged /
// jagged_to_padded_dense.  Here everything stays padded (B, N, D) with the rows at positions >= length held at zero:
// the reference's padded q / k / v rows are zero as well and it drops the outputs at padded positions, so the two are the
// same computation (oracle/hstu_oracle.py reproduces the reference bit for bit this way).
//
// Kernels (fp32 throughout; sizes are tiny next to the retrieval path -- B x N tokens of D <= 256 -- so these are written
// for exactness and few launches, not tuned):
//   hstu_preprocess_kernel   x = [id != 0, n < len] * (emb * sqrt(D) + pos_emb[n])
//   rows_layer_norm_kernel   y = LN(x) (no affine, biased variance), optionally * u         one wave per row
//   gemm_f32_kernel          C = act(A W + bias) + residual, rows of padded positions zeroed;  v_mfma_f32_32x32x2_f32,
//                            one wave per 32 x 32 output tile, W given as (K, N) or as (N, K) (torch Linear)
//   hstu_attention_kernel    a[b, i, h, :] = sum_{j <= i} silu(q_i . k_j + bias[b, i, j]) / N * v_j
//                            register-chained like the scoring kernel: S^T = K Q^T puts key j of a tile in accumulator
//                            register r, which IS the B operand of the K-step {row(r,0), row(r,1)} of O^T += V^T P^T
//   rows_normalize_kernel    LayerNorm or L2 normalisation of selected rows (the postprocessor + get_current_embeddings)
#include <hip/hip_runtime.h>
#include <math.h>

#include "mol_kernels.h"
#include "mol_layout.h"

namespace mol {

typedef float hf32x16 __attribute__((ext_vector_type(16)));

__global__ void hstu_preprocess_kernel(const float* __restrict__ emb, const int64_t* __restrict__ ids,
                                       const int64_t* __restrict__ lengths, const float* __restrict__ pos_emb, int B, int N,
                                       int D, float scale, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * N * D) return;
  const int64_t tok = i / D;
  const int dd = (int)(i - tok * D);
  const int b = (int)(tok / N), n = (int)(tok - (int64_t)b * N);
  const bool valid = ids[tok] != 0 && n < lengths[b];
  out[i] = valid ? emb[i] * scale + pos_emb[(int64_t)n * D + dd] : 0.0f;
}

// one wave per row; two passes over the row (mean, then centred sum of squares), like F.layer_norm
__global__ __launch_bounds__(256) void rows_layer_norm_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int D, float eps,
                                                              const float* __restrict__ mul, int64_t ldm, float* __restrict__ out,
                                                              int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float s = 0.0f;
  for (int k = lane; k < D; k += 64) s += xr[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)D;
  float v = 0.0f;
  for (int k = lane; k < D; k += 64) { const float c = xr[k] - mean; v = __builtin_fmaf(c, c, v); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const float rstd = 1.0f / sqrtf(v / (float)D + eps);
  for (int k = lane; k < D; k += 64) {
    float y = (xr[k] - mean) * rstd;
    if (mul) y *= mul[row * ldm + k];
    out[row * ldo + k] = y;
  }
}

// mode 0: LayerNorm (no affine), 1: x / max(||x||_2, eps).  row_index (nullable) selects source rows.
__global__ __launch_bounds__(256) void rows_normalize_kernel(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ row_index,
                                                             int64_t rows, int D, int mode, float eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (row_index ? row_index[r] : r) * ldx;
  if (mode == 0) {
    float s = 0.0f;
    for (int k = lane; k < D; k += 64) s += xr[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float v = 0.0f;
    for (int k = lane; k < D; k += 64) { const float c = xr[k] - mean; v = __builtin_fmaf(c, c, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const float rstd = 1.0f / sqrtf(v / (float)D + eps);
    for (int k = lane; k < D; k += 64) out[r * D + k] = (xr[k] - mean) * rstd;
  } else {
    float v = 0.0f;
    for (int k = lane; k < D; k += 64) v = __builtin_fmaf(xr[k], xr[k], v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const float nrm = fmaxf(sqrtf(v), eps);
    for (int k = lane; k < D; k += 64) out[r * D + k] = xr[k] / nrm;
  }
}

struct GemmArgs {
  const float* A; int64_t lda;
  const float* W; int w_is_nk;          // 0: W[k * N + n]   1: W[n * K + k] (torch.nn.Linear.weight)
  const float* bias; const float* residual; int64_t ldr;
  int64_t M; int N, K; int act;          // act 1: silu
  const int64_t* lengths; int seq_len;   // rows r = b * seq_len + n with n >= lengths[b] are written as zeros (nullable)
  float* C; int64_t ldc;
};

// one wave per 32 x 32 tile of C; A rows on the MFMA row axis, output columns on the column axis
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 31, h = lane >> 5, col = lane & 31;
  const int64_t tiles_n = (g.N + 31) / 32;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= ((g.M + 31) / 32) * tiles_n) return;
  const int64_t m0 = (tile / tiles_n) * 32;
  const int n0 = (int)(tile % tiles_n) * 32;
  const int64_t ar = m0 + row < g.M ? m0 + row : g.M - 1;   // clamped (always valid) addresses, zeroed below
  const int wc = n0 + col < g.N ? n0 + col : g.N - 1;
  hf32x16 acc = {0};
  for (int k0 = 0; k0 < g.K; k0 += 32) {
    float av[16], bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int k = k0 + 16 * h + s;
      const int kc = k < g.K ? k : g.K - 1;
      const float a = g.A[ar * g.lda + kc];
      const float b = g.w_is_nk ? g.W[(int64_t)wc * g.K + kc] : g.W[(int64_t)kc * g.N + wc];
      av[s] = k < g.K ? a : 0.0f;
      bv[s] = k < g.K ? b : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
  }
  const int n = n0 + col;
  if (n >= g.N) return;
  const float bias = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + acc_row(r, h);
    if (m >= g.M) continue;
    float v = acc[r] + bias;
    if (g.act == 1) v = v / (1.0f + expf(-v));
    if (g.residual) v += g.residual[m * g.ldr + n];
    if (g.lengths) {
      const int64_t b = m / g.seq_len;
      if (m - b * g.seq_len >= g.lengths[b]) v = 0.0f;
    }
    g.C[m * g.ldc + n] = v;
  }
}

struct AttnArgs {
  const float* uvqk; int64_t ld;        // (B * N, ld) rows [u | v | q | k], u/v: H*dv wide, q/k: H*dqk wide
  int B, N, H, dqk, dv;
  const int64_t* lengths;
  const int64_t* timestamps;            // (B, N) or NULL (no bias at all, as the reference)
  const float* ts_w; const float* pos_w;
  const int64_t* thresholds; int num_buckets;   // thresholds[b-1] = smallest |dt| in bucket >= b
  float* out;                           // (B * N, H * dv)
};

// grid (query tiles of 32, H, B), one wave each
__global__ __launch_bounds__(64) void hstu_attention_kernel(AttnArgs a) {
  __shared__ long long thr_s[128];
  const int lane = threadIdx.x;
  const int x = lane & 31, h = lane >> 5;
  const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int N = a.N, H = a.H, dqk = a.dqk, dv = a.dv;
  for (int i = lane; i < a.num_buckets && i < 128; i += 64) thr_s[i] = a.thresholds[i];
  __syncthreads();
  const int64_t len = a.lengths[b];
  const int i0 = qt * 32;
  const float* base = a.uvqk + (int64_t)b * N * a.ld;
  const float* V = base + (int64_t)H * dv + (int64_t)head * dv;
  const float* Q = base + 2 * (int64_t)H * dv + (int64_t)head * dqk;
  const float* Kp = Q + (int64_t)H * dqk;
  const int qi = i0 + x < N ? i0 + x : N - 1;        // this lane's query (column axis)
  const float inv_n = 1.0f / (float)N;
  // ts[b][min(i + 1, N - 1)] of the lane's query
  long long ts_q = 0;
  if (a.timestamps) ts_q = a.timestamps[(int64_t)b * N + (qi + 1 < N ? qi + 1 : N - 1)];
  hf32x16 O = {0};                                    // O^T: row = value dim, column = query
  for (int kt = 0; kt <= qt; ++kt) {                  // causal: key tiles up to the query tile
    const int j0 = kt * 32;
    // S^T = K_tile Q_tile^T : A = keys (rows), B = queries (columns), K axis = dqk
    hf32x16 S = {0};
    const int kj = j0 + x < N ? j0 + x : N - 1;
    for (int s = 0; s < (dqk + 1) / 2; ++s) {
      const int d = 2 * s + h;
      const float ka = d < dqk ? Kp[(int64_t)kj * a.ld + d] : 0.0f;
      const float qb = d < dqk ? Q[(int64_t)qi * a.ld + d] : 0.0f;
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(ka, qb, S, 0, 0, 0);
    }
    // P^T[j][i] = silu(S + bias) / N for j <= i, and the K-step r of O^T += V^T P^T in one go
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + acc_row(r, h);   // this lane's key for register r
      float sc = S[r];
      if (a.timestamps && j < N) {
        long long dt = ts_q - a.timestamps[(int64_t)b * N + j];
        if (dt < 0) dt = -dt;
        int lo = 0, hi2 = a.num_buckets;              // bucket = number of thresholds <= |dt|
        while (lo < hi2) {
          const int mid = (lo + hi2) >> 1;
          if (thr_s[mid] <= dt) lo = mid + 1; else hi2 = mid;
        }
        sc += a.pos_w[N - 1 + j - qi] + a.ts_w[lo];
      }
      float pv = sc / (1.0f + expf(-sc)) * inv_n;
      if (j > qi || j >= N || i0 + x >= N) pv = 0.0f;
      // A operand of this K-step: V[key row(r, h)][d = lane & 31]
      const int key = j0 + acc_row(r, h);
      const float va = (x < dv && key < N) ? V[(int64_t)key * a.ld + x] : 0.0f;
      O = __builtin_amdgcn_mfma_f32_32x32x2f32(va, pv, O, 0, 0, 0);
    }
  }
  const int qrow = i0 + x;
  if (qrow >= N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = acc_row(r, h);
    if (d < dv) a.out[((int64_t)b * N + qrow) * ((int64_t)H * dv) + (int64_t)head * dv + d] = qrow < len ? O[r] : 0.0f;
  }
}

int hstu_preprocess(const float* emb, const int64_t* ids, const int64_t* lengths, const float* pos_emb, int B, int N, int D,
                    float scale, float* out, hipStream_t stream) {
  const int64_t total = (int64_t)B * N * D;
  if (total == 0) return kOk;
  hipLaunchKernelGGL(hstu_preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, emb, ids, lengths, pos_emb, B, N,
                     D, scale, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int rows_layer_norm(const float* x, int64_t ldx, int64_t rows, int D, float eps, const float* mul, int64_t ldm, float* out, int64_t ldo,
                    hipStream_t stream) {
  if (rows == 0) return kOk;
  hipLaunchKernelGGL(rows_layer_norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, ldx, rows, D, eps, mul, ldm, out, ldo);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int rows_normalize(const float* x, int64_t ldx, const int64_t* row_index, int64_t rows, int D, int mode, float eps, float* out,
                   hipStream_t stream) {
  if (rows == 0) return kOk;
  hipLaunchKernelGGL(rows_normalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, ldx, row_index, rows, D, mode, eps, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int gemm_f32(const float* A, int64_t lda, const float* W, int w_is_nk, const float* bias, const float* residual, int64_t ldr, int64_t M,
             int N, int K, int act, const int64_t* lengths, int seq_len, float* C, int64_t ldc, hipStream_t stream) {
  if (M == 0 || N == 0) return kOk;
  GemmArgs g{A, lda, W, w_is_nk, bias, residual, ldr, M, N, K, act, lengths, seq_len, C, ldc};
  const int64_t tiles = ((M + 31) / 32) * ((N + 31) / 32);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, g);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int hstu_attention(const float* uvqk, int64_t ld, int B, int N, int H, int dqk, int dv, const int64_t* lengths,
                   const int64_t* timestamps, const float* ts_w, const float* pos_w, const int64_t* thresholds, int num_buckets,
                   float* out, hipStream_t stream) {
  if (B == 0 || N == 0) return kOk;
  if (dv > 32) { set_error("hstu_attention: dv = %d (supported: <= 32)", dv); return kErrUnsupported; }
  if (num_buckets > 128) { set_error("hstu_attention: %d time buckets (supported: <= 128)", num_buckets); return kErrUnsupported; }
  AttnArgs a{uvqk, ld, B, N, H, dqk, dv, lengths, timestamps, ts_w, pos_w, thresholds, num_buckets, out};
  hipLaunchKernelGGL(hstu_attention_kernel, dim3((N + 31) / 32, H, B), dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace mol
