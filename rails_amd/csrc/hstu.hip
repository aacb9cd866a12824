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
//   hstu_time_buckets_kernel the (B, N, N) time-bucket matrix of the relative bias, once per encode
//   hstu_attention_kernel    a[b, i, h, :] = sum_{j <= i} silu(q_i . k_j + bias[b, i, j]) / N * v_j
//                            register-chained like the scoring kernel: S^T = K Q^T puts key j of a tile in accumulator
//                            register r, which IS the B operand of the K-step {row(r,0), row(r,1)} of O^T += V^T P^T
//   rows_normalize_kernel    LayerNorm or L2 normalisation of selected rows (the postprocessor + get_current_embeddings)
//   hstu_fused_kernel        the whole encoder in one launch for seq_len <= 64: one workgroup per sequence, all in LDS
#include <hip/hip_runtime.h>
#include <math.h>

#include "mol_kernels.h"
#include "mol_layout.h"

namespace mol {

typedef float hf32x16 __attribute__((ext_vector_type(16)));

// silu on the hardware transcendentals (v_exp_f32 = 2^x, v_rcp_f32; ~1 ulp each) instead of expf + an IEEE division (~50
// instructions per element: the attention kernels were VALU-bound on it); the scoring kernel does the same.
__device__ __forceinline__ float silu_fast(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

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

// one wave per 32 x 32 tile of C; A rows on the MFMA row axis, output columns on the column axis.
// A lane needs 16 consecutive k of ITS row (A, and W in the (N, K) layout): as 16 dword loads that is 16 instructions of
// 64 different cache lines each (the first version ran the uvqk GEMM at 19 % of the MFMA peak on address divergence
// alone), so rows that are 16-byte aligned are read with four 16-byte loads; the next K step's operands are requested
// before the current step's MFMAs.
__device__ __forceinline__ void gemm_load16(const float* __restrict__ p, int k, int K, bool vec, float (&v)[16]) {
  if (vec && k + 16 <= K) {
    const float4* q = reinterpret_cast<const float4*>(p + k);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 f = q[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int kk = k + s < K ? k + s : K - 1;
      const float x = p[kk];
      v[s] = k + s < K ? x : 0.0f;
    }
  }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 31, h = lane >> 5, col = lane & 31;
  const int64_t tiles_n = (g.N + 31) / 32;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= ((g.M + 31) / 32) * tiles_n) return;
  const int64_t m0 = (tile / tiles_n) * 32;
  const int n0 = (int)(tile % tiles_n) * 32;
  const int64_t ar = m0 + row < g.M ? m0 + row : g.M - 1;   // clamped (always valid) addresses; such rows / columns are never stored
  const int wc = n0 + col < g.N ? n0 + col : g.N - 1;
  const float* arow = g.A + ar * g.lda;
  const bool a_vec = ((reinterpret_cast<uintptr_t>(g.A) | (uintptr_t)(g.lda * 4)) & 15) == 0;
  const bool w_vec = g.w_is_nk && ((reinterpret_cast<uintptr_t>(g.W) | (uintptr_t)((int64_t)g.K * 4)) & 15) == 0;
  auto load_b = [&](int k, float (&v)[16]) {
    if (g.w_is_nk) {
      gemm_load16(g.W + (int64_t)wc * g.K, k, g.K, w_vec, v);
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int kk = k + s < g.K ? k + s : g.K - 1;
        const float x = g.W[(int64_t)kk * g.N + wc];     // (K, N): lanes read consecutive columns, coalesced
        v[s] = k + s < g.K ? x : 0.0f;
      }
    }
  };
  hf32x16 acc = {0};
  float av[16], bv[16], an[16], bn[16];
  gemm_load16(arow, 16 * h, g.K, a_vec, av);
  load_b(16 * h, bv);
  for (int k0 = 0; k0 < g.K; k0 += 32) {
    const bool more = k0 + 32 < g.K;
    if (more) {
      gemm_load16(arow, k0 + 32 + 16 * h, g.K, a_vec, an);
      load_b(k0 + 32 + 16 * h, bn);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
    if (more) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { av[s] = an[s]; bv[s] = bn[s]; }
    }
  }
  const int n = n0 + col;
  if (n >= g.N) return;
  const float bias = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + acc_row(r, h);
    if (m >= g.M) continue;
    float v = acc[r] + bias;
    if (g.act == 1) v = silu_fast(v);
    if (g.residual) v += g.residual[m * g.ldr + n];
    if (g.lengths) {
      const int64_t b = m / g.seq_len;
      if (m - b * g.seq_len >= g.lengths[b]) v = 0.0f;
    }
    g.C[m * g.ldc + n] = v;
  }
}

// The same GEMM for the sizes the encoder's layers have (K % 32 == 0, 16-byte aligned rows): a workgroup of four waves owns a
// 64 x 64 tile of C, the 64 x 32 blocks of A and W of a K-step are read from memory ONCE per workgroup with coalesced 16-byte
// loads (eight threads cover a row's 128 bytes) into LDS, double buffered, and every wave takes its 32 x 32 quarter's MFMA
// operands from there.  The per-wave kernel above has each lane read 64 bytes of its own row per step: 64 cache lines per load
// instruction and no reuse between the waves that share a row block -- 28 % of the fp32 MFMA peak on the uvqk GEMM of an ML-20M
// block.  Same operand assignment per lane (k = k0 + 16 h + s for MFMA s), same order over k: bit-identical results.
// LDS rows are 36 floats apart: the 16 lanes of a ds_read_b128 phase then hit 16 distinct 16-byte bank groups.
// (A 128 x 64 tile per workgroup -- a wave owning 64 x 32, half the operand traffic per MFMA -- measured slower: 67 vs 52 us on the
// uvqk GEMM; two workgroups per CU instead of four hide less than the halved traffic buys.  Also measured without effect: two K-steps
// of operands in flight in registers (51.1 us), XCD-aware workgroup numbering (52.5 us) -- neither the L2 round trip nor its locality
// is what holds this kernel at 0.44 of the peak.)
constexpr int kGemmLd = 36;

__global__ __launch_bounds__(256) void gemm_f32_tiled_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float sA[2][64][kGemmLd], sW[2][64][kGemmLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = lane & 31, h = lane >> 5, col = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t tiles_n = (g.N + 63) / 64;
  const int64_t m0 = ((int64_t)blockIdx.x / tiles_n) * 64;
  const int n0 = (int)((int64_t)blockIdx.x % tiles_n) * 64;
  // this thread's pieces of a K-step: A rows (tid / 8) and (tid / 8) + 32, 16 bytes at k = (tid % 8) * 4; W likewise, or -- (K, N)
  // layout -- k rows (tid / 16) and (tid / 16) + 16, 16 bytes at n = (tid % 16) * 4
  const int pr = tid >> 3, pc = (tid & 7) * 4;
  const int64_t ar0 = m0 + pr < g.M ? m0 + pr : g.M - 1, ar1 = m0 + pr + 32 < g.M ? m0 + pr + 32 : g.M - 1;   // clamped: never stored
  const int wr0 = n0 + pr < g.N ? n0 + pr : g.N - 1, wr1 = n0 + pr + 32 < g.N ? n0 + pr + 32 : g.N - 1;
  const int kr = tid >> 4, kc = (tid & 15) * 4;
  const int wcn = n0 + kc + 3 < g.N ? n0 + kc : (g.N >= 4 ? g.N - 4 : 0);   // (K, N): a 16-byte piece inside the row
  float4 ra0, ra1, rw0, rw1;
  auto fetch = [&](int k0) {
    ra0 = *reinterpret_cast<const float4*>(g.A + ar0 * g.lda + k0 + pc);
    ra1 = *reinterpret_cast<const float4*>(g.A + ar1 * g.lda + k0 + pc);
    if (g.w_is_nk) {
      rw0 = *reinterpret_cast<const float4*>(g.W + (int64_t)wr0 * g.K + k0 + pc);
      rw1 = *reinterpret_cast<const float4*>(g.W + (int64_t)wr1 * g.K + k0 + pc);
    } else {
      rw0 = *reinterpret_cast<const float4*>(g.W + (int64_t)(k0 + kr) * g.N + wcn);
      rw1 = *reinterpret_cast<const float4*>(g.W + (int64_t)(k0 + kr + 16) * g.N + wcn);
    }
  };
  auto stash = [&](int buf) {
    *reinterpret_cast<float4*>(&sA[buf][pr][pc]) = ra0;
    *reinterpret_cast<float4*>(&sA[buf][pr + 32][pc]) = ra1;
    if (g.w_is_nk) {
      *reinterpret_cast<float4*>(&sW[buf][pr][pc]) = rw0;
      *reinterpret_cast<float4*>(&sW[buf][pr + 32][pc]) = rw1;
    } else if (n0 + kc + 3 < g.N || g.N < 4) {   // transposed into [n][k]; a piece clamped to the row's end belongs to columns nobody stores
      sW[buf][kc][kr] = rw0.x; sW[buf][kc + 1][kr] = rw0.y; sW[buf][kc + 2][kr] = rw0.z; sW[buf][kc + 3][kr] = rw0.w;
      sW[buf][kc][kr + 16] = rw1.x; sW[buf][kc + 1][kr + 16] = rw1.y; sW[buf][kc + 2][kr + 16] = rw1.z; sW[buf][kc + 3][kr + 16] = rw1.w;
    }
  };
  hf32x16 acc = {0};
  fetch(0);
  stash(0);
  __syncthreads();
  const int steps = g.K / 32;
  for (int t = 0; t < steps; ++t) {
    const int buf = t & 1;
    if (t + 1 < steps) fetch((t + 1) * 32);
    float av[16], bv[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 a = *reinterpret_cast<const float4*>(&sA[buf][wm * 32 + row][16 * h + 4 * i]);
      const float4 b = *reinterpret_cast<const float4*>(&sW[buf][wn * 32 + col][16 * h + 4 * i]);
      av[4 * i] = a.x; av[4 * i + 1] = a.y; av[4 * i + 2] = a.z; av[4 * i + 3] = a.w;
      bv[4 * i] = b.x; bv[4 * i + 1] = b.y; bv[4 * i + 2] = b.z; bv[4 * i + 3] = b.w;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
    if (t + 1 < steps) stash(buf ^ 1);
    __syncthreads();
  }
  const int n = n0 + wn * 32 + col;
  if (n >= g.N) return;
  const float bias = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wm * 32 + acc_row(r, h);
    if (m >= g.M) continue;
    float v = acc[r] + bias;
    if (g.act == 1) v = silu_fast(v);
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
  const unsigned char* buckets;         // (B, N keys, N queries) time buckets from hstu_time_buckets, or NULL (no bias at all)
  const float* ts_w; const float* pos_w; int num_buckets;
  float* out;                           // (B * N, H * dv)
};

// buckets[b][j][i] = #{t : thresholds[t] <= |ts[b][min(i + 1, N - 1)] - ts[b][j]|}: the time bucket of (query i, key j),
// which depends on neither the head nor the layer -- computed once per encode() instead of by every attention launch
// (8 heads x 16 layers redid a 7-step binary search per element: 126 of the 146 us of an ML-20M attention launch).
// Key-major so that the attention kernel's lanes (consecutive queries) read consecutive bytes.
__global__ void hstu_time_buckets_kernel(const int64_t* __restrict__ ts, int B, int N, const int64_t* __restrict__ thresholds,
                                         int num_buckets, unsigned char* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * N * N) return;
  const int i = (int)(idx % N);
  const int64_t bj = idx / N;
  const int j = (int)(bj % N), b = (int)(bj / N);
  long long dt = ts[(int64_t)b * N + (i + 1 < N ? i + 1 : N - 1)] - ts[(int64_t)b * N + j];
  if (dt < 0) dt = -dt;
  int lo = 0, hi = num_buckets;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (thresholds[mid] <= dt) lo = mid + 1; else hi = mid;
  }
  out[idx] = (unsigned char)lo;
}

// grid (query tiles of 32, H, B), one wave each.  Everything a bias lookup needs (the row's timestamps, pos_w, ts_w, the
// bucket thresholds) is staged in LDS once, and a tile's Q / K / V fragments are requested in one batch before its MFMA
// chains: the first version chased two dependent global loads per element and one per MFMA step (146 us per ML-20M block).
constexpr int kAttnMaxSteps = 16;   // dqk <= 32

__global__ __launch_bounds__(64) void hstu_attention_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float attn_smem[];   // pos_w[2N-1] | ts_w[nb+1]
  const int lane = threadIdx.x;
  const int x = lane & 31, h = lane >> 5;
  const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int N = a.N, H = a.H, dqk = a.dqk, dv = a.dv;
  float* pos_s = attn_smem;
  float* tsw_s = pos_s + 2 * N;
  const bool biased = a.buckets != nullptr;
  if (biased) {
    for (int i = lane; i < 2 * N - 1; i += 64) pos_s[i] = a.pos_w[i];
    for (int i = lane; i <= a.num_buckets; i += 64) tsw_s[i] = a.ts_w[i];
  }
  __syncthreads();
  const int64_t len = a.lengths[b];
  const int i0 = qt * 32;
  const float* base = a.uvqk + (int64_t)b * N * a.ld;
  const float* V = base + (int64_t)H * dv + (int64_t)head * dv;
  const float* Q = base + 2 * (int64_t)H * dv + (int64_t)head * dqk;
  const float* Kp = Q + (int64_t)H * dqk;
  const int qi = i0 + x < N ? i0 + x : N - 1;        // this lane's query (column axis)
  const float inv_n = 1.0f / (float)N;
  // K-step s of S^T multiplies the d pair {s, 16 + s}: lane half h holds d = 16 h + s, sixteen CONSECUTIVE floats of its row, read
  // as four 16-byte loads where the rows are 16-byte aligned (the first version paired {2 s, 2 s + 1}: sixteen dword loads of 32
  // different cache lines each, for Q and again for every key tile's K).
  const bool vec = dqk == 32 && ((reinterpret_cast<uintptr_t>(Q) | (uintptr_t)(a.ld * 4)) & 15) == 0;
  auto load_d16 = [&](const float* rowp, float (&v)[kAttnMaxSteps]) {
    if (vec) {
      const float4* q4 = reinterpret_cast<const float4*>(rowp + 16 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 f = q4[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
    } else {
#pragma unroll
      for (int s = 0; s < kAttnMaxSteps; ++s) {
        const int d = 16 * h + s;
        const float t = rowp[d < dqk ? d : 0];
        v[s] = d < dqk ? t : 0.0f;
      }
    }
  };
  float qb[kAttnMaxSteps];
  load_d16(Q + (int64_t)qi * a.ld, qb);
  hf32x16 O = {0};                                    // O^T: row = value dim, column = query
  // a key tile's operands: K rows (A operand of S^T), V rows (A operand of O^T), the (key, query) time buckets.  The NEXT tile's are
  // requested before this tile's MFMA chains (one wave per workgroup and < 2 waves per SIMD in flight: nothing else hides the
  // round trip)
  struct Tile { float ka[kAttnMaxSteps], va[16]; unsigned char bk[16]; };
  auto fetch = [&](int kt, Tile& t) {
    const int j0 = kt * 32;
    const int kj = j0 + x < N ? j0 + x : N - 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = j0 + acc_row(r, h);
      t.bk[r] = biased ? a.buckets[((int64_t)b * N + (key < N ? key : N - 1)) * N + qi] : (unsigned char)0;
    }
    load_d16(Kp + (int64_t)kj * a.ld, t.ka);
#pragma unroll
    for (int r = 0; r < 16; ++r) {                    // A operand of O^T's K-step r: V[key row(r, h)][d = lane & 31]
      const int key = j0 + acc_row(r, h);
      const float v = V[(int64_t)(key < N ? key : N - 1) * a.ld + (x < dv ? x : 0)];
      t.va[r] = (x < dv && key < N) ? v : 0.0f;
    }
  };
  Tile cur, nxt;
  fetch(0, cur);
  for (int kt = 0; kt <= qt; ++kt) {                  // causal: key tiles up to the query tile
    const int j0 = kt * 32;
    if (kt < qt) fetch(kt + 1, nxt);
    // S^T = K_tile Q_tile^T : A = keys (rows), B = queries (columns), K axis = dqk
    hf32x16 S = {0};
#pragma unroll
    for (int s = 0; s < kAttnMaxSteps; ++s)
      if (s < dqk) S = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.ka[s], qb[s], S, 0, 0, 0);   // step s carries d = s and d = 16 + s
    // P^T[j][i] = silu(S + bias) / N for j <= i; register r of S^T is the B operand of K-step r of O^T += V^T P^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + acc_row(r, h);               // this lane's key for register r
      float sc = S[r];
      if (biased && j < N) sc += pos_s[N - 1 + j - qi] + tsw_s[cur.bk[r]];
      float pv = silu_fast(sc) * inv_n;
      if (j > qi || j >= N || i0 + x >= N) pv = 0.0f;
      O = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.va[r], pv, O, 0, 0, 0);
    }
    if (kt < qt) cur = nxt;
  }
  const int qrow = i0 + x;
  if (qrow >= N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = acc_row(r, h);
    if (d < dv) a.out[((int64_t)b * N + qrow) * ((int64_t)H * dv) + (int64_t)head * dv + d] = qrow < len ? O[r] : 0.0f;
  }
}

// The same attention with one WORKGROUP per (head, sequence): the head's K, V and Q rows (N x 32 floats each, zero-padded) are staged
// in LDS once and all query tiles of the sequence take their key-tile operands from there -- the per-tile global round trips of the
// one-wave kernel above (51 us per ML-20M block even with the next tile prefetched: < 2 waves per SIMD, nothing to hide them) are
// gone.  Four waves; the causal triangle is dealt in snake order over the tiles sorted by cost (tile t costs t + 1 key tiles:
// 7 tiles -> 7 units per wave).  Operand assignment and summation order are those of the one-wave kernel: bit-identical results.
// LDS rows are 36 floats apart (conflict-free ds_read_b128 of a lane's 16 consecutive d).
constexpr int kAttnLd = 36;

__global__ __launch_bounds__(256) void hstu_attention_wg_kernel(AttnArgs a) {
  // pos_w[NP + N, zero beyond 2N - 1] | ts_w[nb + 2, padded to 4] | K, V, Q [NP][36] each, rows >= N zero.  NP = N rounded up to whole
  // tiles: with the padding rows and bias slots in place the tile loops need NO per-element conditions (the first version guarded
  // every MFMA of the S chain and every bias lookup: ~150 branches and exec-mask regions per unit, nothing could be overlapped).
  extern __shared__ __attribute__((aligned(16))) float attn_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = lane & 31, h = lane >> 5;
  const int head = blockIdx.x, b = blockIdx.y;
  const int N = a.N, H = a.H, dqk = a.dqk, dv = a.dv;
  const int nqt = (N + 31) / 32, NP = nqt * 32;
  float* pos_s = attn_smem;
  float* tsw_s = pos_s + ((NP + N + 3) & ~3);
  float* Ks = tsw_s + ((a.num_buckets + 2 + 3) & ~3);
  float* Vs = Ks + (size_t)NP * kAttnLd;
  float* Qs = Vs + (size_t)NP * kAttnLd;
  const bool biased = a.buckets != nullptr;
  const float* base = a.uvqk + (int64_t)b * N * a.ld;
  const float* V = base + (int64_t)H * dv + (int64_t)head * dv;
  const float* Q = base + 2 * (int64_t)H * dv + (int64_t)head * dqk;
  const float* Kp = Q + (int64_t)H * dqk;
  for (int i = tid; i < NP + N; i += 256) pos_s[i] = (biased && i < 2 * N - 1) ? a.pos_w[i] : 0.0f;
  for (int i = tid; i <= a.num_buckets + 1; i += 256) tsw_s[i] = (biased && i <= a.num_buckets) ? a.ts_w[i] : 0.0f;
  // 32 consecutive threads = one row (coalesced).  Eight rows per thread are requested before the first is written: a rolled loop of
  // load -> wait -> LDS write cost one L2 round trip per row (27 of them at N = 211).
  for (int i0 = tid; i0 < NP * 32; i0 += 256 * 8) {
    float kv[8], vv[8], qv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256 < N * 32 ? i0 + u * 256 : N * 32 - 1;   // clamped (always valid) address
      const int r = i >> 5, d = i & 31;
      kv[u] = Kp[(int64_t)r * a.ld + (d < dqk ? d : 0)];
      qv[u] = Q[(int64_t)r * a.ld + (d < dqk ? d : 0)];
      vv[u] = V[(int64_t)r * a.ld + (d < dv ? d : 0)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      if (i < NP * 32) {
        const int r = i >> 5, d = i & 31;
        const bool real = i < N * 32;
        Ks[r * kAttnLd + d] = (real && d < dqk) ? kv[u] : 0.0f;
        Qs[r * kAttnLd + d] = (real && d < dqk) ? qv[u] : 0.0f;
        Vs[r * kAttnLd + d] = (real && d < dv) ? vv[u] : 0.0f;
      }
    }
  }
  __syncthreads();
  const int64_t len = a.lengths[b];
  const float inv_n = 1.0f / (float)N;
  const unsigned char* brow = biased ? a.buckets + (int64_t)b * N * N : nullptr;
  for (int i = 0; i < nqt; ++i) {
    if (((i & 7) < 4 ? (i & 7) : 7 - (i & 7)) != wave) continue;   // snake over the tiles in descending cost
    const int qt = nqt - 1 - i;
    const int i0 = qt * 32;
    const int qi = i0 + x;                              // this lane's query (column axis); rows >= N are zero rows
    const int qc = qi < N ? qi : N - 1;                 // ... clamped for the bucket matrix
    float qb[kAttnMaxSteps];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 f = *reinterpret_cast<const float4*>(Qs + qi * kAttnLd + 16 * h + 4 * q4);
      qb[4 * q4] = f.x; qb[4 * q4 + 1] = f.y; qb[4 * q4 + 2] = f.z; qb[4 * q4 + 3] = f.w;
    }
    hf32x16 O = {0};
    unsigned char bk[16], bkn[16];
    auto fetch_bk = [&](int kt, unsigned char (&o)[16]) {   // (key, query) time buckets of a key tile: L2-resident bytes, one tile ahead
      if (biased) {     // ONE branch per tile (written per element the compiler made it sixteen)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + acc_row(r, h);
          o[r] = brow[(int64_t)(key < N ? key : N - 1) * N + qc];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (unsigned char)(a.num_buckets + 1);   // slot nb + 1 holds 0
      }
    };
    fetch_bk(0, bk);
    for (int kt = 0; kt <= qt; ++kt) {
      const int j0 = kt * 32;
      if (kt < qt) fetch_bk(kt + 1, bkn);
      float ka[kAttnMaxSteps], va[16], bias[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 f = *reinterpret_cast<const float4*>(Ks + (j0 + x) * kAttnLd + 16 * h + 4 * q4);
        ka[4 * q4] = f.x; ka[4 * q4 + 1] = f.y; ka[4 * q4 + 2] = f.z; ka[4 * q4 + 3] = f.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = j0 + acc_row(r, h);
        va[r] = Vs[j * kAttnLd + x];
        bias[r] = pos_s[N - 1 + j - qc] + tsw_s[bk[r]];
      }
      hf32x16 S = {0};
#pragma unroll
      for (int s = 0; s < kAttnMaxSteps; ++s) S = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s], qb[s], S, 0, 0, 0);   // zero-padded beyond dqk
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = j0 + acc_row(r, h);
        const float t = silu_fast(S[r] + bias[r]) * inv_n;
        pv[r] = (j > qi || j >= N || qi >= N) ? 0.0f : t;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) O = __builtin_amdgcn_mfma_f32_32x32x2f32(va[r], pv[r], O, 0, 0, 0);
      if (kt < qt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bk[r] = bkn[r];
      }
    }
    if (qi < N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = acc_row(r, h);
        if (d < dv) a.out[((int64_t)b * N + qi) * ((int64_t)H * dv) + (int64_t)head * dv + d] = qi < len ? O[r] : 0.0f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused encoder for short sequences (seq_len <= 64, e.g. amzn-books: N = 61, D = 64, 8 heads x 8): ONE workgroup per
// sequence keeps the residual stream X, the uvqk activations Y and the attention output A in LDS (~103 KiB of the
// 160 KiB) and runs every block back to back in one launch -- sequences are independent, so nothing crosses workgroups.
// The multi-kernel path spends its time in per-kernel latency (83 launches of 5-20 us each for 16 blocks: 0.87 ms; a
// hipGraph replay does not help); here a block is five barrier-separated phases of a few hundred MFMAs.
//   LN1 -> NX | GEMM uvqk (+silu, padded rows zero) -> Y | attention per (head, query tile) -> A | LN2 * u -> A |
//   GEMM o (+bias, +X, padded rows zero) -> X
// 16 waves: phase work is dealt tile-wise (t = wave, wave + 16, ...).  LDS rows use odd strides (conflict-free column
// walks).  Same arithmetic as the multi-kernel kernels except the K order inside the GEMMs (pairs 2s, 2s+1).
// ---------------------------------------------------------------------------------------------
#ifdef RAILS_HSTU_PHASES   // tools/hstu_phases.sh: wall-clock stamps (100 MHz) of sequence 0, block 1
__device__ long long g_hphase[8];
#define RAILS_HPHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && blk == 1) g_hphase[i] = (long long)wall_clock64(); } while (0)
#else
#define RAILS_HPHASE(i)
#endif

struct FusedLayer { const float* uvqk; const float* o_w; const float* o_b; const float* ts_w; const float* pos_w; };

struct FusedArgs {
  const float* emb; const int64_t* ids; const int64_t* lengths; const unsigned char* buckets; const float* pos_emb;
  const FusedLayer* layers; int n_blocks;
  int B, N, D, H, dqk, dv, num_buckets, mode;
  float eps;
  float* out;      // (B, D) current embeddings
};

constexpr int kFusedThreads = 1024;
constexpr int kFusedWaves = kFusedThreads / 64;
constexpr int kFusedRows = 64;
constexpr int kFusedMaxK = 128;   // D and heads * dv <= 128

// LayerNorm (no affine) of the 64 LDS rows, optionally times `mul`: 16 lanes per row (a wave normalises its four rows
// together: 2 x 4 shuffle steps instead of 4 x 12 dependent ones: 3.1 -> 1.5 us per call).  src == dst allowed.
__device__ __forceinline__ void fused_layer_norm_rows(const float* src, int ss, float* dst, int ds, const float* mul, int ms, int dim,
                                                      float eps, int wave, int lane) {
  const int row = wave * 4 + (lane >> 4), sub = lane & 15;
  float sm = 0.0f;
  for (int k = sub; k < dim; k += 16) sm += src[row * ss + k];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
  const float mean = sm / (float)dim;
  float vr = 0.0f;
  for (int k = sub; k < dim; k += 16) { const float c = src[row * ss + k] - mean; vr = __builtin_fmaf(c, c, vr); }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) vr += __shfl_xor(vr, o, 64);
  const float rstd = 1.0f / sqrtf(vr / (float)dim + eps);
  for (int k = sub; k < dim; k += 16) {
    float y = (src[row * ss + k] - mean) * rstd;
    if (mul) y *= mul[row * ms + k];
    dst[row * ds + k] = y;
  }
}

__global__ __launch_bounds__(kFusedThreads) void hstu_fused_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  const int N = a.N, D = a.D, H = a.H, dqk = a.dqk, dv = a.dv;
  const int HV = H * dv, W = 2 * H * (dv + dqk);
  const int XS = D + 1, AS = (HV > D ? HV : D) + 1, YS = W + 1;      // odd row strides
  float* X = fsm;                                   // [64][XS]
  float* A = X + kFusedRows * XS;                   // [64][AS]  LN1 output, then attention output / o input
  float* Y = A + kFusedRows * AS;                   // [64][YS]  u | v | q | k
  float* pos_s = Y + kFusedRows * YS;               // [2N - 1]
  float* tsw_s = pos_s + 2 * kFusedRows;            // [num_buckets + 1]
  unsigned char* bk_s = reinterpret_cast<unsigned char*>(tsw_s + 132);   // [N][N] key-major
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 31, h = lane >> 5;
  const int64_t len = a.lengths[b];
  const bool biased = a.buckets != nullptr;
  const float scale = sqrtf((float)D);

  // ---- prologue: X = [id != 0, n < len] * (emb * sqrt(D) + pos_emb[n]); rows >= N zero; bucket matrix to LDS
  for (int i = tid; i < kFusedRows * D; i += kFusedThreads) {
    const int n = i / D, dd = i - n * D;
    float v = 0.0f;
    if (n < N && n < len && a.ids[(int64_t)b * N + n] != 0) v = a.emb[((int64_t)b * N + n) * D + dd] * scale + a.pos_emb[(int64_t)n * D + dd];
    X[n * XS + dd] = v;
  }
  if (biased)
    for (int i = tid; i < N * N; i += kFusedThreads) bk_s[i] = a.buckets[(int64_t)b * N * N + i];
  __syncthreads();

  const float inv_n = 1.0f / (float)N;
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    const FusedLayer L = a.layers[blk];
    if (biased) {
      for (int i = tid; i < 2 * N - 1; i += kFusedThreads) pos_s[i] = L.pos_w[i];
      for (int i = tid; i <= a.num_buckets; i += kFusedThreads) tsw_s[i] = L.ts_w[i];
    }
    RAILS_HPHASE(0);
    // ---- LN1: A[row][:D] = layer_norm(X[row])
    fused_layer_norm_rows(X, XS, A, AS, nullptr, 0, D, a.eps, wave, lane);
    __syncthreads();
    RAILS_HPHASE(1);
    // ---- GEMM uvqk: Y = silu(A[:, :D] Wuvqk), rows >= len zero
    for (int t = wave; t < 2 * (W / 32); t += kFusedWaves) {
      const int mt = t / (W / 32), nt = t - mt * (W / 32);
      // the tile's weight column first (one batch of loads; inside the MFMA loop every step waited for its own L2 trip)
      hf32x16 acc = {0};
      for (int k0 = 0; k0 < D; k0 += 64) {     // groups of 32 K-steps: 32 registers (1024 threads leave 128 per lane)
        float bw[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          const int k = k0 + 2 * s + h;
          bw[s] = L.uvqk[(int64_t)(k < D ? k : D - 1) * W + nt * 32 + x];
        }
#pragma unroll
        for (int s = 0; s < 32; ++s)
          if (k0 + 2 * s < D) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(mt * 32 + x) * AS + k0 + 2 * s + h], bw[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + acc_row(r, h);
        Y[row * YS + nt * 32 + x] = row < len && row < N ? silu_fast(acc[r]) : 0.0f;
      }
    }
    __syncthreads();
    RAILS_HPHASE(2);
    // ---- attention: A[i][head*dv + d] = sum_{j <= i} silu(q_i . k_j + bias) / N * v_j
    for (int t = wave; t < 2 * H; t += kFusedWaves) {
      const int head = t % H, qt = t / H;
      const int i0 = qt * 32;
      const int qi = i0 + x;                                  // < 64 always
      const float* Vc = Y + HV + head * dv;
      const float* Qc = Y + 2 * HV + head * dqk;
      const float* Kc = Qc + H * dqk;
      hf32x16 O = {0};
      for (int kt = 0; kt <= qt; ++kt) {
        const int j0 = kt * 32;
        hf32x16 S = {0};
        for (int s = 0; s < (dqk + 1) / 2; ++s) {
          const int d = 2 * s + h;
          const float ka = d < dqk ? Kc[(j0 + x) * YS + d] : 0.0f;
          const float qb = d < dqk ? Qc[qi * YS + d] : 0.0f;
          S = __builtin_amdgcn_mfma_f32_32x32x2f32(ka, qb, S, 0, 0, 0);
        }
        float pv[16], va[16];      // all 16 probabilities / V operands first (independent LDS lookups overlap), then the MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + acc_row(r, h);
          float sc = S[r];
          if (biased && j < N && qi < N) sc += pos_s[N - 1 + j - qi] + tsw_s[bk_s[j * N + qi]];
          float p = silu_fast(sc) * inv_n;
          if (j > qi || j >= N || qi >= N) p = 0.0f;
          pv[r] = p;
          va[r] = x < dv ? Vc[j * YS + x] : 0.0f;               // rows >= N of Y are zero
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) O = __builtin_amdgcn_mfma_f32_32x32x2f32(va[r], pv[r], O, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = acc_row(r, h);
        if (d < dv) A[qi * AS + head * dv + d] = (qi < len && qi < N) ? O[r] : 0.0f;
      }
    }
    __syncthreads();
    RAILS_HPHASE(3);
    // ---- LN2 * u: A[row][:HV] = layer_norm(A[row][:HV]) * Y[row][:HV]
    fused_layer_norm_rows(A, AS, A, AS, Y, YS, HV, a.eps, wave, lane);
    __syncthreads();
    RAILS_HPHASE(4);
    // ---- GEMM o: X = A[:, :HV] Wo^T + bo + X, rows >= len zero
    for (int t = wave; t < 2 * (D / 32); t += kFusedWaves) {
      const int mt = t / (D / 32), nt = t - mt * (D / 32);
      hf32x16 acc = {0};
      for (int k0 = 0; k0 < HV; k0 += 64) {
        float bw[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          const int k = k0 + 2 * s + h;
          bw[s] = L.o_w[(int64_t)(nt * 32 + x) * HV + (k < HV ? k : HV - 1)];
        }
#pragma unroll
        for (int s = 0; s < 32; ++s)
          if (k0 + 2 * s < HV) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(mt * 32 + x) * AS + k0 + 2 * s + h], bw[s], acc, 0, 0, 0);
      }
      const float bias = L.o_b[nt * 32 + x];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + acc_row(r, h);
        const float v = acc[r] + bias + X[row * XS + nt * 32 + x];
        X[row * XS + nt * 32 + x] = row < len && row < N ? v : 0.0f;
      }
    }
    __syncthreads();
    RAILS_HPHASE(5);
  }
  // ---- postprocessor on row len - 1
  if (wave == 0) {
    const int row = (int)len - 1;
    if (a.mode == 0) {
      float sm = 0.0f;
      for (int k = lane; k < D; k += 64) sm += X[row * XS + k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
      const float mean = sm / (float)D;
      float vr = 0.0f;
      for (int k = lane; k < D; k += 64) { const float c = X[row * XS + k] - mean; vr = __builtin_fmaf(c, c, vr); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) vr += __shfl_xor(vr, o, 64);
      const float rstd = 1.0f / sqrtf(vr / (float)D + a.eps);
      for (int k = lane; k < D; k += 64) a.out[(int64_t)b * D + k] = (X[row * XS + k] - mean) * rstd;
    } else {
      float vr = 0.0f;
      for (int k = lane; k < D; k += 64) vr = __builtin_fmaf(X[row * XS + k], X[row * XS + k], vr);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) vr += __shfl_xor(vr, o, 64);
      const float nrm = fmaxf(sqrtf(vr), a.eps);
      for (int k = lane; k < D; k += 64) a.out[(int64_t)b * D + k] = X[row * XS + k] / nrm;
    }
  }
}

// 1 if the fused kernel handles this geometry
bool hstu_fused_supported(int N, int D, int H, int dqk, int dv, int num_buckets) {
  const int HV = H * dv, W = 2 * H * (dv + dqk);
  if (N < 1 || N > kFusedRows || D % 32 != 0 || D > 128 || HV % 32 != 0 || HV > 128 || W % 32 != 0 || W > 512) return false;
  if (dqk > 32 || dv > 32 || num_buckets > 128) return false;
  const int XS = D + 1, AS = (HV > D ? HV : D) + 1, YS = W + 1;
  const size_t lds = sizeof(float) * ((size_t)kFusedRows * (XS + AS + YS) + 2 * kFusedRows + 132) + (size_t)kFusedRows * kFusedRows;
  return lds <= 150 * 1024;
}

int hstu_encode_fused(const float* emb, const int64_t* ids, const int64_t* lengths, const unsigned char* buckets, const float* pos_emb,
                      const void* layers, int n_blocks, int B, int N, int D, int H, int dqk, int dv, int num_buckets, int mode,
                      float eps, float* out, hipStream_t stream) {
  if (B == 0) return kOk;
  if (!hstu_fused_supported(N, D, H, dqk, dv, num_buckets)) { set_error("hstu_encode_fused: geometry not supported"); return kErrUnsupported; }
  const int HV = H * dv, W = 2 * H * (dv + dqk);
  const int XS = D + 1, AS = (HV > D ? HV : D) + 1, YS = W + 1;
  const size_t lds = sizeof(float) * ((size_t)kFusedRows * (XS + AS + YS) + 2 * kFusedRows + 132) + (size_t)kFusedRows * kFusedRows;
  static DynLdsOnce once;
  if (ensure_dyn_lds(once, reinterpret_cast<const void*>(&hstu_fused_kernel), 150 * 1024) != kOk) return kErrLaunch;
  FusedArgs a{emb, ids, lengths, buckets, pos_emb, static_cast<const FusedLayer*>(layers), n_blocks, B, N, D, H, dqk, dv, num_buckets, mode, eps, out};
  hipLaunchKernelGGL(hstu_fused_kernel, dim3(B), dim3(kFusedThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
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

// out[r][j] = act(t[r][j]) * t[r][F + j]: the gate of GeGLU (erf gelu) / SwiGLU (x * sigmoid x) over the (rows, 2F) pre-activations
// t = x W + b written by gemm_f32 (reference rails/similarities/layers.py:36-43, :68-74).  Precise erff / expf, a true division.
__global__ __launch_bounds__(256) void glu_gate_kernel(const float* __restrict__ t, int64_t ldt, int64_t rows, int F, int kind,
                                                       float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * F) return;
  const int64_t r = i / F;
  const int j = (int)(i - r * F);
  const float l = t[r * ldt + j], g = t[r * ldt + F + j];
  const float act = kind == RAILS_GEGLU ? 0.5f * l * (1.0f + erff(l * 0.70710678118654752440f)) : l / (1.0f + expf(-l));
  out[i] = act * g;
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
  // RAILS_GEMM: 0 / unset = choose, 1 = per-wave kernel, 2 = tiled kernel where its alignment conditions hold (measurement override)
  static const int forced = [] { const char* e = getenv("RAILS_GEMM"); return e ? atoi(e) : 0; }();
  const bool aligned = K % 32 == 0 && K >= 32 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                       (w_is_nk ? true : (N % 4 == 0 && N >= 4));
  if (aligned && forced != 1 && (forced == 2 || (M >= 256 && N >= 64))) {
    const int64_t wgs = ((M + 63) / 64) * ((N + 63) / 64);
    hipLaunchKernelGGL(gemm_f32_tiled_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, g);
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
  }
  const int64_t tiles = ((M + 31) / 32) * ((N + 31) / 32);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, g);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- stand-alone MoLGatingFn / SoftmaxDropoutCombiner (reference rails/similarities/mol/similarity_fn.py:31-46, :148-201) ----
// One wave per (query, item) row r = b * X + x:  g = gq[b] * gi + gqi (glu_silu) or the sum of the parts that exist (none);
// w = g * sigmoid(g) (glu_silu) or g;  pi = softmax(w) [/ clamp(sum pi, eps) when the combiner's dropout rate is > 0];
// out[r] = sum_l pi[l] * y[r][l].  Precise expf and true divisions: this is the module API for callers that use the pieces on
// their own, not the scoring path (there all of it is fused into the scoring kernels).
__global__ __launch_bounds__(256) void gate_combine_kernel(const float* __restrict__ y, int64_t ldy, const float* __restrict__ gqi, int64_t ldq,
                                                          const float* __restrict__ gq, const float* __restrict__ gi, int64_t rows, int X,
                                                          int L, int gi_per_row, int glu_silu, int renorm, float eps, float* __restrict__ out,
                                                          float* __restrict__ pi_out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int64_t b = r / X, x = r - b * X;
  const float* gqr = gq ? gq + b * L : nullptr;
  const float* gir = gi ? gi + (gi_per_row ? r : x) * (int64_t)L : nullptr;
  constexpr int kMaxPerLane = 16;   // L <= 1024
  float w[kMaxPerLane];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int l = lane + 64 * i;
    float g = -INFINITY;
    if (l < L) {
      const float a = gqr ? gqr[l] : 0.0f, c = gir ? gir[l] : 0.0f, d = gqi ? gqi[r * ldq + l] : 0.0f;
      if (glu_silu) {
        g = a * c + d;
        g = g * (1.0f / (1.0f + expf(-g)));
      } else {
        g = (gqr ? a : 0.0f) + (gir ? c : 0.0f) + (gqi ? d : 0.0f);
      }
    }
    w[i] = g;
    mx = fmaxf(mx, g);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float den = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    w[i] = lane + 64 * i < L ? expf(w[i] - mx) : 0.0f;
    den += w[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o, 64);
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) { w[i] = w[i] / den; sum += w[i]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float norm = renorm ? fmaxf(sum, eps) : 1.0f;
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int l = lane + 64 * i;
    if (l < L) {
      const float pi = renorm ? w[i] / norm : w[i];
      if (pi_out) pi_out[r * L + l] = pi;
      acc = __builtin_fmaf(pi, y[r * ldy + l], acc);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) out[r] = acc;
}

int gate_combine(const float* y, int64_t ldy, const float* gqi, int64_t ldq, const float* gq, const float* gi, int64_t rows, int X, int L,
                 int gi_per_row, int glu_silu, int renorm, float eps, float* out, float* pi_out, hipStream_t stream) {
  if (rows == 0) return kOk;
  if (L > 1024) { set_error("gate_combine: %d logits (supported: <= 1024)", L); return kErrUnsupported; }
  hipLaunchKernelGGL(gate_combine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, y, ldy, gqi, ldq, gq, gi, rows, X, L, gi_per_row,
                     glu_silu, renorm, eps, out, pi_out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int glu_gate(const float* t, int64_t ldt, int64_t rows, int F, int kind, float* out, hipStream_t stream) {
  if (rows == 0 || F == 0) return kOk;
  hipLaunchKernelGGL(glu_gate_kernel, dim3((unsigned)((rows * F + 255) / 256)), dim3(256), 0, stream, t, ldt, rows, F, kind, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int hstu_time_buckets(const int64_t* timestamps, int B, int N, const int64_t* thresholds, int num_buckets, unsigned char* out,
                      hipStream_t stream) {
  const int64_t total = (int64_t)B * N * N;
  if (total == 0) return kOk;
  if (num_buckets > 255) { set_error("hstu_time_buckets: %d buckets do not fit a byte", num_buckets); return kErrUnsupported; }
  hipLaunchKernelGGL(hstu_time_buckets_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, timestamps, B, N, thresholds,
                     num_buckets, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int hstu_attention(const float* uvqk, int64_t ld, int B, int N, int H, int dqk, int dv, const int64_t* lengths,
                   const unsigned char* buckets, const float* ts_w, const float* pos_w, int num_buckets, float* out,
                   hipStream_t stream) {
  if (B == 0 || N == 0) return kOk;
  if (dv > 32) { set_error("hstu_attention: dv = %d (supported: <= 32)", dv); return kErrUnsupported; }
  if (dqk > 2 * kAttnMaxSteps) { set_error("hstu_attention: dqk = %d (supported: <= %d)", dqk, 2 * kAttnMaxSteps); return kErrUnsupported; }
  AttnArgs a{uvqk, ld, B, N, H, dqk, dv, lengths, buckets, ts_w, pos_w, num_buckets, out};
  const size_t lds = sizeof(float) * ((size_t)2 * N + num_buckets + 2);
  if (lds > 60 * 1024) { set_error("hstu_attention: seq_len = %d does not fit LDS", N); return kErrUnsupported; }
  // RAILS_ATTN: 0 / unset = choose, 1 = one wave per (query tile, head, sequence), 2 = one workgroup per (head, sequence) with K / V in LDS
  static const int forced = [] { const char* e = getenv("RAILS_ATTN"); return e ? atoi(e) : 0; }();
  const int NP = (N + 31) / 32 * 32;
  const size_t lds_wg = sizeof(float) * ((size_t)((NP + N + 3) & ~3) + ((num_buckets + 2 + 3) & ~3) + (size_t)3 * NP * kAttnLd);
  if (forced != 1 && lds_wg <= 150 * 1024 && (forced == 2 || N > 64)) {
    static DynLdsOnce once;
    if (ensure_dyn_lds(once, reinterpret_cast<const void*>(&hstu_attention_wg_kernel), 150 * 1024) != kOk) return kErrLaunch;
    hipLaunchKernelGGL(hstu_attention_wg_kernel, dim3(H, B), dim3(256), lds_wg, stream, a);
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
  }
  hipLaunchKernelGGL(hstu_attention_kernel, dim3((N + 31) / 32, H, B), dim3(64), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

#ifdef RAILS_HSTU_PHASES
}  // namespace mol
extern "C" int rails_debug_hstu_phases(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mol::g_hphase), sizeof(long long) * 8) == hipSuccess ? 0 : -1;
}
namespace mol {
#endif
}  // namespace mol
