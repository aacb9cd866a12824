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

#include "mol_kernels.h"
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

// scores[b][x] for all b of one item per thread; query sums staged in LDS
__global__ __launch_bounds__(256) void coarse_score_kernel(const float* __restrict__ eq, int B, int PQ, int d, int avg,
                                                          const unsigned short* __restrict__ table, int64_t n,
                                                          float* __restrict__ scores, int64_t ld) {
  extern __shared__ __attribute__((aligned(16))) float qs[];  // [B][d]
  for (int i = threadIdx.x; i < B * d; i += blockDim.x) {
    const int b = i / d, dd = i - b * d;
    float acc = 0.0f;
    for (int p = 0; p < PQ; ++p) acc += eq[((int64_t)b * PQ + p) * d + dd];
    qs[i] = bf16_rn(avg ? acc / (float)PQ : acc);
  }
  __syncthreads();
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) {
    const uint4* row = reinterpret_cast<const uint4*>(table + x * d);
    for (int b0 = 0; b0 < B; b0 += 8) {  // 8 accumulators per pass over the row (row stays in L1/registers)
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < d / 8; ++c) {
        const uint4 v = row[c];
        const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = bf16_to_f32((unsigned short)((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu));
#pragma unroll
          for (int bb = 0; bb < 8; ++bb)
            if (b0 + bb < B) acc[bb] = __builtin_fmaf(qs[(b0 + bb) * d + c * 8 + j], t, acc[bb]);
        }
      }
#pragma unroll
      for (int bb = 0; bb < 8; ++bb)
        if (b0 + bb < B) scores[(int64_t)(b0 + bb) * ld + x] = bf16_rn(acc[bb]);
    }
  }
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
                 hipStream_t stream) {
  const int d = s.dot_product_dimension;
  if (B <= 0 || n <= 0) return kOk;
  const size_t lds = sizeof(float) * (size_t)B * d;
  if (lds > 64 * 1024) { set_error("coarse_score: batch %d x d %d does not fit LDS", B, d); return kErrUnsupported; }
  int64_t grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(coarse_score_kernel, dim3((unsigned)grid), dim3(256), lds, stream, eq, B, s.query_dot_product_groups,
                     d, avg, static_cast<const unsigned short*>(table), n, scores, ld);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}


// ---------------------------------------------------------------------------------------------
// Per-component candidate generation of MoLNaiveTopK / MoLCombTopK (reference rails/indexing/mol_top_k.py:
// component table :61-73 and :172-174, scoring :242-255 / :495-506).  For every query group i and item group m the
// reference scores  bf16(Eq[b,i,:]) . bf16(Ex[x,m,:])  with a bf16 mm and takes the top k_per_group per (b, m) row.
//   table[x][m][:] = bf16(Ex[x,m,:])                                     2*P_X*d bytes per item, item-major
//   score[(b*P_Q + i)*P_X + m][x] = bf16( sum_d bf16(Eq[b,i,d]) * table[x][m][d] )   fp32 holding bf16 values
// Row order (b, i, m) makes the top-k output reshape to (B, P_Q*P_X*k_g) directly.
// ---------------------------------------------------------------------------------------------
__global__ void component_build_kernel(const float* __restrict__ ipack, int64_t n, int PQ, int PX, int d,
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
  table[i] = bf16_bits(tEx[((m * (d / 8) + (s >> 2)) * 64 + hi * 32 + x) * 4 + (s & 3)]);
}

__global__ __launch_bounds__(256) void component_score_kernel(const float* __restrict__ eq, int B, int PQ, int PX, int d,
                                                             const unsigned short* __restrict__ table, int64_t n,
                                                             float* __restrict__ scores, int64_t ld) {
  extern __shared__ __attribute__((aligned(16))) float qs[];  // [B*PQ][d], bf16-rounded
  for (int i = threadIdx.x; i < B * PQ * d; i += blockDim.x) qs[i] = bf16_rn(eq[i]);
  __syncthreads();
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) {
    for (int m = 0; m < PX; ++m) {
      const uint4* row = reinterpret_cast<const uint4*>(table + (x * PX + m) * d);
      for (int r0 = 0; r0 < B * PQ; r0 += 8) {  // eight (b, i) rows per pass over the component
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < d / 8; ++c) {
          const uint4 v = row[c];
          const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float t = bf16_to_f32((unsigned short)((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu));
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
              if (r0 + rr < B * PQ) acc[rr] = __builtin_fmaf(qs[(r0 + rr) * d + c * 8 + j], t, acc[rr]);
          }
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
          if (r0 + rr < B * PQ) scores[((int64_t)(r0 + rr) * PX + m) * ld + x] = bf16_rn(acc[rr]);
      }
    }
  }
}

int component_build(const Shape& s, const float* ipack, int64_t n, void* table, hipStream_t stream) {
  const int64_t total = n * s.item_dot_product_groups * s.dot_product_dimension;
  if (total == 0) return kOk;
  hipLaunchKernelGGL(component_build_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ipack, n,
                     s.query_dot_product_groups, s.item_dot_product_groups, s.dot_product_dimension,
                     static_cast<unsigned short*>(table));
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int component_score(const Shape& s, const float* eq, int B, const void* table, int64_t n, float* scores, int64_t ld,
                    hipStream_t stream) {
  const int d = s.dot_product_dimension, PQ = s.query_dot_product_groups;
  if (B <= 0 || n <= 0) return kOk;
  const size_t lds = sizeof(float) * (size_t)B * PQ * d;
  if (lds > 64 * 1024) { set_error("component_score: batch %d x P_Q %d x d %d does not fit LDS", B, PQ, d); return kErrUnsupported; }
  int64_t grid = (n + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(component_score_kernel, dim3((unsigned)grid), dim3(256), lds, stream, eq, B, PQ,
                     s.item_dot_product_groups, d, static_cast<const unsigned short*>(table), n, scores, ld);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace mol
