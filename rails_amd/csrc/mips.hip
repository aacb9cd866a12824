// Dot-product (MIPS) scoring: logits[b, x] = <q[b, :], X[x, :]> in fp32.
// Replaces torch.mm(query_embeddings, item_embeddings_t) of MIPSBruteForceTopK.forward
// (rails/indexing/mips_top_k.py:56-81) and DotProductSimilarity.forward's shared-corpus branch
// (rails/similarities/dot_product_similarity_fn.py:48-54).
//
// Same operand scheme as GEMM1 of the MoL kernel: 32 queries on the MFMA row axis, 32 items on the column axis,
// K = D (zero padded to a multiple of 8) walked two at a time by v_mfma_f32_32x32x2_f32; items are stored
// once in fragment order so each operand fetch is 1 KiB contiguous per wave.  At B = 32 the scan moves
// 4*D bytes per item against 64*D flops: 16 flop/B, right at the fp32-MFMA / HBM crossover (157 TF / 8 TB/s = 20).
#include <hip/hip_runtime.h>

#include "mol_kernels.h"
#include "mol_layout.h"

namespace mol {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// items (n, D) row-major -> tiles of 32 items, [sc in Dp/8][lane][4]: lane (x, hi) holds X[x][hi*Dp/2 + 4sc + j]
__global__ void mips_pack_items_kernel(const float* __restrict__ items, int64_t n, int D, int Dp, float* __restrict__ out) {
  const int64_t tile = blockIdx.x;
  const int per_tile = kTileItems * Dp;
  for (int i = threadIdx.x; i < per_tile; i += blockDim.x) {
    const int j = i & 3, lane = (i >> 2) & 63, sc = i >> 8;
    const int x = lane & 31, hi = lane >> 5;
    const int k = hi * (Dp / 2) + 4 * sc + j;
    const int64_t item = tile * kTileItems + x;
    out[tile * per_tile + i] = (item < n && k < D) ? items[item * D + k] : 0.0f;
  }
}

// queries (B, D) -> groups of 32 rows, [g][sc][lane][4]: lane (row, hi) holds q[g*32 + row][hi*Dp/2 + 4sc + j]
__global__ void mips_pack_queries_kernel(const float* __restrict__ q, int B, int D, int Dp, float* __restrict__ out) {
  const int g = blockIdx.x;
  const int per_group = 32 * Dp;
  for (int i = threadIdx.x; i < per_group; i += blockDim.x) {
    const int j = i & 3, lane = (i >> 2) & 63, sc = i >> 8;
    const int row = lane & 31, hi = lane >> 5;
    const int k = hi * (Dp / 2) + 4 * sc + j;
    const int b = g * 32 + row;
    out[g * per_group + i] = (b < B && k < D) ? q[(int64_t)b * D + k] : 0.0f;
  }
}

__global__ __launch_bounds__(256) void mips_score_kernel(const float4* __restrict__ qfrag, const float4* __restrict__ ifrag,
                                                        int B, int n_groups, int64_t n, int64_t n_tiles, int Dp,
                                                        float* __restrict__ logits, int64_t ld) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, x = lane & 31;
  const int nsc = Dp / 8;
  const int64_t n_units = n_tiles * n_groups;
  for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < n_units; u += (int64_t)gridDim.x * 4) {
    const int64_t tile = u / n_groups;
    const int g = (int)(u - tile * n_groups);
    const float4* a4 = qfrag + (int64_t)g * nsc * 64;
    const float4* b4 = ifrag + tile * nsc * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int sc = 0; sc < nsc; ++sc) {
      const float4 a = a4[sc * 64 + lane], b = b4[sc * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    const int64_t item = tile * kTileItems + x;
    if (item < n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = g * 32 + acc_row(r, hi);
        if (b < B) logits[(int64_t)b * ld + item] = acc[r];
      }
    }
  }
}

int mips_pack_items(const float* items, int64_t n, int D, float* out, hipStream_t stream) {
  const int Dp = (D + 7) / 8 * 8;
  const int64_t tiles = num_tiles(n);
  if (tiles == 0) return kOk;
  hipLaunchKernelGGL(mips_pack_items_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, items, n, D, Dp, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int mips_score(const float* q, int B, int D, const float* ifrag, int64_t n, float* qfrag_ws, float* logits, int64_t ld,
               int n_cu, hipStream_t stream) {
  const int Dp = (D + 7) / 8 * 8;
  const int n_groups = (B + 31) / 32;
  const int64_t tiles = num_tiles(n);
  if (B <= 0 || tiles == 0) return kOk;
  hipLaunchKernelGGL(mips_pack_queries_kernel, dim3(n_groups), dim3(256), 0, stream, q, B, D, Dp, qfrag_ws);
  int64_t grid = (tiles * n_groups + 3) / 4;
  if (grid > (int64_t)n_cu * 8) grid = (int64_t)n_cu * 8;
  hipLaunchKernelGGL(mips_score_kernel, dim3((unsigned)grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(qfrag_ws),
                     reinterpret_cast<const float4*>(ifrag), B, n_groups, n, tiles, Dp, logits, ld);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace mol

namespace mol {
// Per-row candidates: out[bq, x] = <q[bq, :], items[bq / r, x, :]>  (DotProductSimilarity.forward's bmm branches,
// rails/similarities/dot_product_similarity_fn.py:55-68).  Small training/rerank-sized problems: one thread per output.
__global__ void dot_rowwise_kernel(const float* __restrict__ q, const float* __restrict__ items, int64_t total, int X, int D,
                                   int r, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t bq = i / X;
  const int x = (int)(i - bq * X);
  const float* qv = q + bq * D;
  const float* xv = items + ((bq / r) * X + x) * (int64_t)D;
  float acc = 0.0f;
  for (int k = 0; k < D; ++k) acc = __builtin_fmaf(qv[k], xv[k], acc);
  out[i] = acc;
}

int dot_rowwise(const float* q, const float* items, int64_t Bq, int X, int D, int r, float* out, hipStream_t stream) {
  const int64_t total = Bq * X;
  if (total == 0) return kOk;
  hipLaunchKernelGGL(dot_rowwise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, q, items, total, X, D, r, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}
}  // namespace mol
