// Probes of the matrix instructions' arithmetic (rails_mfma_probe_f16 / _f32, include/rails_amd.h): one instruction per wave on
// caller-supplied operands, the accumulator returned as it is.  The a-priori error bound of the proved exact top-k
// (rails_amd/f16x3_bound.py, hypotheses H1 and H2) models these two instructions; the tests measure the model on the part.
#include <hip/hip_runtime.h>

#include "mol_kernels.h"
#include "mol_layout.h"

namespace mol {

typedef _Float16 h8p __attribute__((ext_vector_type(8)));
typedef float f32x16p __attribute__((ext_vector_type(16)));

// D = C + A B with v_mfma_f32_32x32x16_f16: A (32 x 16) and B (16 x 32) row-major f16 bit patterns, C / D (32 x 32) row-major fp32.
// Lane (x = lane & 31, hi = lane >> 5) supplies A[x][8 hi .. 8 hi + 8) and B[8 hi .. 8 hi + 8)[x]; accumulator register r of the
// lane is D[acc_row(r, hi)][x] (mol_layout.h).
__global__ __launch_bounds__(64) void mfma_probe_f16_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                            const float* __restrict__ c, float* __restrict__ d) {
  const int64_t t = blockIdx.x;
  const int lane = threadIdx.x, x = lane & 31, hi = lane >> 5;
  a += t * 32 * 16; b += t * 16 * 32; c += t * 32 * 32; d += t * 32 * 32;
  h8p av, bv;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    av[j] = __builtin_bit_cast(_Float16, a[x * 16 + 8 * hi + j]);
    bv[j] = __builtin_bit_cast(_Float16, b[(8 * hi + j) * 32 + x]);
  }
  f32x16p acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = c[acc_row(r, hi) * 32 + x];
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) d[acc_row(r, hi) * 32 + x] = acc[r];
}

// D = C + A B with v_mfma_f32_32x32x2_f32: A (32 x 2), B (2 x 32), C / D (32 x 32), row-major fp32; lane (x, hi) supplies A[x][hi], B[hi][x]
__global__ __launch_bounds__(64) void mfma_probe_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ c, float* __restrict__ d) {
  const int64_t t = blockIdx.x;
  const int lane = threadIdx.x, x = lane & 31, hi = lane >> 5;
  a += t * 32 * 2; b += t * 2 * 32; c += t * 32 * 32; d += t * 32 * 32;
  f32x16p acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = c[acc_row(r, hi) * 32 + x];
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x * 2 + hi], b[hi * 32 + x], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) d[acc_row(r, hi) * 32 + x] = acc[r];
}

// The kernels' own scalar arithmetic on caller-supplied values (hypothesis H3): out[0][i] = v_exp_f32(x[i]), out[1][i] = v_rcp_f32(x[i]),
// out[2][i] = x / (1 + 2^x) exactly as the scoring kernels spell it (exp2, add, rcp, mul)
__global__ void scalar_probe_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  out[i] = __builtin_amdgcn_exp2f(v);
  out[n + i] = __builtin_amdgcn_rcpf(v);
  const float e = __builtin_amdgcn_exp2f(v) + 1.0f;
  out[2 * n + i] = v * __builtin_amdgcn_rcpf(e);
}

int mfma_probe_f16(const unsigned short* a, const unsigned short* b, const float* c, float* d, int64_t n, hipStream_t stream) {
  if (n <= 0) return kOk;
  hipLaunchKernelGGL(mfma_probe_f16_kernel, dim3((unsigned)n), dim3(64), 0, stream, a, b, c, d);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int mfma_probe_f32(const float* a, const float* b, const float* c, float* d, int64_t n, hipStream_t stream) {
  if (n <= 0) return kOk;
  hipLaunchKernelGGL(mfma_probe_f32_kernel, dim3((unsigned)n), dim3(64), 0, stream, a, b, c, d);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int scalar_probe(const float* x, int64_t n, float* out, hipStream_t stream) {
  if (n <= 0) return kOk;
  hipLaunchKernelGGL(scalar_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, n, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace mol
