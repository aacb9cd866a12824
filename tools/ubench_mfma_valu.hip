// Microbenchmark: does v_mfma_f32_32x32x2_f32 co-execute with independent VALU work on gfx950?
// Each iteration issues 4 independent MFMAs (4 accumulators) and NV independent VALU ops per MFMA.
// Prints cycles per MFMA for NV in {0,2,4,8,16}, at 1 and 2 waves per SIMD, for fma and for v_exp.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int TRANS>
__global__ void k(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
  float a = seed, b = seed * 0.5f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (TRANS) v[j % 16] = __builtin_amdgcn_exp2f(v[j % 16]);
        else v[j % 16] = __builtin_fmaf(v[j % 16], 1.0001f, 0.5f);
      }
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 4.0f);
}

template <int NV, int TRANS>
void run(int waves_per_simd) {
  float* d; hipMalloc(&d, 1 << 20);
  const int threads = 256 * waves_per_simd;  // one workgroup per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL((k<NV, TRANS>), dim3(256), dim3(threads), 0, 0, d, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, TRANS>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  // per-SIMD: waves_per_simd waves each issue iters*4 MFMAs
  double ns_per_mfma_simd = ms * 1e6 / (iters * 4.0 * waves_per_simd);
  printf("%s NV=%2d waves/SIMD=%d : %.1f ns per MFMA per SIMD (64 cyc @2.4GHz = 26.7 ns)  wave-clock %.0f per MFMA\n",
         TRANS ? "exp2" : "fma ", NV, waves_per_simd, ns_per_mfma_simd, cyc);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>(w); run<2, 0>(w); run<4, 0>(w); run<8, 0>(w); run<16, 0>(w);
    run<2, 1>(w); run<4, 1>(w); run<8, 1>(w);
  }
  return 0;
}
