// Microbenchmark (round 4): v_mfma_f32_16x16x4_f32 issue rate per SIMD with 1 / 2 / 4 waves per SIMD, with and without
// the small-unit kernel's companions: one ds_read_b128 per 4 MFMAs (A operands from LDS), chains of CH dependent MFMAs,
// and NV independent VALU ops (fma or exp2) per MFMA.  Prints ns per MFMA per SIMD (32 cycles @ 2.4 GHz = 13.3 ns).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma16_waves.hip -o tools/ubench_mfma16_waves && tools/ubench_mfma16_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ACC independent accumulators, visited round-robin in runs of CH consecutive MFMAs on the same accumulator
template <int ACC, int CH, int LDS, int NV, int TRANS>
__global__ void k(float* out, int iters, float seed) {
  __shared__ float4 w[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) w[i] = make_float4(seed, seed * 0.5f, seed * 0.25f, 1.0f);
  __syncthreads();
  f32x4 acc[ACC];
  for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
  float b = seed * 0.5f;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < ACC; ++m) {
      float4 a = LDS ? w[((it * ACC + m) & 15) * 64 + lane] : make_float4(seed, b, seed, b);
      static_assert(CH == 4 || CH == 1, "");
      if (CH == 4) {
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b, acc[m], 0, 0, 0);
      } else {
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, acc[m], 0, 0, 0);
        acc[(m + 1) % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, acc[(m + 1) % ACC], 0, 0, 0);
        acc[(m + 2) % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b, acc[(m + 2) % ACC], 0, 0, 0);
        acc[(m + 3) % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b, acc[(m + 3) % ACC], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NV * 4; ++j) {
        if (TRANS) v[j % 8] = __builtin_amdgcn_exp2f(v[j % 8]);
        else v[j % 8] = __builtin_fmaf(v[j % 8], 1.0001f, 0.5f);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC, int CH, int LDS, int NV, int TRANS>
void run(int waves_per_simd, const char* what) {
  float* d; hipMalloc(&d, 1 << 22);
  const int threads = 256 * waves_per_simd;   // one workgroup per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  hipLaunchKernelGGL((k<ACC, CH, LDS, NV, TRANS>), dim3(256), dim3(threads), 0, 0, d, 50, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ACC, CH, LDS, NV, TRANS>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / (iters * 4.0 * ACC * waves_per_simd);
  printf("%-34s acc=%d chain=%d lds=%d valu/mfma=%d%s waves/SIMD=%d : %.2f ns per MFMA per SIMD (%.1f cycles @2.4GHz)\n", what, ACC, CH, LDS, NV,
         TRANS ? "(exp2)" : "(fma)", waves_per_simd, ns, ns * 2.4);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<8, 1, 0, 0, 0>(w, "independent, operands in regs");
    run<8, 4, 0, 0, 0>(w, "chains of 4, operands in regs");
    run<8, 4, 1, 0, 0>(w, "chains of 4, A from LDS");
    run<8, 1, 1, 0, 0>(w, "independent, A from LDS");
    run<8, 4, 1, 1, 0>(w, "chains of 4, LDS, 1 fma per MFMA");
    run<8, 4, 1, 1, 1>(w, "chains of 4, LDS, 1 exp2 per MFMA");
    run<8, 4, 0, 1, 0>(w, "chains of 4, regs, 1 fma per MFMA");
    run<8, 4, 0, 1, 1>(w, "chains of 4, regs, 1 exp2 per MFMA");
  }
  return 0;
}
