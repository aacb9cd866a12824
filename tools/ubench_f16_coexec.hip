// Microbenchmarks behind the f16x3 scoring kernel's design (gfx950):
//  (1) are f16 SUBNORMAL operands kept by v_mfma_f32_32x32x16_f16 / produced by v_cvt_pkrtz_f16_f32 and v_fma_mixlo_f16?
//  (2) what does one VALU instruction of each kind cost next to a stream of f16 MFMAs, at 1 and 2 waves per SIMD?
//  (3) do two waves per SIMD that alternate an MFMA-only phase with a VALU-only phase overlap on their own, with and
//      without a workgroup barrier every few phases (the per-tile barrier of the staged kernel)?
// hipcc --offload-arch=gfx950 -O3 tools/ubench_f16_coexec.hip -o tools/ubench_f16_coexec
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// ---- (1) subnormals -------------------------------------------------------------------------------------------
__global__ void denorm_kernel(float* out) {
  const float tiny = 9.5367431640625e-07f;   // 2^-20: an f16 subnormal (min normal 2^-14)
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)tiny; b[i] = (_Float16)1024.0f; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);   // every element: 16 * 2^-20 * 2^10 = 2^-6
  float x0 = tiny, x1 = 3.0f * tiny;
  asm volatile("" : "+v"(x0), "+v"(x1));
  const unsigned pk = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  unsigned mixlo = 0;
  asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, 0 op_sel_hi:[0,0,0]" : "+v"(mixlo) : "v"(x1));
  // f32 subnormal accumulate: does the MFMA output keep an f32 subnormal?  2^-24 (f16 subnormal) * 2^-14 * 16 = 2^-34: normal f32. skip.
  if (threadIdx.x == 0) {
    out[0] = c[0];
    out[1] = (float)(pk & 0xffff);
    out[2] = (float)(pk >> 16);
    out[3] = (float)(mixlo & 0xffff);
  }
}

// ---- (2) one VALU kind next to MFMAs ---------------------------------------------------------------------------
enum { K_FMA, K_PKFMA, K_EXP, K_RCP, K_CVTPK, K_MIX, K_MIXLO, K_PKMUL, K_PKADD, K_MIN3, K_LDS128, K_PKFMA16, K_PKMUL16, K_PKMAX16, K_EXP16, K_RCP16, K_DOT2, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_cvt_pkrtz", "v_fma_mix_f32",
                                      "v_fma_mixlo_f16", "v_pk_mul_f32", "v_pk_add_f32", "v_min3_f32", "ds_read_b128",
                                      "v_pk_fma_f16", "v_pk_mul_f16", "v_pk_max_f16", "v_exp_f16", "v_rcp_f16", "v_dot2_f32_f16"};

template <int KIND>
__device__ __forceinline__ void valu(float (&v)[16], f32x2 (&w)[8], int j, const float4* lds, float4 (&ld)[4]) {
  float& x = v[j % 16];
  f32x2& y = w[j % 8];
  if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(v[(j + 5) % 16]), "v"(v[(j + 9) % 16]));
  else if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(w[(j + 3) % 8]), "v"(w[(j + 5) % 8]));
  else if constexpr (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  else if constexpr (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
  else if constexpr (KIND == K_CVTPK) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(x) : "v"(v[(j + 5) % 16]), "v"(v[(j + 9) % 16]));
  else if constexpr (KIND == K_MIX) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(x) : "v"(v[(j + 5) % 16]), "v"(v[(j + 9) % 16]));
  else if constexpr (KIND == K_MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(v[(j + 5) % 16]), "v"(v[(j + 9) % 16]));
  else if constexpr (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y) : "v"(w[(j + 3) % 8]));
  else if constexpr (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y) : "v"(w[(j + 3) % 8]));
  else if constexpr (KIND == K_MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(v[(j + 5) % 16]), "v"(v[(j + 9) % 16]));
  else if constexpr (KIND == K_LDS128) ld[j % 4] = lds[(j * 64 + threadIdx.x) & 1023];
  else if constexpr (KIND == K_PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(v[(j + 5) % 16]), "v"(v[(j + 9) % 16]));
  else if constexpr (KIND == K_PKMUL16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x) : "v"(v[(j + 5) % 16]));
  else if constexpr (KIND == K_PKMAX16) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x) : "v"(v[(j + 5) % 16]));
  else if constexpr (KIND == K_EXP16) asm volatile("v_exp_f16 %0, %0" : "+v"(x));
  else if constexpr (KIND == K_RCP16) asm volatile("v_rcp_f16 %0, %0" : "+v"(x));
  else if constexpr (KIND == K_DOT2) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x) : "v"(v[(j + 5) % 16]), "v"(v[(j + 9) % 16]));
}

template <int KIND, int NV>
__global__ void coexec_kernel(float* out, int iters, float seed) {
  __shared__ float4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(seed, seed, seed, seed);
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[16];
  f32x2 w[8];
  float4 ld[4] = {};
  for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * i;
  for (int i = 0; i < 8; ++i) w[i] = f32x2{seed + 0.01f * i, seed - 0.01f * i};
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) valu<KIND>(v, w, m * NV + j, lds, ld);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += w[i].x + w[i].y;
  for (int i = 0; i < 4; ++i) s += ld[i].x + ld[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV>
double run_coexec(int waves_per_simd, float* d) {
  const int threads = 256 * waves_per_simd, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((coexec_kernel<KIND, NV>), dim3(256), dim3(threads), 0, 0, d, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((coexec_kernel<KIND, NV>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / (iters * 4.0 * waves_per_simd);   // ns per MFMA per SIMD
}

template <int KIND>
void sweep(float* d) {
  for (int w = 1; w <= 2; ++w) {
    const double t[6] = {run_coexec<KIND, 0>(w, d), run_coexec<KIND, 2>(w, d), run_coexec<KIND, 4>(w, d), run_coexec<KIND, 6>(w, d),
                         run_coexec<KIND, 8>(w, d), run_coexec<KIND, 12>(w, d)};
    printf("%-16s waves/SIMD=%d  ns per MFMA per SIMD at NV=0,2,4,6,8,12: %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f   (+%.2f ns per op from NV=4 to 12)\n",
           kNames[KIND], w, t[0], t[1], t[2], t[3], t[4], t[5], (t[5] - t[2]) / 8.0);
  }
}

// ---- (3) phase-separated waves ---------------------------------------------------------------------------------
// every iteration: NM back-to-back MFMAs, then NVAL VALU ops (3/4 plain fma, 1/4 exp); BAR > 0: workgroup barrier every BAR iterations
template <int NM, int NVAL, int BAR, int INTERLEAVE>
__global__ void phase_kernel(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * i;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f); }
  for (int it = 0; it < iters; ++it) {
    if constexpr (INTERLEAVE) {
      constexpr int PER = NVAL / NM;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % 4], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int jj = m * PER + j;
          if (jj % 4 == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[jj % 16]));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[jj % 16]) : "v"(v[(jj + 5) % 16]), "v"(v[(jj + 9) % 16]));
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % 4], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NVAL; ++j) {
        if (j % 4 == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j % 16]));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 16]) : "v"(v[(j + 5) % 16]), "v"(v[(j + 9) % 16]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (BAR > 0) { if (it % BAR == BAR - 1) __syncthreads(); }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NVAL, int BAR, int INTERLEAVE>
void run_phase(int waves_per_simd, float* d) {
  const int threads = 256 * waves_per_simd, iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((phase_kernel<NM, NVAL, BAR, INTERLEAVE>), dim3(256), dim3(threads), 0, 0, d, 50, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((phase_kernel<NM, NVAL, BAR, INTERLEAVE>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)iters * waves_per_simd);   // per (wave-iteration) per SIMD
  printf("phase NM=%3d NVAL=%4d barrier-every=%d %s waves/SIMD=%d : %8.1f ns per wave-iteration per SIMD   (MFMA alone %.0f ns at 2.4 GHz)\n",
         NM, NVAL, BAR, INTERLEAVE ? "interleaved" : "phased     ", waves_per_simd, ns, NM * 32 / 2.4);
}

// ---- (4) which feature of the real kernel's stream kills MFMA/VALU overlap? --------------------------------------------
// per iteration: 12 MFMAs rotating over NACC accumulators, each followed by NV VALU instructions (3/4 v_fma, 1/4 v_exp).
//   CHAIN = 1: the VALU instructions form dependent chains of 4 (exp -> fma -> fma -> fma on one register) instead of being independent
//   BVALU = 1: the MFMA's B operand is produced by v_cvt_pkrtz right before it (as the operand split does)
//   ALDS  = 1: the MFMA's A operand is a fresh ds_read_b128 per MFMA (as the weight fragments are)
template <int NACC, int NV, int CHAIN, int BVALU, int ALDS>
__global__ void mix_kernel(float* out, int iters, float seed) {
  __shared__ float4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(seed, seed, seed, seed);
  __syncthreads();
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * i;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      h8 aa = a, bb = b;
      if constexpr (ALDS) aa = __builtin_bit_cast(h8, lds[(m * 64 + threadIdx.x) & 1023]);
      if constexpr (BVALU) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[(2 * j + m) % 16], v[(2 * j + 1 + m) % 16]));
        bb = __builtin_bit_cast(h8, pk);
      }
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, bb, acc[m % NACC], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int jj = m * NV + j;
        if constexpr (CHAIN) {
          float& x = v[(jj / 4) % 16];
          if (jj % 4 == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(v[(jj / 4 + 5) % 16]), "v"(v[(jj / 4 + 9) % 16]));
        } else {
          if (jj % 4 == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[jj % 16]));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[jj % 16]) : "v"(v[(jj + 5) % 16]), "v"(v[(jj + 9) % 16]));
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int NV, int CHAIN, int BVALU, int ALDS>
void run_mix(float* d) {
  for (int w = 1; w <= 2; ++w) {
    const int threads = 256 * w, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix_kernel<NACC, NV, CHAIN, BVALU, ALDS>), dim3(256), dim3(threads), 0, 0, d, 50, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_kernel<NACC, NV, CHAIN, BVALU, ALDS>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mix NACC=%d NV=%d chain=%d b-from-valu=%d a-from-lds=%d waves/SIMD=%d : %6.2f ns per MFMA per SIMD\n", NACC, NV, CHAIN, BVALU, ALDS, w,
           ms * 1e6 / (iters * 12.0 * w));
  }
}

int main() {
  float* d; hipMalloc(&d, 1 << 22);
  hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, d);
  float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("denorm: mfma(16 x 2^-20 x 2^10) = %g (expect 0.015625 if f16 subnormal inputs are kept)\n", h[0]);
  printf("denorm: cvt_pkrtz(2^-20, 3*2^-20) -> 0x%04x 0x%04x (expect 0x0010 0x0030), fma_mixlo_f16(3*2^-20) -> 0x%04x (expect 0x0030)\n",
         (unsigned)h[1], (unsigned)h[2], (unsigned)h[3]);
  sweep<K_FMA>(d); sweep<K_PKFMA>(d); sweep<K_EXP>(d); sweep<K_RCP>(d); sweep<K_CVTPK>(d); sweep<K_MIX>(d); sweep<K_MIXLO>(d);
  sweep<K_PKMUL>(d); sweep<K_PKADD>(d); sweep<K_MIN3>(d); sweep<K_LDS128>(d);
  sweep<K_PKFMA16>(d); sweep<K_PKMUL16>(d); sweep<K_PKMAX16>(d); sweep<K_EXP16>(d); sweep<K_RCP16>(d); sweep<K_DOT2>(d);
  if (getenv("UBENCH_SWEEPS_ONLY")) { hipFree(d); return 0; }
  for (int w = 1; w <= 2; ++w) {
    run_phase<48, 384, 0, 0>(w, d);
    run_phase<48, 384, 4, 0>(w, d);
    run_phase<48, 384, 1, 0>(w, d);
    run_phase<48, 384, 0, 1>(w, d);
    run_phase<48, 384, 4, 1>(w, d);
    run_phase<48, 192, 0, 0>(w, d);
    run_phase<48, 192, 0, 1>(w, d);
  }
  run_mix<4, 7, 0, 0, 0>(d); run_mix<2, 7, 0, 0, 0>(d); run_mix<1, 7, 0, 0, 0>(d); run_mix<2, 0, 0, 0, 0>(d); run_mix<1, 0, 0, 0, 0>(d);
  run_mix<4, 7, 1, 0, 0>(d); run_mix<2, 7, 1, 0, 0>(d);
  run_mix<4, 7, 0, 1, 0>(d); run_mix<2, 7, 0, 1, 0>(d);
  run_mix<4, 7, 0, 0, 1>(d); run_mix<2, 7, 0, 0, 1>(d);
  run_mix<2, 7, 1, 1, 1>(d); run_mix<4, 7, 1, 1, 1>(d);
  hipFree(d);
  return 0;
}
