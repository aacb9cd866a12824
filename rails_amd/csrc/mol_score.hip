// Fused Mixture-of-Logits scoring kernel for gfx950 (MI355X).
//
// Computes, for every (query b, item x) pair, steps 5-8 of the reference's eval-mode
// MoLSimilarity.forward (rails/similarities/mol/similarity_fn.py:389-413, gate :148-201,
// combiner :31-46) from the precomputed query-side (Eq, gq) and item-side (Ex, gi) operands:
//
//   cl[l]  = <Eq[b,p,:], Ex[x,m,:]> / tau                 l = p*P_X + m
//   hid    = silu(W1 cl + b1)                             (H)
//   gqi    = W2 hid + b2                                  (L)
//   g      = gq[b] * gi[x] + gqi ;  w = g * sigmoid(g)
//   pi     = softmax(w) ; pi /= clamp(sum pi, 1e-6)       (the eval-time renormalisation)
//   out    = sum_l pi[l] * cl[l]
//
// None of the (B, N, L) / (B, N, H) intermediates the reference materialises ever leaves the
// register file: one wave owns one unit = (query group of 32/P_Q queries) x (tile of 32 items),
// runs the sub-embedding contraction as 32x32x2 fp32 MFMAs with the items on the column axis,
// and feeds the accumulator registers straight back as the B operand of the two gate GEMMs
// (see mol_layout.h).  Arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32 == an fmaf chain), which is
// what lets the result sit within 1e-4 of the fp32 CPU path; the bound is the fp32 MFMA rate.
#include <hip/hip_runtime.h>

#include "mol_kernels.h"
#include "mol_layout.h"

namespace mol {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// x / tau with a compile-time-unknown but launch-constant tau: q = x*r, one Newton correction on the
// residual.  Correctly rounded except for rare last-bit cases (r = RN(1/tau)).
__device__ __forceinline__ float div_const(float x, float tau, float rcp_tau) {
  const float q = x * rcp_tau;
  const float e = __builtin_fmaf(-q, tau, x);
  return __builtin_fmaf(e, rcp_tau, q);
}

// x * sigmoid(x) = x / (1 + exp(-x)); v_exp_f32 / v_rcp_f32 are 1 ulp
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

template <int N>
__device__ __forceinline__ f32x16 rotate_down(f32x16 v) {
  f32x16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = v[(i + N) & 15];
  return r;
}

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

template <int PQ, int PX, int DD, int H>
__global__ __launch_bounds__(kScoreThreads, 2) void mol_score_kernel(ScoreArgs p) {
  using G = Geo<PQ, PX, DD, H>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const float4* sW1 = reinterpret_cast<const float4*>(smem);
  const float4* sW2 = sW1 + G::kW1Floats / 4;
  const float* sB1 = smem + G::kW1Floats + G::kW2Floats;
  const float4* sB2 = reinterpret_cast<const float4*>(sB1 + H);

  {  // stage the packed gate weights once per workgroup
    const float4* src = reinterpret_cast<const float4*>(p.wpack);
    float4* dst = reinterpret_cast<float4*>(smem);
    for (int i = threadIdx.x; i < G::kWpackFloats / 4; i += kScoreThreads) dst[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int hi = lane >> 5;
  const int x = lane & 31;
  // shared corpus: unit = (tile, query group), groups fastest so the waves of one workgroup share a
  // tile through L1/L2.  per-row candidates: unit = (row b, tile of b's candidates), one query per unit.
  const int inner = p.per_row ? (int)p.n_tiles : p.n_groups;
  const int64_t n_units = p.per_row ? (int64_t)p.B * p.n_tiles : p.n_tiles * p.n_groups;
  const int64_t stride = (int64_t)gridDim.x * kScoreWaves;

  for (int64_t u = (int64_t)blockIdx.x * kScoreWaves + wave; u < n_units; u += stride) {
    const int64_t outer = u / inner;
    const int innr = (int)(u - outer * inner);
    const int64_t tile = p.per_row ? innr : outer;          // tile index inside the row / corpus
    const int row = p.per_row ? (int)outer : -1;            // per-row mode: the only query of this unit
    const int g = p.per_row ? row / G::QT : innr;
    const int64_t tile_addr = p.per_row ? (int64_t)row * p.n_tiles + tile : tile;
    const float4* tEx = reinterpret_cast<const float4*>(p.ipack + tile_addr * (int64_t)G::kTileFloats);
    const float4* tGi = tEx + G::kTileExFloats / 4;
    const float4* eq = reinterpret_cast<const float4*>(p.eqfrag + (int64_t)g * G::kEqGroupFloats);

    // ---- GEMM1: D1[m][(qj,p), x] = sum_d Eq[qj,p,d] * Ex[x,m,d] -------------------------------
    f32x16 D1[PX];
#pragma unroll
    for (int m = 0; m < PX; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) D1[m][r] = 0.0f;

#pragma unroll
    for (int sc = 0; sc < DD / 8; ++sc) {
      const float4 a = eq[sc * 64 + lane];
#pragma unroll
      for (int m = 0; m < PX; ++m) {
        const float4 b = tEx[(m * (DD / 8) + sc) * 64 + lane];
        D1[m] = mfma32(a.x, b.x, D1[m]);
        D1[m] = mfma32(a.y, b.y, D1[m]);
        D1[m] = mfma32(a.z, b.z, D1[m]);
        D1[m] = mfma32(a.w, b.w, D1[m]);
      }
    }

    // ---- per query of the group: gate MLP + mixture ----------------------------------------
#pragma unroll 1
    for (int qj = 0; qj < G::QT; ++qj) {
      const int q = g * G::QT + qj;
      if (q < p.B && (row < 0 || q == row)) {
        // cl = <.,.> / tau, in place in registers [0, RPQ) of every GEMM1 tile
#pragma unroll
        for (int m = 0; m < PX; ++m)
#pragma unroll
          for (int r = 0; r < G::RPQ; ++r) D1[m][r] = div_const(D1[m][r], p.temperature, p.rcp_temperature);

        // GEMM2: hid[h, x] = b1[h] + sum_l W1[h, l] cl[l, x]
        f32x16 D2[G::TH];
#pragma unroll
        for (int t = 0; t < G::TH; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) D2[t][r] = sB1[t * 32 + hi * 16 + r];
#pragma unroll
        for (int ec = 0; ec < G::E / 4; ++ec) {
#pragma unroll
          for (int t = 0; t < G::TH; ++t) {
            const float4 a = sW1[(ec * G::TH + t) * 64 + lane];
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int e = ec * 4 + j;
              D2[t] = mfma32(av[j], D1[e / G::RPQ][e % G::RPQ], D2[t]);
            }
          }
        }
#pragma unroll
        for (int t = 0; t < G::TH; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) D2[t][r] = silu_f(D2[t][r]);

        // GEMM3: gqi[l, x] = sum_h W2[l, h] hid[h, x]   (b2 joins in the epilogue)
        f32x16 D3[G::TL];
#pragma unroll
        for (int v = 0; v < G::TL; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) D3[v][r] = 0.0f;
#pragma unroll
        for (int fc = 0; fc < G::F / 4; ++fc) {
#pragma unroll
          for (int v = 0; v < G::TL; ++v) {
            const float4 a = sW2[(fc * G::TL + v) * 64 + lane];
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int f = fc * 4 + j;
              D3[v] = mfma32(av[j], D2[f / 16][f % 16], D3[v]);
            }
          }
        }

        // epilogue: combine, softmax over all L (this lane's E values + the partner half's), mix
        const float4* gq4 = reinterpret_cast<const float4*>(p.gqfrag + (int64_t)q * G::L + hi * G::E);
        float mx = -INFINITY;
#pragma unroll
        for (int ec = 0; ec < G::E / 4; ++ec) {
          const float4 gi = tGi[ec * 64 + lane];
          const float4 gq = gq4[ec];
          const float4 b2 = sB2[hi * (G::E / 4) + ec];
          const float giv[4] = {gi.x, gi.y, gi.z, gi.w};
          const float gqv[4] = {gq.x, gq.y, gq.z, gq.w};
          const float b2v[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = ec * 4 + j;
            const float gqi = D3[e / 16][e % 16] + b2v[j];
            const float gg = gqv[j] * giv[j] + gqi;
            const float w = silu_f(gg);
            D3[e / 16][e % 16] = w;
            mx = fmaxf(mx, w);
          }
        }
        mx = fmaxf(mx, xor32(mx));
        float den = 0.0f;
#pragma unroll
        for (int e = 0; e < G::E; ++e) {
          const float ex = __expf(D3[e / 16][e % 16] - mx);
          D3[e / 16][e % 16] = ex;
          den += ex;
        }
        den += xor32(den);
        const float rden = __builtin_amdgcn_rcpf(den);
        float s2 = 0.0f, num = 0.0f;
#pragma unroll
        for (int e = 0; e < G::E; ++e) {
          const float pi = D3[e / 16][e % 16] * rden;
          s2 += pi;
          num = __builtin_fmaf(pi, D1[e / G::RPQ][e % G::RPQ], num);
        }
        s2 += xor32(s2);
        num += xor32(num);
        const float out = num / fmaxf(s2, 1e-6f);
        const int64_t item = tile * kTileItems + x;
        if (hi == 0 && item < p.n_items) p.logits[(int64_t)q * p.ld + item] = out;
      }
      // bring the next query's rows down to registers [0, RPQ)
#pragma unroll
      for (int m = 0; m < PX; ++m) D1[m] = rotate_down<G::RPQ>(D1[m]);
    }
  }
}

template <int PQ, int PX, int DD, int H>
static int launch_score(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using G = Geo<PQ, PX, DD, H>;
  const size_t lds = (size_t)G::kWpackFloats * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mol_score_kernel<PQ, PX, DD, H>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return kErrLaunch;
    attr_set = true;
  }
  const int64_t n_units = a.per_row ? (int64_t)a.B * a.n_tiles : a.n_tiles * a.n_groups;
  int64_t grid = (n_units + kScoreWaves - 1) / kScoreWaves;
  if (grid > n_cu) grid = n_cu;
  if (grid < 1) return kOk;
  hipLaunchKernelGGL((mol_score_kernel<PQ, PX, DD, H>), dim3((unsigned)grid), dim3(kScoreThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

bool score_supported(const Shape& s) {
  if (s.gating_qi_hidden_dim != 128) return false;
  const int pq = s.query_dot_product_groups, px = s.item_dot_product_groups, dd = s.dot_product_dimension;
  return (pq == 8 && px == 4 && dd == 64) || (pq == 8 && px == 4 && dd == 128) || (pq == 8 && px == 8 && dd == 32);
}

int score_launch(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream) {
  if (!score_supported(s)) return kErrUnsupported;
#define MOL_CASE(pq, px, dd)                                                                             \
  if (s.query_dot_product_groups == pq && s.item_dot_product_groups == px && s.dot_product_dimension == dd) \
    return launch_score<pq, px, dd, 128>(a, n_cu, stream);
  MOL_CASE(8, 4, 64)
  MOL_CASE(8, 4, 128)
  MOL_CASE(8, 8, 32)
#undef MOL_CASE
  return kErrUnsupported;
}

}  // namespace mol
