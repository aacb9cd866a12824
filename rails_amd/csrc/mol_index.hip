// Index-side kernels: pair-gate weight packing, item index build / unpack / gather.
//
// The item index is what the reference's MoLTopKModule keeps per corpus
// (rails/indexing/mol_top_k.py:29-81) plus the item-only work MoLSimilarity.forward redoes on every
// call: component embeddings (rails/similarities/mol/item_embeddings_fns.py:149-183) and the item gate
// (similarity_fn.py:170-171).  It is computed once, in fp32, and laid out in tiles of 32 items in the
// exact order the scoring kernel's MFMA operands consume it (mol_layout.h), so that every operand
// fetch of the hot loop is one contiguous 1 KiB per wave.
#include <hip/hip_runtime.h>

#include "mol_kernels.h"
#include "mol_layout.h"

namespace mol {

// ---- pair-gate weights -> fragment order --------------------------------------------------------
__global__ void pack_gate_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                 float* __restrict__ out, int PQ, int PX, int H) {
  const int L = PQ * PX, TH = H / 32, TL = L / 32, E = L / 2;
  if (H <= 0) {   // no hidden layer: Wfrag[ec][v][lane][j] = -log2e * W[lrow(v, lane&31)][logit_of(4ec+j, lane>>5)], then b2frag[hi][e]
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L * L + L; i += gridDim.x * blockDim.x) {
      float v;
      if (i < L * L) {
        const int j = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
        const int tv = blk % TL, ec = blk / TL;
        const int row = lane & 31;
        const int l = logit_of(16 * tv + reg_of_row(row), half_of_row(row), PQ, PX);
        v = -kLog2e * w1[l * L + logit_of(4 * ec + j, lane >> 5, PQ, PX)];
      } else {
        const int k = i - L * L;
        v = -kLog2e * b1[logit_of(k % E, k / E, PQ, PX)];
      }
      out[i] = v;
    }
    return;
  }
  const int total = 2 * H * L + H + L;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float v;
    if (i < H * L) {  // W1frag[ec][t][lane][j] = W1[32t + (lane&31)][logit_of(4ec+j, lane>>5)]
      const int j = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
      const int t = blk % TH, ec = blk / TH;
      v = -kLog2e * w1[(32 * t + (lane & 31)) * L + logit_of(4 * ec + j, lane >> 5, PQ, PX)];
    } else if (i < 2 * H * L) {  // W2frag[fc][v][lane][j] = W2[lrow(v, lane&31)][hidden_of(4fc+j, lane>>5)]
      const int k = i - H * L;
      const int j = k & 3, lane = (k >> 2) & 63, blk = k >> 8;
      const int tv = blk % TL, fc = blk / TL;
      const int row = lane & 31;
      const int l = logit_of(16 * tv + reg_of_row(row), half_of_row(row), PQ, PX);
      v = w2[l * H + hidden_of(4 * fc + j, lane >> 5)];
    } else if (i < 2 * H * L + H) {  // b1frag[t][hi][r]
      const int k = i - 2 * H * L;
      const int r = k & 15, hi = (k >> 4) & 1, t = k >> 5;
      v = -kLog2e * b1[32 * t + acc_row(r, hi)];
    } else {  // b2frag[hi][e]
      const int k = i - 2 * H * L - H;
      const int hi = k / E, e = k % E;
      v = -kLog2e * b2[logit_of(e, hi, PQ, PX)];
    }
    out[i] = v;
  }
}

// The same weights in the fragment order of the small-unit kernel (mol_score_small.h, layout in mol_layout.h; P_Q = 8):
//   W1s[mc][t][lane][c]   = -log2e W1[hidden16(4t + i, go)][logit16(4mc + c, g)]     lane = 16g + 4go + i: A operand (row 4go + i, k = g)
//   W2s[t][v][lane][c]    =        W2[logit16(4v + i, go)][hidden16(4t + c, g)]      of K-steps 4mc + c / 4t + c
//   b1s[t][g][i]          = -log2e b1[hidden16(4t + i, g)]                           accumulator start of D2 tile t, register i
//   b2s[v][g][i]          = -log2e b2[logit16(4v + i, g)]
__global__ void pack_gate16_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                   const float* __restrict__ b2, float* __restrict__ out, int PX, int H) {
  const int L = 8 * PX, TH = H / 16, TL = L / 16;
  const int total = 2 * H * L + H + L;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    float v;
    if (idx < 2 * H * L) {
      const bool second = idx >= H * L;
      const int k = second ? idx - H * L : idx;
      const int c = k & 3, lane = (k >> 2) & 63, blk = k >> 8;
      const int g = lane >> 4, go = (lane >> 2) & 3, i = lane & 3;
      if (!second) {
        const int t = blk % TH, mc = blk / TH;
        v = -kLog2e * w1[hidden16(4 * t + i, go) * L + logit16(4 * mc + c, g, PX)];
      } else {
        const int tv = blk % TL, t = blk / TL;
        v = w2[logit16(4 * tv + i, go, PX) * H + hidden16(4 * t + c, g)];
      }
    } else if (idx < 2 * H * L + H) {
      const int k = idx - 2 * H * L;
      v = -kLog2e * b1[hidden16(4 * (k >> 4) + (k & 3), (k >> 2) & 3)];
    } else {
      const int k = idx - 2 * H * L - H;
      v = -kLog2e * b2[logit16(4 * (k >> 4) + (k & 3), (k >> 2) & 3, PX)];
    }
    out[idx] = v;
  }
}

int pack_gate_weights(const Shape& s, const Weights& w, float* wpack, hipStream_t stream) {
  const int H = s.gating_qi_hidden_dim > 0 ? s.gating_qi_hidden_dim : 0, L = num_logits(s);
  const int total = H > 0 ? 2 * H * L + H + L : L * L + L;
  hipLaunchKernelGGL(pack_gate_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w.gqi_w1, w.gqi_b1,
                     w.gqi_w2, w.gqi_b2, wpack, s.query_dot_product_groups, s.item_dot_product_groups, H);
  if (hipGetLastError() != hipSuccess) return kErrLaunch;
  if (score_small_shape(s))
    hipLaunchKernelGGL(pack_gate16_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w.gqi_w1, w.gqi_b1, w.gqi_w2, w.gqi_b2,
                       wpack + total, s.item_dot_product_groups, H);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- pair-gate weights -> f16 hi/lo fragments for the f16x3 precision mode (mol_layout.h) -------------------
__device__ __forceinline__ void split_f16(float x, unsigned short& hi, unsigned short& lo) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x, 0.0f));   // round toward zero
  const _Float16 l = (_Float16)(x - (float)h.x);                              // remainder is exact in fp32; RNE to f16
  hi = __builtin_bit_cast(unsigned short, h.x);
  lo = __builtin_bit_cast(unsigned short, l);
}

__global__ void pack_gate_split_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                       float* __restrict__ out, int PQ, int PX, int H) {
  const int L = PQ * PX, TH = H / 32, TL = L / 32, E = L / 2;
  unsigned short* w1hi = reinterpret_cast<unsigned short*>(out);
  unsigned short* w1lo = w1hi + H * L;
  unsigned short* w2hi = w1lo + H * L;
  unsigned short* w2lo = w2hi + H * L;
  float* b1f = out + 2 * H * L;
  float* b2f = b1f + H;
  const int total = 2 * H * L + H + L;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (i < H * L) {  // W1 fragment [s][t][lane][jj]: row 32t + (lane&31), logit logit_of(8s + jj, lane>>5)
      const int jj = i & 7, lane = (i >> 3) & 63, blk = i >> 9;
      const int t = blk % TH, ks = blk / TH;
      const float v = -kLog2e * w1[(32 * t + (lane & 31)) * L + logit_of(8 * ks + jj, lane >> 5, PQ, PX)];
      split_f16(v, w1hi[i], w1lo[i]);
    } else if (i < 2 * H * L) {  // W2 fragment [s][v][lane][jj]: row lrow(v, lane&31), hidden hidden_of(8s + jj, lane>>5)
      const int k = i - H * L;
      const int jj = k & 7, lane = (k >> 3) & 63, blk = k >> 9;
      const int tv = blk % TL, ks = blk / TL;
      const int row = lane & 31;
      const int l = logit_of(16 * tv + reg_of_row(row), half_of_row(row), PQ, PX);
      const float v = w2[l * H + hidden_of(8 * ks + jj, lane >> 5)];
      split_f16(v, w2hi[k], w2lo[k]);
    } else if (i < 2 * H * L + H) {  // b1frag[t][hi][r], carries -log2e
      const int k = i - 2 * H * L;
      const int r = k & 15, hi = (k >> 4) & 1, t = k >> 5;
      b1f[k] = -kLog2e * b1[32 * t + acc_row(r, hi)];
    } else {  // b2frag[hi][e], carries -log2e
      const int k = i - 2 * H * L - H;
      const int hi = k / E, e = k % E;
      b2f[k] = -kLog2e * b2[logit_of(e, hi, PQ, PX)];
    }
  }
}

int pack_gate_weights_split(const Shape& s, const Weights& w, float* wpack, hipStream_t stream) {
  const int H = s.gating_qi_hidden_dim, L = num_logits(s);
  const int total = 2 * H * L + H + L;
  hipLaunchKernelGGL(pack_gate_split_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w.gqi_w1, w.gqi_b1,
                     w.gqi_w2, w.gqi_b2, wpack, s.query_dot_product_groups, s.item_dot_product_groups, H);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- index build -------------------------------------------------------------------------------
// One workgroup per tile of 32 items.  Every dense layer is done "column per thread": thread c keeps
// 32 accumulators (one per item) and walks k in order, reading X / hidden values as LDS broadcasts.
// fp32 throughout, precise expf and true divisions: this runs once per corpus and has to sit as close
// to the reference's CPU arithmetic as a different summation order allows.
constexpr int kBuildThreads = 256;

struct BuildArgs {
  const float* items;   // (n, D)
  const float* gluw;    // item_hidden_dim > 0: (D, 2*IH) GLU weight (x @ W), else NULL
  const float* glub;    // (2*IH)
  int IH, glu_kind;
  int has_gate;         // 0: no item-only gate part -> gi = 0
  const float* pw;      // (PX*d, D) or (PX*d, IH)
  const float* pb;      // (PX*d)
  const float* g1w;     // (Hi, D)
  const float* g1b;     // (Hi)
  const float* g2w;     // (L, Hi)
  float* ipack;
  int64_t n;
  int D, PQ, PX, d, Hi;
  int l2norm;
  float eps;
};

__device__ __forceinline__ void dense_cols(const float* __restrict__ W, const float* __restrict__ bias, int col0,
                                           int ncols, int K, const float* __restrict__ in_s, int in_ld,
                                           float* __restrict__ out_s, int out_ld, bool silu) {
  // out_s[x][c - col0] = act(bias[c] + sum_k W[c][k] * in_s[x][k])  for c in [col0, col0 + ncols)
  for (int c = threadIdx.x; c < ncols; c += kBuildThreads) {
    const float* wrow = W + (int64_t)(col0 + c) * K;
    float acc[kTileItems];
#pragma unroll
    for (int x = 0; x < kTileItems; ++x) acc[x] = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float wv = wrow[k];
#pragma unroll
      for (int x = 0; x < kTileItems; ++x) acc[x] = __builtin_fmaf(wv, in_s[x * in_ld + k], acc[x]);
    }
    const float bv = bias ? bias[col0 + c] : 0.0f;
#pragma unroll
    for (int x = 0; x < kTileItems; ++x) {
      float v = acc[x] + bv;
      if (silu) v = v / (1.0f + expf(-v));
      out_s[x * out_ld + c] = v;
    }
  }
}

__global__ __launch_bounds__(kBuildThreads) void index_build_kernel(BuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, d = a.d, PX = a.PX, L = a.PQ * a.PX, Hi = a.Hi;
  const int mpc = (kBuildThreads / d) > 0 ? (kBuildThreads / d) : 1;  // component groups per column chunk
  const int chunk_cols = mpc * d;
  const int xs_ld = D + 1, hs_ld = Hi + 1;
  const int cs_ld = (chunk_cols > L ? chunk_cols : L) + 1;
  float* xs = smem;                          // [32][D+1]
  float* hs = xs + kTileItems * xs_ld;       // [32][Hi+1]
  float* cs = hs + kTileItems * hs_ld;       // [32][max(chunk_cols, L)+1]
  float* ns = cs + kTileItems * cs_ld;       // [32][mpc] inverse norms
  const int gs_ld = a.IH + 1;
  float* gs = ns + kTileItems * mpc;         // [32][IH+1] GLU hidden layer of the item projection (item_hidden_dim > 0)
  const int64_t tile = blockIdx.x;
  const int64_t item0 = tile * kTileItems;
  float* tEx = a.ipack + tile * (int64_t)(kTileItems * (PX * d + L));
  float* tGi = tEx + kTileItems * PX * d;

  for (int i = threadIdx.x; i < kTileItems * D; i += kBuildThreads) {
    const int x = i / D, k = i - x * D;
    xs[x * xs_ld + k] = (item0 + x < a.n) ? a.items[(item0 + x) * D + k] : 0.0f;
  }
  __syncthreads();

  // item projection with a GLU hidden layer (similarity_utils.py:127-143, layers.py:19-74): h = act(x W_l + b_l) * (x W_r + b_r),
  // W = (D, 2 IH) applied as x @ W.  Thread c owns hidden column c: both halves, 32 items each.
  if (a.IH > 0) {
    for (int c = threadIdx.x; c < a.IH; c += kBuildThreads) {
      float al[kTileItems], ar[kTileItems];
#pragma unroll
      for (int x = 0; x < kTileItems; ++x) { al[x] = 0.0f; ar[x] = 0.0f; }
      for (int k = 0; k < D; ++k) {
        const float wl = a.gluw[(int64_t)k * 2 * a.IH + c], wr = a.gluw[(int64_t)k * 2 * a.IH + a.IH + c];
#pragma unroll
        for (int x = 0; x < kTileItems; ++x) {
          const float xv = xs[x * xs_ld + k];
          al[x] = __builtin_fmaf(xv, wl, al[x]);
          ar[x] = __builtin_fmaf(xv, wr, ar[x]);
        }
      }
      const float bl = a.glub[c], br = a.glub[a.IH + c];
#pragma unroll
      for (int x = 0; x < kTileItems; ++x) {
        const float l = al[x] + bl, r = ar[x] + br;
        const float act = a.glu_kind == RAILS_GEGLU ? 0.5f * l * (1.0f + erff(l * 0.70710678118654752440f)) : l / (1.0f + expf(-l));
        gs[x * gs_ld + c] = act * r;
      }
    }
    __syncthreads();
  }
  const float* pin = a.IH > 0 ? gs : xs;
  const int pin_ld = a.IH > 0 ? gs_ld : xs_ld, pK = a.IH > 0 ? a.IH : D;

  // component embeddings, one chunk of whole component groups at a time
  for (int m0 = 0; m0 < PX; m0 += mpc) {
    const int groups = (PX - m0 < mpc) ? (PX - m0) : mpc;
    dense_cols(a.pw, a.pb, m0 * d, groups * d, pK, pin, pin_ld, cs, cs_ld, false);
    __syncthreads();
    for (int i = threadIdx.x; i < kTileItems * groups; i += kBuildThreads) {
      const int x = i / groups, mg = i - x * groups;
      float ss = 0.0f;
      for (int k = 0; k < d; ++k) {
        const float v = cs[x * cs_ld + mg * d + k];
        ss = __builtin_fmaf(v, v, ss);
      }
      // x / clamp(norm, min=eps)  (item_embeddings_fns.py:173-182)
      ns[x * mpc + mg] = a.l2norm ? fmaxf(sqrtf(ss), a.eps) : 1.0f;
    }
    __syncthreads();
    // fragment order: Ex slot (m, c8, lane)[j] = Ex[x = lane&31][m][kdim_of(4*c8 + j, lane>>5)]
    const int slots = groups * (d / 8) * 64;
    for (int i = threadIdx.x; i < slots; i += kBuildThreads) {
      const int lane = i & 63, c8 = (i >> 6) % (d / 8), mg = (i >> 6) / (d / 8);
      const int x = lane & 31, hi = lane >> 5;
      const float nv = ns[x * mpc + mg];
      float4 o;
      float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
      for (int j = 0; j < 4; ++j) ov[j] = cs[x * cs_ld + mg * d + kdim_of(4 * c8 + j, hi, d)] / nv;
      if (item0 + x >= a.n) o = make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(tEx)[((m0 + mg) * (d / 8) + c8) * 64 + lane] = o;
    }
    __syncthreads();
  }

  // item gate: gi = W2 silu(W1 x + b1)   (modeling/similarity_utils.py:169-185); absent part (gating_item_fn = False): zeros
  if (a.has_gate) {
    dense_cols(a.g1w, a.g1b, 0, Hi, D, xs, xs_ld, hs, hs_ld, true);
    __syncthreads();
    dense_cols(a.g2w, nullptr, 0, L, Hi, hs, hs_ld, cs, cs_ld, false);
  } else {
    for (int i = threadIdx.x; i < kTileItems * L; i += kBuildThreads) cs[(i / L) * cs_ld + i % L] = 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (L / 8) * 64; i += kBuildThreads) {
    const int lane = i & 63, ec = i >> 6;
    const int x = lane & 31, hi = lane >> 5;
    float4 o;
    float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int j = 0; j < 4; ++j) ov[j] = cs[x * cs_ld + logit_of(4 * ec + j, hi, a.PQ, PX)];
    if (item0 + x >= a.n) o = make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(tGi)[ec * 64 + lane] = o;
  }
}

static size_t build_lds_bytes(const Shape& s) {
  const int d = s.dot_product_dimension, L = num_logits(s);
  const int mpc = (kBuildThreads / d) > 0 ? (kBuildThreads / d) : 1;
  const int chunk_cols = mpc * d;
  const int cs_ld = (chunk_cols > L ? chunk_cols : L) + 1;
  const int hi_w = s.gating_has_item ? s.gating_item_hidden_dim : 0, ih = s.item_hidden_dim > 0 ? s.item_hidden_dim : 0;
  return sizeof(float) * (size_t)kTileItems * ((s.item_embedding_dim + 1) + (hi_w + 1) + cs_ld + mpc + (ih + 1));
}

int index_build(const Shape& s, const Weights& w, const float* items, int64_t n, float* ipack, hipStream_t stream) {
  const int64_t tiles = num_tiles(n);
  if (tiles == 0) return kOk;
  BuildArgs a;
  a.items = items; a.pw = w.i_proj_w; a.pb = w.i_proj_b; a.g1w = w.gi_w1; a.g1b = w.gi_b1; a.g2w = w.gi_w2;
  a.ipack = ipack; a.n = n; a.D = s.item_embedding_dim; a.PQ = s.query_dot_product_groups;
  a.PX = s.item_dot_product_groups; a.d = s.dot_product_dimension; a.Hi = s.gating_has_item ? s.gating_item_hidden_dim : 0;
  a.gluw = w.i_glu_w; a.glub = w.i_glu_b; a.IH = s.item_hidden_dim > 0 ? s.item_hidden_dim : 0; a.glu_kind = s.item_nonlinearity;
  a.has_gate = s.gating_has_item;
  a.l2norm = s.dot_product_l2_norm; a.eps = s.eps;
  const size_t lds = build_lds_bytes(s);
  if (lds > 160 * 1024) { set_error("index build needs %zu B of LDS (> 160 KiB) for this shape", lds); return kErrUnsupported; }
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&index_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return kErrLaunch;
  hipLaunchKernelGGL(index_build_kernel, dim3((unsigned)tiles), dim3(kBuildThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- precision f16x3: the Ex fragments of a tile, fp32 -> f16 hi/lo, in place ---------------------------------------
// A lane's two consecutive 16-byte chunks (2ks, 2ks+1) of an item group are its 8 k-values of K=16 step ks (k = hi*d/2 +
// 8ks + jj); they are replaced by the 8 hi halves (chunk 2ks) and the 8 lo halves (chunk 2ks+1).  Each lane rewrites only
// what it read, so the conversion is race-free in place; gi stays fp32.  Bytes per item unchanged.
__global__ void index_split_kernel(float4* __restrict__ ipack, int64_t n_tiles, int pairs_per_tile, int tile_f4) {
  const int64_t total = n_tiles * pairs_per_tile * 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const int64_t pr = i >> 6;
    const int64_t tile = pr / pairs_per_tile;
    const int pair = (int)(pr - tile * pairs_per_tile);
    float4* a = ipack + tile * tile_f4 + (2 * pair) * 64 + lane;
    const float4 u = a[0], v = a[64];
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    unsigned short hb[8], lb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_f16(x[j], hb[j], lb[j]);
    uint4 H, Lo;
    H.x = hb[0] | (unsigned)hb[1] << 16; H.y = hb[2] | (unsigned)hb[3] << 16; H.z = hb[4] | (unsigned)hb[5] << 16; H.w = hb[6] | (unsigned)hb[7] << 16;
    Lo.x = lb[0] | (unsigned)lb[1] << 16; Lo.y = lb[2] | (unsigned)lb[3] << 16; Lo.z = lb[4] | (unsigned)lb[5] << 16; Lo.w = lb[6] | (unsigned)lb[7] << 16;
    *reinterpret_cast<uint4*>(a) = H;
    *reinterpret_cast<uint4*>(a + 64) = Lo;
  }
}

int index_split_inplace(const Shape& s, float* ipack, int64_t n, hipStream_t stream) {
  const int64_t tiles = num_tiles(n);
  if (tiles == 0) return kOk;
  if (s.dot_product_dimension % 16 != 0) { set_error("precision f16x3 needs dot_product_dimension % 16 == 0"); return kErrUnsupported; }
  const int pairs = s.item_dot_product_groups * s.dot_product_dimension / 16;
  int64_t blocks = (tiles * pairs * 64 + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(index_split_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<float4*>(ipack), tiles, pairs,
                     (int)(tile_floats(s) / 4));
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- unpack: fragment order -> plain (n, PX, d) / (n, L) -----------------------------------------
__global__ void index_unpack_kernel(const float* __restrict__ ipack, int64_t n, int PQ, int PX, int d,
                                    float* __restrict__ ex, float* __restrict__ gi, int split) {
  const int L = PQ * PX;
  const int64_t tile = blockIdx.x;
  const float* tEx = ipack + tile * (int64_t)(kTileItems * (PX * d + L));
  const float* tGi = tEx + kTileItems * PX * d;
  if (ex) {
    for (int i = threadIdx.x; i < kTileItems * PX * d; i += blockDim.x) {
      const int j = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
      const int c8 = blk % (d / 8), m = blk / (d / 8);
      const int64_t item = tile * kTileItems + (lane & 31);
      float v = tEx[i];
      if (split) {   // hi + lo of the f16x3 fragments (exact in fp32): chunk pair (2ks, 2ks+1) holds 8 hi, then 8 lo halves
        const unsigned short* th = reinterpret_cast<const unsigned short*>(tEx);
        const int jj = 4 * (c8 & 1) + j, base = ((m * (d / 8) + (c8 & ~1)) * 64 + lane) * 8;
        const _Float16 h = __builtin_bit_cast(_Float16, th[base + jj]), l = __builtin_bit_cast(_Float16, th[base + 512 + jj]);
        v = (float)h + (float)l;
      }
      if (item < n) ex[(item * PX + m) * d + kdim_of(4 * c8 + j, lane >> 5, d)] = v;
    }
  }
  if (gi) {
    for (int i = threadIdx.x; i < kTileItems * L; i += blockDim.x) {
      const int j = i & 3, lane = (i >> 2) & 63, ec = i >> 8;
      const int64_t item = tile * kTileItems + (lane & 31);
      if (item < n) gi[item * L + logit_of(4 * ec + j, lane >> 5, PQ, PX)] = tGi[i];
    }
  }
}

int index_unpack(const Shape& s, const float* ipack, int64_t n, float* ex, float* gi, hipStream_t stream) {
  const int64_t tiles = num_tiles(n);
  if (tiles == 0) return kOk;
  hipLaunchKernelGGL(index_unpack_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, ipack, n,
                     s.query_dot_product_groups, s.item_dot_product_groups, s.dot_product_dimension, ex, gi, is_split(s) ? 1 : 0);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- row-major copy of an fp32 index: item i's fragment slot s, lane half h (one float4) at rows[i * RP + 2 s + h], RP = floats per item / 4.
// A candidate's operands are then RP consecutive float4 (amzn-books: 1 280 B in ten 128-byte lines) instead of RP pieces of 16 B each in a line
// of its own across the tile (8 x read amplification when candidates are re-scored in place: rails_mol_score_indexed_rows).
__global__ void index_rows_kernel(const float4* __restrict__ ipack, int64_t n, int rp, int tile_f4, float4* __restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * rp) return;
  const int64_t item = i / rp;
  const int j = (int)(i - item * rp), slot = j >> 1, h = j & 1;
  rows[i] = ipack[(item >> 5) * tile_f4 + slot * 64 + h * 32 + (item & 31)];
}

int index_rows_build(const Shape& s, const float* ipack, int64_t n, float* rows, hipStream_t stream) {
  if (n <= 0) return kOk;
  const int tile_f4 = (int)(tile_floats(s) / 4), rp = tile_f4 / 32;
  const int64_t total = n * rp;
  hipLaunchKernelGGL(index_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const float4*>(ipack), n, rp, tile_f4,
                     reinterpret_cast<float4*>(rows));
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- gather: rows x n_cand candidate positions -> a tile-packed index of their own -----------------
__global__ void index_gather_kernel(const float4* __restrict__ ipack, int64_t n, const int64_t* __restrict__ idx,
                                    int64_t tiles_per_row, int64_t n_cand, int tile_f4, float4* __restrict__ out) {
  const int64_t otile = blockIdx.x;                 // = row * tiles_per_row + t
  const int64_t row = otile / tiles_per_row, t = otile - row * tiles_per_row;
  __shared__ int64_t src[kTileItems];
  if (threadIdx.x < kTileItems) {
    int64_t v = idx[row * n_cand + t * kTileItems + threadIdx.x];
    src[threadIdx.x] = (v >= 0 && v < n) ? v : -1;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tile_f4; i += blockDim.x) {
    const int lane = i & 63, slot = i >> 6;
    const int64_t sidx = src[lane & 31];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sidx >= 0) v = ipack[(sidx >> 5) * tile_f4 + slot * 64 + (lane & 32) + (sidx & 31)];
    out[otile * tile_f4 + i] = v;
  }
}

int index_gather(const Shape& s, const float* ipack, int64_t n, const int64_t* idx, int64_t rows, int64_t n_cand,
                 float* out, hipStream_t stream) {
  if (n_cand % kTileItems != 0) { set_error("n_cand (%lld) must be a multiple of 32", (long long)n_cand); return kErrInvalid; }
  const int64_t tiles_per_row = n_cand / kTileItems;
  if (rows * tiles_per_row == 0) return kOk;
  hipLaunchKernelGGL(index_gather_kernel, dim3((unsigned)(rows * tiles_per_row)), dim3(256), 0, stream,
                     reinterpret_cast<const float4*>(ipack), n, idx, tiles_per_row, n_cand, (int)(tile_floats(s) / 4),
                     reinterpret_cast<float4*>(out));
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- synthetic item table on the device (measurement / test infrastructure) ------------------------------------------------------
// The counter-hash generator of the synthetic corpora ("generated by a counter-based hash of the item id so any
// shard / sub-range is reproducible on CPU without materialising the table"), bit-equal to oracle/mol_oracle.py hash_item_table:
//   h = splitmix64((item * dim + col) ^ (seed * 0xD1B54A32D192ED03));  value = float(sum of h's four 16-bit lanes - 2 * 65535) * scale
// Integer arithmetic + one exact int -> float conversion + one fp32 multiply: the same bits on every platform.  A 125 M-item shard
// (32 GB) is drawn in place in HBM instead of being hashed on the host and copied.
__global__ void hash_item_table_kernel(unsigned long long seed_mix, int64_t first_item, int64_t n_items, int dim, float scale, float* __restrict__ out) {
  const int64_t total = n_items * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long x = ((unsigned long long)first_item * (unsigned long long)dim + (unsigned long long)i) ^ seed_mix;
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const long long s = (long long)((z & 0xFFFFull) + ((z >> 16) & 0xFFFFull) + ((z >> 32) & 0xFFFFull) + (z >> 48)) - 2 * 65535;
    out[i] = (float)s * scale;
  }
}

int hash_item_table(unsigned long long seed, int64_t first_item, int64_t n_items, int dim, float scale, float* out, hipStream_t stream) {
  const int64_t total = n_items * dim;
  if (total <= 0) return kOk;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(hash_item_table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed * 0xD1B54A32D192ED03ull, first_item, n_items, dim, scale, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace mol
