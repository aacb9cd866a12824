// Launch side of the exact-fp32 scoring kernels (the unit arithmetic is in mol_score_fp32_unit.h) and the k-split kernel
// for L = 256.
#include "mol_score_fp32_unit.h"

namespace mol {

// ---------------------------------------------------------------------------------------------
// Kernel C ("ksplit"): shapes whose logit axis is too long for the register-resident scheme above
// (16x16x64: L = 256 -> a whole D1 would be 256 registers and the gate weights 256 KiB).
// One wave per SIMD (4 waves, 512 registers).  Per unit = (query group of 32/P_Q queries, tile of 32 items):
//   pass 1     for each chunk of MC item groups: GEMM1 of the chunk (D1c), then the chunk's K-slice of GEMM2
//              accumulated into every query's D2                                  (W1 fragments in LDS)
//   per query  silu(D2) -> GEMM3 (W2 fragments streamed from L2: they do not fit LDS next to W1)
//              -> u, min, ex = 2^(min u - u) kept in the D3 registers, den
//   sweep      GEMM1 of every chunk again (cl is not kept: QT x L/2 values per lane) -> num += ex * cl
// MFMA cost: GEMM1 twice, i.e. (2*2Ld + 4LH) / (2Ld + 4LH) = 1.2x the algorithmic flops for 16x16x64.
// ---------------------------------------------------------------------------------------------
// Buffer addressing (SGPR descriptor + scalar byte offset + one per-lane VGPR offset) for the ksplit kernel: its
// operands span 128-160 KiB per base, far beyond a global load's +-4 KiB immediate, and flat addressing made
// the compiler keep (and spill) one 64-bit address pair per fragment.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct FragBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ FragBuf(const void* base, unsigned bytes)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000)) {}
  // float4 fragment `idx` (1 KiB per wave): lane reads 16 B at idx*1024 + lane*16
  __device__ __forceinline__ float4 frag(int idx, int lane16) const {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, idx * 1024, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
};

#ifdef RAILS_SCORE_PHASES   // tools/ksplit_phases.sh: wall-clock stamps (100 MHz) of workgroup 0 / wave 0's first unit
__device__ long long g_sphase[16];
#define RAILS_SPHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && it == 0) g_sphase[i] = (long long)wall_clock64(); } while (0)
#else
#define RAILS_SPHASE(i)
#endif

// `bnext` carries block 0 of THIS chunk, requested by the previous call (or by the unit prologue); before returning, block 0
// of chunk `next_chunk` (< 0: none) is requested into it.  The chunk order of a unit is 0,1,2,3 (pass 1), 0,1 (sweep of
// half 0), 2,3 (sweep of half 1): eight calls whose first-block round trip used to be exposed (~2 us each of a 118 us unit).
template <int MC, int DD>
__device__ __forceinline__ void gemm1_chunk(f32x16 (&D1c)[MC], const FragBuf& eq, const FragBuf& tile, int chunk,
                                            int next_chunk, float4 (&bnext)[4], int lane16) {
#pragma unroll
  for (int m = 0; m < MC; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) D1c[m][r] = 0.0f;
  // Blocks of 4 fragment pairs (16 MFMAs, ~1000 cycles); the HBM/L2-side operand is double buffered (+16 VGPRs; buffering
  // the L1-resident eq fragments as well only added spills): block i+1's loads are issued before block i's MFMAs.
  constexpr int NB = DD / 32;          // blocks per item group
  constexpr int NBLK = MC * NB;
  float4 a[4], b[2][4];
  auto load_b = [&](int c, int i, float4 (&dst)[4]) {
    const int m = i / NB, s0 = (i % NB) * 4;
#pragma unroll
    for (int sc = 0; sc < 4; ++sc) dst[sc] = tile.frag((c * MC + m) * (DD / 8) + s0 + sc, lane16);
  };
#pragma unroll
  for (int sc = 0; sc < 4; ++sc) b[0][sc] = bnext[sc];
#pragma unroll
  for (int i = 0; i < NBLK; ++i) {
    const int m = i / NB, s0 = (i % NB) * 4;
#pragma unroll
    for (int sc = 0; sc < 4; ++sc) a[sc] = eq.frag(s0 + sc, lane16);  // L1-resident, re-read instead of pinning DD/2 registers
    if (i + 1 < NBLK) load_b(chunk, i + 1, b[(i + 1) & 1]);
    else if (next_chunk >= 0) load_b(next_chunk, 0, bnext);
    asm volatile("" ::: "memory");  // bound the operands in flight: exactly one block ahead
#pragma unroll
    for (int sc = 0; sc < 4; ++sc) {
      D1c[m] = mfma32(a[sc].x, b[i & 1][sc].x, D1c[m]);
      D1c[m] = mfma32(a[sc].y, b[i & 1][sc].y, D1c[m]);
      D1c[m] = mfma32(a[sc].z, b[i & 1][sc].z, D1c[m]);
      D1c[m] = mfma32(a[sc].w, b[i & 1][sc].w, D1c[m]);
    }
  }
}

// GEMM3 + gate for ONE HALF of the logit axis of one query (row tiles [HALF*TL/2, (HALF+1)*TL/2) of W2, i.e.
// K-steps e in [HALF*E/2, (HALF+1)*E/2)), with the online-softmax bookkeeping: on return D3h holds
// ex = 2^(mn - u) against the updated running minimum `mn` (both lane halves agree on it), `den` and `num`
// have been rescaled to it and den has this half's ex added (this lane half's partial sums).
template <class G, int HALF>
__device__ __forceinline__ void ksplit_half_gate(const f32x16 (&D2q)[G::TH], f32x16 (&D3h)[G::TL / 2],
                                                 const FragBuf& gW2, const float* sB2, const float4 (&gih)[G::E / 8],
                                                 const float4* gq4, int lane16, int hi, float& mn,
                                                 float& den, f32x2& num) {
  constexpr int HV = G::TL / 2;       // row tiles per half
  constexpr int EH = G::E / 2;        // K-steps (per lane values) per half
  constexpr int E0 = HALF * EH;
#pragma unroll
  for (int v = 0; v < HV; ++v)
#pragma unroll
    for (int r = 0; r < 16; ++r) D3h[v][r] = sB2[hi * G::E + E0 + v * 16 + r];
  // W2 fragments from L2, one K-chunk (4 hidden steps x HV row tiles) ahead of their MFMAs
  float4 cur[HV], nxt[HV];
#pragma unroll
  for (int v = 0; v < HV; ++v) cur[v] = gW2.frag(HALF * HV + v, lane16);
#pragma unroll
  for (int fc = 0; fc < G::F / 4; ++fc) {
    if (fc + 1 < G::F / 4) {
#pragma unroll
      for (int v = 0; v < HV; ++v) nxt[v] = gW2.frag((fc + 1) * G::TL + HALF * HV + v, lane16);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const float av[4] = {cur[v].x, cur[v].y, cur[v].z, cur[v].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = fc * 4 + j;
        D3h[v] = mfma32(av[j], D2q[f / 16][f % 16], D3h[v]);
      }
    }
#pragma unroll
    for (int v = 0; v < HV; ++v) cur[v] = nxt[v];
  }
  __builtin_amdgcn_sched_barrier(0);
  float lmn = INFINITY;
#pragma unroll
  for (int ec = 0; ec < EH / 4; ++ec) {
    const float4 gi = gih[ec];            // this half's item-gate fragments, loaded once for both queries by the caller
    const float4 gq = gq4[E0 / 4 + ec];   // the query's gate row, staged in LDS at the start of the unit
    const f32x2 giv[2] = {{gi.x, gi.y}, {gi.z, gi.w}};
    const f32x2 gqv[2] = {{gq.x, gq.y}, {gq.z, gq.w}};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = ec * 4 + 2 * j;  // local to the half
      const f32x2 t2 = pk_fma(gqv[j], giv[j], f32x2{D3h[e / 16][e % 16], D3h[e / 16][e % 16 + 1]});
      const f32x2 uu = t2 * pk_sigmoid_arg(t2);
      D3h[e / 16][e % 16] = uu.x;
      D3h[e / 16][e % 16 + 1] = uu.y;
      lmn = fminf(lmn, fminf(uu.x, uu.y));
    }
  }
  lmn = fminf(lmn, xor32(lmn));
  const float mnew = fminf(mn, lmn);
  // rescale what was accumulated against the old minimum (2^(mnew - mn) <= 1; first half: mn = +inf -> factor 0)
  const float scale = __builtin_amdgcn_exp2f(mnew - mn);
  den *= scale;
  num = num * scale;
  mn = mnew;
  f32x2 den2 = {0.0f, 0.0f};
#pragma unroll
  for (int e = 0; e < EH; e += 2) {
    const f32x2 d = mn - f32x2{D3h[e / 16][e % 16], D3h[e / 16][e % 16 + 1]};
    const f32x2 ex = {__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
    D3h[e / 16][e % 16] = ex.x;
    D3h[e / 16][e % 16 + 1] = ex.y;
    den2 = den2 + ex;
  }
  den += den2.x + den2.y;
  __builtin_amdgcn_sched_barrier(0);
}

template <int PQ, int PX, int DD, int H, int MC>
__global__ __launch_bounds__(256, 1) void mol_score_ksplit_kernel(ScoreArgs p) {
  using G = Geo<PQ, PX, DD, H>;
  constexpr int NW = 4;
  constexpr int NCH = PX / MC;      // chunks of item groups
  constexpr int ECH = MC * G::RPQ;  // K-steps over the logit axis per chunk
  static_assert(PX % (2 * MC) == 0 && ECH % 16 == 0, "chunks must align with the halves and the D3 register tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const float4* sW1 = reinterpret_cast<const float4*>(smem);
  const float* sB1 = smem + G::kW1Floats;
  const float* sB2 = sB1 + H;
  float* sGq = smem + G::kW1Floats + H + G::L;   // [NW][QT][L]: the current unit's query-gate rows, per wave
  {  // W1 fragments + both bias vectors into LDS; W2 stays in HBM/L2
    const float4* src = reinterpret_cast<const float4*>(p.wpack);
    float4* dst = reinterpret_cast<float4*>(smem);
    for (int i = threadIdx.x; i < G::kW1Floats / 4; i += NW * 64) dst[i] = src[i];
    const float4* srcb = reinterpret_cast<const float4*>(p.wpack + G::kW1Floats + G::kW2Floats);
    float4* dstb = reinterpret_cast<float4*>(smem + G::kW1Floats);
    for (int i = threadIdx.x; i < (H + G::L) / 4; i += NW * 64) dstb[i] = srcb[i];
  }
  __syncthreads();
  const FragBuf gW2(p.wpack + G::kW1Floats, G::kW2Floats * 4);

  const int lane = threadIdx.x & 63;
  const int lane16 = lane * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, x = lane & 31;
  const int inner = p.per_row ? (int)p.n_tiles : p.n_groups;
  const int64_t n_units = p.per_row ? (int64_t)p.B * p.n_tiles : p.n_tiles * p.n_groups;
  // Full rounds: the NW waves of a workgroup take NW consecutive units (the query groups of one item tile, so the tile's
  // fragments are shared through L1).  The leftover round is dealt wave-major instead -- unit r goes to workgroup
  // r % grid, wave r / grid -- so that it lands one unit per SIMD across the whole chip rather than two per SIMD on the
  // first few CUs (ML-20M: 6824 units on 1024 SIMDs, worst SIMD 7 passes instead of 8; ML-1M: all 256 CUs busy).
  const int64_t stride = (int64_t)gridDim.x * NW;
  const int64_t rounds = n_units / stride;
  // XCD-aware numbering: hardware workgroup b runs on XCD b % 8 (each XCD has its own L2).  A tile's 16 query groups are
  // 4 consecutive workgroups' worth of units; with the hardware numbering those land on 4 different XCDs and the tile is
  // fetched from HBM into 4 L2s (PMC: 9 x the index bytes).  Logical id = (b % 8) * (grid / 8) + b / 8 keeps consecutive
  // logical workgroups on one XCD.
  const int64_t bx = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  for (int64_t it = 0; it <= rounds; ++it) {
    const int64_t u = it < rounds ? it * stride + bx * NW + wave
                                  : rounds * stride + (int64_t)wave * gridDim.x + bx;
    if (u >= n_units) break;
    const int64_t outer = u / inner;
    const int innr = (int)(u - outer * inner);
    const int64_t tile = p.per_row ? innr : outer;
    const int row = p.per_row ? (int)outer : -1;
    const int g = p.per_row ? row / G::QT : innr;
    const int64_t tile_addr = p.per_row ? (int64_t)row * p.n_tiles + tile : tile;
    const FragBuf tileb(p.ipack + tile_addr * (int64_t)G::kTileFloats, G::kTileFloats * 4);
    const FragBuf eqb(p.eqfrag + (int64_t)g * G::kEqGroupFloats, G::kEqGroupFloats * 4);
    bool active[G::QT];
#pragma unroll
    for (int q = 0; q < G::QT; ++q) active[q] = (g * G::QT + q < p.B) && (row < 0 || g * G::QT + q == row);

    // the unit's gq rows -> this wave's LDS slot (read back by the same wave only: DS operations of a wave are in order).
    // Loading them (and gi) inside the gate loop exposed 4 round trips per call, 16 per unit: ~32 of a unit's 136 us.
    float4 bnext[4];
#pragma unroll
    for (int sc = 0; sc < 4; ++sc) bnext[sc] = tileb.frag(sc, lane16);   // chunk 0, block 0
    static_assert(G::L == 256, "one float4 per lane per query");
    float4* sGqW = reinterpret_cast<float4*>(sGq + wave * G::QT * G::L);
#pragma unroll
    for (int q = 0; q < G::QT; ++q) {
      // queries past the batch end (padding of the last group) run on zero operands; their store is skipped
      const int qq = (g * G::QT + q < p.B) ? g * G::QT + q : p.B - 1;
      sGqW[q * (G::L / 4) + lane] = *reinterpret_cast<const float4*>(p.gqfrag + (int64_t)qq * G::L + lane * 4);
    }
    RAILS_SPHASE(0);
    // ---- pass 1: chunked GEMM1 -> K-slices of GEMM2 (t = -log2e * pre)
    f32x16 D2[G::QT][G::TH];
#pragma unroll
    for (int q = 0; q < G::QT; ++q)
#pragma unroll
      for (int t = 0; t < G::TH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) D2[q][t][r] = sB1[t * 32 + hi * 16 + r];
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      f32x16 D1c[MC];
      gemm1_chunk<MC, DD>(D1c, eqb, tileb, c, (c + 1) % NCH, bnext, lane16);   // after the last chunk: chunk 0 again (sweep)
      if (c == 0) RAILS_SPHASE(1);
#pragma unroll
      for (int q = 0; q < G::QT; ++q) {
        {
          // W1 fragments of K-step group es + 1 are read (LDS) before group es's 16 MFMAs: with one wave per SIMD the
          // ~150-cycle ds_read latency ahead of every group was exposed.  Exactly one group ahead (16 VGPRs).
          float4 wa[G::TH], wn[G::TH];
#pragma unroll
          for (int t = 0; t < G::TH; ++t) wa[t] = sW1[((c * (ECH / 4)) * G::TH + t) * 64 + lane];
#pragma unroll
          for (int es = 0; es < ECH / 4; ++es) {
            if (es + 1 < ECH / 4) {
#pragma unroll
              for (int t = 0; t < G::TH; ++t) wn[t] = sW1[((c * (ECH / 4) + es + 1) * G::TH + t) * 64 + lane];
            }
            asm volatile("" ::: "memory");  // keep later K-steps' LDS reads below these MFMAs (register pressure)
#pragma unroll
            for (int t = 0; t < G::TH; ++t) {
              const float av[4] = {wa[t].x, wa[t].y, wa[t].z, wa[t].w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int e = es * 4 + j;
                D2[q][t] = mfma32(av[j], D1c[e / G::RPQ][q * G::RPQ + e % G::RPQ], D2[q][t]);
              }
            }
#pragma unroll
            for (int t = 0; t < G::TH; ++t) wa[t] = wn[t];
          }
        }
      }
    }

    // ---- hid' = t / (1 + 2^t), both queries, in place
    RAILS_SPHASE(3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < G::QT; ++q)
#pragma unroll
      for (int t = 0; t < G::TH; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 tv = {D2[q][t][r], D2[q][t][r + 1]};
          const f32x2 h = tv * pk_sigmoid_arg(tv);
          D2[q][t][r] = h.x;
          D2[q][t][r + 1] = h.y;
        }
    __builtin_amdgcn_sched_barrier(0);

    // ---- two halves of the logit axis, online softmax across them: GEMM3 half -> gate -> ex (kept in D3h),
    //      then the GEMM1 chunks of that half again for num += ex * cl (cl is not kept across pass 1)
    RAILS_SPHASE(4);
    float mn[G::QT], den[G::QT];
    f32x2 num[G::QT];
#pragma unroll
    for (int q = 0; q < G::QT; ++q) { mn[q] = INFINITY; den[q] = 0.0f; num[q] = f32x2{0.0f, 0.0f}; }
    [&]<int... HALF>(std::integer_sequence<int, HALF...>) {
      (
          [&] {
            f32x16 D3h[G::QT][G::TL / 2];
            // this half's item-gate fragments: requested here, consumed after the first query's GEMM3 (256 MFMAs later)
            float4 gih[G::E / 8];
#pragma unroll
            for (int ec = 0; ec < G::E / 8; ++ec) gih[ec] = tileb.frag(G::kTileExFloats / 256 + HALF * (G::E / 8) + ec, lane16);
#pragma unroll
            for (int q = 0; q < G::QT; ++q) {
              const float4* gq4 = reinterpret_cast<const float4*>(sGq + (wave * G::QT + q) * G::L + hi * G::E);
              ksplit_half_gate<G, HALF>(D2[q], D3h[q], gW2, sB2, gih, gq4, lane16, hi, mn[q], den[q], num[q]);
            }
            RAILS_SPHASE(5 + 2 * HALF);
#pragma unroll
            for (int cc = 0; cc < NCH / 2; ++cc) {
              const int c = HALF * (NCH / 2) + cc;
              f32x16 D1c[MC];
              gemm1_chunk<MC, DD>(D1c, eqb, tileb, c, c + 1 < NCH ? c + 1 : -1, bnext, lane16);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int q = 0; q < G::QT; ++q)
#pragma unroll
                for (int el = 0; el < ECH; el += 2) {
                  const int e = cc * ECH + el;  // K-step local to this half
                  num[q] = pk_fma(f32x2{D3h[q][e / 16][e % 16], D3h[q][e / 16][e % 16 + 1]},
                                  f32x2{D1c[el / G::RPQ][q * G::RPQ + el % G::RPQ],
                                        D1c[el / G::RPQ][q * G::RPQ + el % G::RPQ + 1]},
                                  num[q]);
                }
              __builtin_amdgcn_sched_barrier(0);
            }
            RAILS_SPHASE(6 + 2 * HALF);
          }(),
          ...);
    }(std::integer_sequence<int, 0, 1>{});

#pragma unroll
    for (int q = 0; q < G::QT; ++q) {
      const float dn = den[q] + xor32(den[q]);
      float nm = num[q].x + num[q].y;
      nm += xor32(nm);
      const float rden = __builtin_amdgcn_rcpf(dn);
      const float out = (nm * rden) / fmaxf(dn * rden, 1e-6f);
      const int64_t item = tile * kTileItems + x;
      if (active[q] && hi == 0 && item < p.n_items) p.logits[(int64_t)(g * G::QT + q) * p.ld + item] = out;
    }
  }
}

template <int PQ, int PX, int DD, int H, int MC>
static int launch_ksplit(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using G = Geo<PQ, PX, DD, H>;
  constexpr size_t lds = ((size_t)G::kW1Floats + H + G::L + 4 * G::QT * G::L) * sizeof(float);   // + per-wave gq rows
  static_assert(lds <= 160 * 1024, "W1 fragments must fit LDS");
  static DynLdsOnce once;
  if (ensure_dyn_lds(once, reinterpret_cast<const void*>(&mol_score_ksplit_kernel<PQ, PX, DD, H, MC>), (int)lds) != kOk) return kErrLaunch;
  const int64_t n_units = a.per_row ? (int64_t)a.B * a.n_tiles : a.n_tiles * a.n_groups;
  int64_t grid = (n_units + 3) / 4;
  if (grid > n_cu) grid = n_cu;
  if (grid < 1) return kOk;
  hipLaunchKernelGGL((mol_score_ksplit_kernel<PQ, PX, DD, H, MC>), dim3((unsigned)grid), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

template <int PQ, int PX, int DD, int H>
static int launch_score(const ScoreArgs& a, int n_cu, hipStream_t stream) {
  using U = Fp32Unit;
  const int variant = choose_variant<PQ, PX, DD, H>(a, n_cu);
  if ((variant == 2 || variant == 4 || variant == 5) && a.per_row) { set_error("staged scoring kernel does not do per-row candidates"); return kErrUnsupported; }
  switch (variant) {
    case 1: return launch_kernel<U, PQ, PX, DD, H, 8, false>(a, n_cu, stream);
    case 2: return launch_kernel<U, PQ, PX, DD, H, 8, true>(a, n_cu, stream);
    case 3: return launch_kernel<U, PQ, PX, DD, H, 4, false>(a, n_cu, stream);
    case 4: return launch_kernel<U, PQ, PX, DD, H, 4, true>(a, n_cu, stream);
    case 5: return launch_staged1<U, PQ, PX, DD, H, 8>(a, n_cu, stream);
    default: set_error("unknown RAILS_SCORE_VARIANT %d", variant); return kErrInvalid;
  }
}

bool score_supported(const Shape& s) {
  if (score_extra_shape(s)) return true;
  if (s.gating_qi_hidden_dim != 128) return false;
  const int pq = s.query_dot_product_groups, px = s.item_dot_product_groups, dd = s.dot_product_dimension;
  return (pq == 8 && px == 4 && dd == 64) || (pq == 8 && px == 4 && dd == 128) || (pq == 8 && px == 8 && dd == 32) ||
         (pq == 16 && px == 16 && dd == 64);
}

int score_launch(const Shape& s, const ScoreArgs& a, int n_cu, hipStream_t stream) {
  if (!score_supported(s)) return kErrUnsupported;
  if (score_extra_shape(s))
    return !a.split ? score_launch_extra(s, a, n_cu, stream) : a.single ? score_launch_f16x1_extra(s, a, n_cu, stream) : score_launch_f16_extra(s, a, n_cu, stream);
  if (a.split) return a.single ? score_launch_f16x1(s, a, n_cu, stream) : score_launch_f16(s, a, n_cu, stream);
#define MOL_CASE(pq, px, dd)                                                                             \
  if (s.query_dot_product_groups == pq && s.item_dot_product_groups == px && s.dot_product_dimension == dd) \
    return launch_score<pq, px, dd, 128>(a, n_cu, stream);
  MOL_CASE(8, 4, 64)
  MOL_CASE(8, 4, 128)
  MOL_CASE(8, 8, 32)
#undef MOL_CASE
  if (s.query_dot_product_groups == 16 && s.item_dot_product_groups == 16 && s.dot_product_dimension == 64) {
    return launch_ksplit<16, 16, 64, 128, 4>(a, n_cu, stream);
  }
  return kErrUnsupported;
}

#ifdef RAILS_SCORE_PHASES
}  // namespace mol
extern "C" int rails_debug_score_phases(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mol::g_sphase), sizeof(long long) * 16) == hipSuccess ? 0 : -1;
}
namespace mol {
#endif
}  // namespace mol
