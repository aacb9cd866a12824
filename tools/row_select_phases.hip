// Phase timing of mol::row_select_kernel (workgroup 0): build with
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -DRAILS_TOPK_PHASES tools/row_select_phases.hip -o tools/row_select_phases
// Includes the product source directly; set_error is the only symbol it needs from capi.hip.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../rails_amd/csrc/topk.hip"
namespace mol { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); } }

int main(int argc, char** argv) {
  const int rows = 32, n = argc > 1 ? atoi(argv[1]) : 27278, k = argc > 2 ? atoi(argv[2]) : 200;
  std::vector<float> h((size_t)rows * n);
  std::mt19937 g(1); std::normal_distribution<float> d(0.f, 2.f);
  for (auto& x : h) x = d(g);
  float *sc, *os; int64_t* oi; void* ws;
  hipMalloc(&sc, h.size() * 4); hipMalloc(&os, rows * k * 4); hipMalloc(&oi, rows * k * 8);
  const size_t wsb = mol::topk_workspace_bytes(rows, n, k); hipMalloc(&ws, wsb);
  hipMemcpy(sc, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) mol::topk(sc, n, rows, n, k, nullptr, 0, os, oi, ws, wsb, 256, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long p[16]; hipMemcpyFromSymbol(p, HIP_SYMBOL(mol::g_phase), sizeof(p));
    printf("n=%d k=%d: %.1f us/call back-to-back | last workgroup-0 launch, ticks of 10 ns: load %lld  bound %lld  compact %lld  sort+emit %lld  candidates %lld\n", n, k,
           ms * 1e3 / 20, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[5]);
  }
  return 0;
}
