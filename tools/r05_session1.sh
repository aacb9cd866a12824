#!/bin/bash
# round 5, first GPU session: arithmetic-model probes + proved-mode tests, top-k timings at candidate counts, a short bench line, the GPU suite
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests/test_proved_gpu.py -q -s -x > gpurun_out/s1/proved_tests.log 2>&1; echo "proved tests rc=$?"
timeout 300 python - > gpurun_out/s1/topk_kc.log 2>&1 <<'PY'
import torch, time
from rails_amd import engine as E
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 695762), device=dev, generator=g) * 3
for k in (200, 288, 512, 544, 640, 768, 1024, 1536, 2048, 4096, 5152, 8192):
    ws = torch.empty(E._lib.load().rails_topk_workspace_bytes(32, 695762, k), dtype=torch.uint8, device=dev)
    for _ in range(3):
        E.topk(x, k, workspace=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        E.topk(x, k, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    flag = torch.ones(1, dtype=torch.int32, device=dev)
    s = torch.empty((32, k), device=dev); i = torch.empty((32, k), dtype=torch.int64, device=dev)
    try:
        for _ in range(3):
            E.topk(x, k, workspace=ws, out=(s, i), run_if=flag)
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        for _ in range(20):
            E.topk(x, k, workspace=ws, out=(s, i), run_if=flag)
        f1.record(); torch.cuda.synchronize()
        pred = f0.elapsed_time(f1) / 20 * 1e3
    except Exception as ex:
        pred = str(ex)[:80]
    print(f"k={k:5d}  topk {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us   under a launch predicate (two-level to 4096): {pred}")
PY
echo "topk rc=$?"
timeout 900 python bench.py --steps 10 --warmup 3 --no-other-workloads --no-cpu-baseline > gpurun_out/s1/bench_short.json 2> gpurun_out/s1/bench_short.err; echo "bench rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_proved_gpu.py > gpurun_out/s1/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -5 gpurun_out/s1/proved_tests.log; tail -3 gpurun_out/s1/gpu_suite.log; cat gpurun_out/s1/topk_kc.log | tail -14
