#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/s3
timeout 1200 python -m pytest tests/test_proved_gpu.py -q -s > gpurun_out/s3/proved_tests.log 2>&1; echo "proved tests rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_proved_gpu.py > gpurun_out/s3/gpu_suite.log 2>&1; echo "suite rc=$?"
timeout 900 python bench.py --steps 20 --warmup 3 --no-other-workloads --no-cpu-baseline > gpurun_out/s3/bench_short.json 2> gpurun_out/s3/bench_short.err; echo "bench rc=$?"
grep -E "passed|failed" gpurun_out/s3/proved_tests.log | tail -3; grep -E "^FAILED|passed|failed" gpurun_out/s3/gpu_suite.log | tail -12
