#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
for B in 8 32; do
  rocprofv3 --kernel-trace --stats -d $O/prof_b$B -o p -- python tools/exact_step_profile.py --precisions proved --steps 30 --batch $B > $O/step_b$B.log 2>&1
  f=$(find $O/prof_b$B -name '*kernel_stats.csv' | head -1)
  echo "== B=$B $(tail -1 $O/step_b$B.log)"; python tools/kernel_stats_top.py "$f" 14
done
timeout 900 python -m pytest tests/test_proved_gpu.py -q -s > $O/proved_tests.log 2>&1; echo "proved tests rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_proved_gpu.py > $O/gpu_suite.log 2>&1; echo "suite rc=$?"
grep -E "passed|failed" $O/proved_tests.log | tail -3; grep -E "^FAILED|passed|failed" $O/gpu_suite.log | tail -12
grep -E "f16 MFMA|v_exp" $O/proved_tests.log
