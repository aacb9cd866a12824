#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
for B in 8 32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$B -o p -- python tools/exact_step_profile.py --precisions proved --steps 30 --batch $B > $O/step_b$B.log 2>&1
  f=$(find $O/prof_b$B -name '*kernel_stats.csv' | head -1)
  echo "== B=$B $(grep proved $O/step_b$B.log | tail -1)"; python tools/kernel_stats_top.py "$f" 16 | tee $O/top_b$B.txt
  rm -rf $O/prof_b$B
done
timeout 900 python -m pytest tests/test_proved_gpu.py -q -s > $O/proved_tests.log 2>&1; echo "proved tests rc=$?"
grep -E "passed|failed" $O/proved_tests.log | tail -3
grep -E "f16 MFMA|v_exp|planted|stressed" $O/proved_tests.log
