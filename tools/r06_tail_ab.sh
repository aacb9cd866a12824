#!/bin/bash
# round 6: fused tail of the proved flow -- tests, same-box A/B of the step (RAILS_FUSED_TAIL=0: the round-5 launches), kernel trace
cd /root/repo
mkdir -p gpurun_out/r06
python -m pytest tests/test_candidates_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r06/test_candidates.txt
python -m pytest tests/test_proved_gpu.py -x -q -k "proved_mode or small_batches or per_pair or unprovable" 2>&1 | tail -15 > gpurun_out/r06/test_proved.txt
for rep in 1 2; do
for ft in 0 1; do
  echo "== RAILS_FUSED_TAIL=$ft" >> gpurun_out/r06/ab.txt
  RAILS_FUSED_TAIL=$ft python tools/exact_step_profile.py --precisions proved,fp32 --steps 100 >> gpurun_out/r06/ab.txt 2>&1
  RAILS_FUSED_TAIL=$ft python tools/exact_step_profile.py --precisions proved --steps 100 --batch 8 >> gpurun_out/r06/ab.txt 2>&1
  RAILS_FUSED_TAIL=$ft python tools/exact_step_profile.py --precisions proved,fp32 --steps 200 --workload ml-20m >> gpurun_out/r06/ab.txt 2>&1
  RAILS_FUSED_TAIL=$ft python tools/exact_step_profile.py --precisions proved --steps 200 --items 86971 >> gpurun_out/r06/ab.txt 2>&1
done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r06/prof -o tail -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 50 > /root/repo/gpurun_out/r06/prof.log 2>&1
cd /root/repo
f=$(find gpurun_out/r06/prof -name "*kernel_stats.csv" | head -1); python tools/kernel_stats_top.py "$f" 30 > gpurun_out/r06/prof_top.txt 2>&1; cp "$f" gpurun_out/r06/kernel_stats.csv
