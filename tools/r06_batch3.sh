#!/bin/bash
cd /root/repo
O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_candidates_gpu.py -x -q 2>&1 | tail -5 > $O/test_candidates.txt
python -m pytest tests/test_proved_gpu.py -x -q -k "proved_mode or small_batches or per_pair or unprovable" 2>&1 | tail -5 > $O/test_proved.txt
python -m pytest tests/test_sharded_gpu.py -x -q -k "70001-None or 331-None" 2>&1 | tail -5 > $O/test_sharded.txt
for ft in 0 1; do
  echo "== RAILS_FUSED_TAIL=$ft" >> $O/ab.txt
  RAILS_FUSED_TAIL=$ft python tools/exact_step_profile.py --precisions proved --steps 200 2>&1 | grep -v amdgpu >> $O/ab.txt
  RAILS_FUSED_TAIL=$ft python tools/exact_step_profile.py --precisions proved --steps 200 --batch 8 2>&1 | grep -v amdgpu >> $O/ab.txt
  RAILS_FUSED_TAIL=$ft python tools/exact_step_profile.py --precisions proved --steps 400 --workload ml-20m 2>&1 | grep -v amdgpu >> $O/ab.txt
done
python tools/shard_step_profile.py --world 8 --precision proved-global 2>&1 | grep world >> $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 4; do
RAILS_FINISH_DEBUG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_d$dbg -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 60 > /root/repo/$O/prof_d$dbg.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_r8 -o t -- python /root/repo/tools/shard_step_profile.py --world 8 --precision proved-global --steps 100 > /root/repo/$O/prof_r8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_c2 -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 100 --workload ml-20m > /root/repo/$O/prof_c2.log 2>&1
cd /root/repo
for n in d0 d1 d2 d4 r8 c2; do f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); python tools/kernel_stats_top.py "$f" 14 > $O/top_$n.txt 2>&1; done
f=$(find $O/prof_d0 -name "*kernel_trace.csv" | head -1); cp "$f" $O/kernel_trace_c3.csv
f=$(find $O/prof_c2 -name "*kernel_trace.csv" | head -1); cp "$f" $O/kernel_trace_c2.csv
rm -rf $O/prof_*/
