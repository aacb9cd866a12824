#!/bin/bash
cd /root/repo
O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_sharded_gpu.py -x -q --durations=10 2>&1 | tail -25 > $O/test_sharded.txt
python -m pytest tests/test_full_shard_gpu.py -x -q -k "verified" 2>&1 | tail -8 > $O/test_c4.txt
for R in 8 2; do
  python tools/shard_step_profile.py --world $R --precision proved-global >> $O/shard_steps.txt 2>&1
  python tools/shard_step_profile.py --world $R --precision proved-global --pipeline >> $O/shard_steps.txt 2>&1
  python tools/shard_step_profile.py --world $R >> $O/shard_steps.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_c3 -o c3 -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 50 > /root/repo/$O/prof_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_c2 -o c2 -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 100 --workload ml-20m > /root/repo/$O/prof_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_r8 -o r8 -- python /root/repo/tools/shard_step_profile.py --world 8 --precision proved-global --steps 100 > /root/repo/$O/prof_r8.log 2>&1
cd /root/repo
for n in c3 c2 r8; do f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); python tools/kernel_stats_top.py "$f" 30 > $O/top_$n.txt 2>&1; cp "$f" $O/kernel_stats_$n.csv; done
rm -rf $O/prof_c3 $O/prof_c2 $O/prof_r8
