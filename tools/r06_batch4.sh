#!/bin/bash
cd /root/repo
O=gpurun_out/r06d; mkdir -p $O
for w in 512 2048 4096; do
  echo "== RAILS_CAND_WGS=$w" >> $O/cand_wgs.txt
  RAILS_CAND_WGS=$w python tools/exact_step_profile.py --precisions proved --steps 200 2>&1 | grep -v amdgpu | cut -c1-90 >> $O/cand_wgs.txt
done
python tools/exact_step_profile.py --precisions proved --steps 200 2>&1 | grep -v amdgpu | cut -c1-90 >> $O/cand_wgs.txt
cd /tmp && export TMPDIR=/tmp
for alg in MoLNaiveTopK5 MoLNaiveTopK100 MoLCombTopK100_1000; do
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_$alg -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms $alg > /root/repo/$O/$alg.json 2> /root/repo/$O/$alg.err
done
for w in 512 2048; do
RAILS_CAND_WGS=$w rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_w$w -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 60 > /root/repo/$O/prof_w$w.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_fin -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 60 > /root/repo/$O/prof_fin.log 2>&1
cd /root/repo
for n in MoLNaiveTopK5 MoLNaiveTopK100 MoLCombTopK100_1000 w512 w2048 fin; do f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); python tools/kernel_stats_top.py "$f" 16 > $O/top_$n.txt 2>&1; done
f=$(find $O/prof_MoLNaiveTopK5 -name "*kernel_trace.csv" | head -1); cp "$f" $O/kernel_trace_naive5.csv
rm -rf $O/prof_*/
