#!/bin/bash
cd /root/repo
O=gpurun_out/r06e; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "component or naive or comb or f10 or F10 or large_batches or algorithms or coarse or two_pass or avg" 2>&1 | tail -15 > $O/test_comp.txt
python tools/algorithms_bench.py --workload amzn-books > $O/algorithms_amzn_books.json 2> $O/algo.err
for w in 128 256 384; do
  echo "== RAILS_CAND_WGS=$w" >> $O/cand_wgs.txt
  RAILS_CAND_WGS=$w python tools/exact_step_profile.py --precisions proved --steps 200 2>&1 | grep -v amdgpu | cut -c1-90 >> $O/cand_wgs.txt
done
cd /tmp && export TMPDIR=/tmp
for alg in MoLNaiveTopK5 MoLNaiveTopK100 MoLCombTopK100_1000; do
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_$alg -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms $alg > /root/repo/$O/$alg.json 2> /root/repo/$O/$alg.err
done
for w in 128 256; do
RAILS_CAND_WGS=$w rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_w$w -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 60 > /root/repo/$O/prof_w$w.log 2>&1
done
cd /root/repo
for n in MoLNaiveTopK5 MoLNaiveTopK100 MoLCombTopK100_1000 w128 w256; do f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); python tools/kernel_stats_top.py "$f" 18 > $O/top_$n.txt 2>&1; done
rm -rf $O/prof_*/
