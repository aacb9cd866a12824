#!/bin/bash
cd /root/repo
O=gpurun_out/r06g; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "component or naive or comb or f10 or F10 or large_batches or algorithms or coarse or two_pass or avg" 2>&1 | tail -15 > $O/test_comp.txt
python tools/algorithms_bench.py --workload amzn-books > $O/algorithms_amzn_books.json 2> $O/algo.err
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; alg=$2; shift; shift; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_$tag -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms $alg > /root/repo/$O/$tag.json 2> /root/repo/$O/$tag.err; f=$(find /root/repo/$O/prof_$tag -name "*kernel_stats.csv" | head -1); python /root/repo/tools/kernel_stats_top.py "$f" 16 > /root/repo/$O/top_$tag.txt 2>&1; rm -rf /root/repo/$O/prof_$tag; }
run n5 MoLNaiveTopK5 X=1
run n5s8 MoLNaiveTopK5 RAILS_COMP_STRIDE=8
run n5s2 MoLNaiveTopK5 RAILS_COMP_STRIDE=2
run n5nh MoLNaiveTopK5 RAILS_COMP_DEBUG=1
run n100 MoLNaiveTopK100 X=1
run n100s8 MoLNaiveTopK100 RAILS_COMP_STRIDE=8
run c100 MoLCombTopK100_1000 X=1
