#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/r06f; mkdir -p $O
run() { tag=$1; shift; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms MoLNaiveTopK5 > $O/$tag.json 2> $O/$tag.err; f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); python /root/repo/tools/kernel_stats_top.py "$f" 8 > $O/top_$tag.txt 2>&1; rm -rf $O/prof_$tag; }
run base X=1
run nohits RAILS_COMP_DEBUG=1
run s4 RAILS_COMP_STRIDE=4
run s1 RAILS_COMP_STRIDE=1
run s1nohits RAILS_COMP_STRIDE=1 RAILS_COMP_DEBUG=1
