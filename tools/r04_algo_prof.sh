cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/algo_r04; mkdir -p $O
python tools/algorithms_bench.py --workload amzn-books > $O/algorithms_amzn_books.json 2> $O/err.txt; tail -c 1500 $O/algorithms_amzn_books.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_naive -- python tools/algorithms_bench.py --workload amzn-books --algorithms MoLNaiveTopK50 > $O/naive50.json 2>> $O/err.txt
python tools/kernel_stats_top.py $(find $O/prof_naive -name '*kernel_stats.csv' | head -1) 16
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_avg -- python tools/algorithms_bench.py --workload amzn-books --algorithms MoLAvgTopK1000 > $O/avg1000.json 2>> $O/err.txt
python tools/kernel_stats_top.py $(find $O/prof_avg -name '*kernel_stats.csv' | head -1) 14
