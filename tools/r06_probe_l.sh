cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/r06l; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --no-hr-parity > $O/bench.json 2> $O/bench.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); python /root/repo/tools/kernel_stats_top.py "$f" 20 > $O/top.txt; rm -rf $O/prof
