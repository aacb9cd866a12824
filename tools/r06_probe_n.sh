cd /root/repo; O=gpurun_out/r06n; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "component or naive or comb or f10 or F10 or large_batches or algorithms or filter or f3" 2>&1 | tail -4 > $O/test.txt
python tools/algorithms_bench.py --workload amzn-books --algorithms MoLBruteForceTopK,MoLNaiveTopK5,MoLNaiveTopK10,MoLNaiveTopK50,MoLNaiveTopK100,MoLCombTopK5_200,MoLCombTopK50_500,MoLCombTopK100_1000 > $O/algo.json 2> $O/algo.err
cd /tmp && export TMPDIR=/tmp
for alg in MoLNaiveTopK5 MoLNaiveTopK100; do
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_$alg -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms $alg > /dev/null 2>&1
f=$(find /root/repo/$O/prof_$alg -name "*kernel_stats.csv" | head -1); python /root/repo/tools/kernel_stats_top.py "$f" 18 > /root/repo/$O/top_$alg.txt; rm -rf /root/repo/$O/prof_$alg
done
