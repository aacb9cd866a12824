cd /root/repo
for d in normal narrow; do for k in 200 2561; do python tools/topk_bench.py --k $k --dist $d; done; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/topk_prof -- python tools/topk_bench.py --k 2561 --dist narrow > /dev/null 2>&1
f=$(find gpurun_out/topk_prof -name '*kernel_stats.csv' | head -1); cut -d, -f1-4 $f | cut -c1-120 | head -12
