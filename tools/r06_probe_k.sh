cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/r06k; mkdir -p $O
python /root/repo/tools/r06_scale_probe.py 1.0 > $O/s1.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python /root/repo/tools/r06_scale_probe.py 1.5 > $O/s15.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); python /root/repo/tools/kernel_stats_top.py "$f" 14 > $O/top.txt; rm -rf $O/prof
cd /root/repo; python -m pytest tests -q -m gpu --durations=70 -x 2>&1 | grep -v amdgpu | tail -80 > $O/durations.txt
