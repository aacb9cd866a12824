cd /root/repo; O=/root/repo/gpurun_out/r06o; mkdir -p $O
python tools/exact_step_profile.py --precisions proved,fp32 --steps 500 --workload ml-20m 2>&1 | grep -v amdgpu > $O/ml20m.txt
python tools/exact_step_profile.py --precisions proved --steps 500 --workload ml-20m 2>&1 | grep -v amdgpu >> $O/ml20m.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 100 > $O/c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 200 --workload ml-20m > $O/c2.log 2>&1
for n in p1 p2; do f=$(find $O/$n -name "*kernel_stats.csv" | head -1); python /root/repo/tools/kernel_stats_top.py "$f" 14 > $O/top_$n.txt; rm -rf $O/$n; done
