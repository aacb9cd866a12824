#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r06v; mkdir -p $O
{ python tools/r06_shard_rccl_probe.py --world 8 --kc 1024; python tools/r06_shard_rccl_probe.py --world 8 --kc 1024 --host-times; python tools/r06_shard_rccl_probe.py --world 8 --kc 1024 --pipeline;
  python tools/shard_step_profile.py --world 8 --precision proved-global; } 2>&1 | grep -v "amdgpu\|verdict state\|^global proof\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp" | cut -c1-330 > $O/times.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python /root/repo/tools/r06_shard_rccl_probe.py --world 8 --kc 1024 --steps 100 > /dev/null 2>&1
cd /root/repo
python tools/r06_timeline.py $(find $O/tl -name '*kernel_trace.csv' | head -1) > $O/timeline.txt 2>&1
rm -rf $O/tl
