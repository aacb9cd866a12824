#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=/root/repo/gpurun_out/r06w; mkdir -p $O
( time python -m pytest tests -q -m gpu -x ) > $O/suite.txt 2>&1
{ python tools/r06_shard_rccl_probe.py --world 8 --kc 1024; python tools/r06_shard_rccl_probe.py --world 8 --kc 1024 --host-times; python tools/r06_shard_rccl_probe.py --world 8 --kc 1024 --pipeline;
  python tools/shard_step_profile.py --world 8 --precision proved-global; } 2>&1 | grep -v "amdgpu\|verdict state\|^global proof\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp" | cut -c1-330 > $O/times.txt
python tools/algorithms_bench.py --workload amzn-books --algorithms MoLBruteForceTopK,MoLNaiveTopK5,MoLNaiveTopK100,MoLAvgTopK200,MoLCombTopK100_1000 > $O/algo.json 2> /dev/null
