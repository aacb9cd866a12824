#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=/root/repo/gpurun_out/r06y; mkdir -p $O; rm -f $O/out.txt
python -m pytest tests -x -q -m gpu -k "coarse or avg or two_pass or config5 or component or naive or comb or f10 or f9 or f8 or algorithms or prefilter or fused or sharded" 2>&1 | tail -4 > $O/test.txt
python tools/algorithms_bench.py --workload amzn-books > $O/algo.json 2> /dev/null
