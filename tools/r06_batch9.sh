#!/bin/bash
cd /root/repo
O=gpurun_out/r06i; mkdir -p $O
python tools/algorithms_bench.py --workload amzn-books > $O/algorithms_amzn_books.json 2> $O/algo.err
( time python -m pytest tests -q -m gpu --durations=40 ) > $O/gpu_suite.txt 2>&1
