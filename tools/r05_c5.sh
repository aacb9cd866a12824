#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_c5; mkdir -p $O
python bench.py --workload synthetic-8x8x32 --two-pass 1000 --no-recall --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 20 --warmup 3 > $O/two_pass.json 2> $O/two_pass.err
python -c "
import json; d=json.load(open('$O/two_pass.json')); print('plain', round(d['ms_per_step'],4), 'pipelined', d['pipelined']['ms_per_step'], 'roofline', d['roofline']['kernel_ms'], d['roofline']['frac'])"
