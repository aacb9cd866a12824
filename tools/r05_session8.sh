#!/bin/bash
# the per-pair upper bound of the 16x16x64 first pass: kernel test, module tests, the full config-4 shard (test + bench legs)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s18; mkdir -p $O
python -m pytest tests/test_proved_gpu.py -x -q -m gpu -s -k "upper or default_and_equals" > $O/proved_tests.txt 2>&1; echo "proved tests rc=$?"; grep -E "16x16x64|passed|failed|Error|assert" $O/proved_tests.txt | cut -c1-300 | tail -30
timeout 1500 python -m pytest tests/test_full_shard_gpu.py -x -q -m gpu -s -k "config4" > $O/c4_tests.txt 2>&1; echo "c4 shard tests rc=$?"; grep -E "proved mode|passed|failed|Error|assert" $O/c4_tests.txt | cut -c1-400 | tail -20
python bench.py --workload synthetic-16x16x64 --no-cpu-baseline --no-matrix --no-other-workloads --no-hr-parity --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"; tail -3 $O/bench_c4.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s18/bench_c4.json'))
print('value',d['value'],'ms',d['ms_per_step'],d['config'])
for k in ('proved','fp32_dense','fast_path','exact_fast_path','roofline'):
    print(k, json.dumps(d.get(k))[:900])
PY
