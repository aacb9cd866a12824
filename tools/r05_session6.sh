#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py --steps 5 --warmup 1 --no-other-workloads --no-cpu-baseline --no-hr-parity --no-fast-path > $O/bench.json 2> $O/bench.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); python tools/kernel_stats_top.py "$f" 30 | tee $O/top.txt
rm -rf $O/prof
python - <<'PY'
import json
d=json.load(open("gpurun_out/s6/bench.json"))
for p in d["matrix"]: print(p["precision"],p["batch"],p["k_prime"],round(p["ms_per_step"],3),p.get("proved_calls"),p.get("dense_fp32_fallbacks"))
PY
