#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s12; mkdir -p $O
for i in 1 2 3 4; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-path --no-matrix --no-other-workloads --no-hr-parity > $O/b$i.json 2> $O/b$i.err
  python -c "
import json; d=json.load(open('$O/b$i.json')); print('run $i value', round(d['value']), 'proved per-step', d['proved']['per_step_ms'], 'dense', round(d['fp32_dense']['value']), 'kernel', round(d['proved']['first_pass_kernel_ms'],3))"
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/b20.json 2> $O/b20.err
python -c "
import json; d=json.load(open('$O/b20.json')); print('20 steps value', round(d['value']), 'proved per-step', d['proved']['per_step_ms'], 'dense', round(d['fp32_dense']['value']))"
