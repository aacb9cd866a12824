#!/bin/bash
# Round-5 measurement batch (one gpurun call): bash tools/r05_measure.sh [tag] -> gpurun_out/r05_<tag>/   (copy what is judged into profiles/)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_${1:-final}; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
# the headline step alone under rocprofv3 (every launch of the proved step at B = 32)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o r05 -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --no-hr-parity > $O/bench_headline_under_profiler.json 2> $O/prof_headline.err
f=$(find $O/prof_headline -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_headline.csv; python tools/kernel_stats_top.py "$f" 24 > $O/kernel_stats_headline_top.txt; rm -rf $O/prof_headline
# HBM traffic and SQ counters of the first-pass kernel (f16x3), separate --pmc passes
bash tools/pmc_traffic.sh $O/pmc_traffic_f16x3 --workload amzn-books --batch 32 --precision f16x3 > $O/pmc_traffic_f16x3_summary.txt 2>&1
PMC_EXTRA="--workload amzn-books --batch 32 --precision f16x3" bash tools/pmc.sh $O/pmc_f16x3 0 > $O/pmc_f16x3_summary.txt 2>&1
rm -rf $O/pmc_traffic_f16x3 $O/pmc_f16x3
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.txt
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 3), d["config"].get("exact_path"))
print("roofline", {k: d["roofline"][k] for k in ("kernel_ms", "achieved", "frac", "traffic")})
print("proved", {k: v for k, v in d["proved"].items() if k not in ("what", "per_step_ms")})
print("fp32_dense", round(d["fp32_dense"]["value"]), d["fp32_dense"]["roofline"]["frac"])
for w in d.get("other_workloads", []): print(w["workload"], w["precision"], round(w["ms_per_step"], 4), w.get("graph_replay_ms_per_step"), w.get("proved_calls"), w.get("runs_dense_fp32"))
for l in d.get("full_shards", []): print({k: v for k, v in l.items() if k not in ("recall",)})
print("hr_parity", d["hr_parity"]["parity"], d["hr_parity"]["what"][:120])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
cat $O/kernel_stats_headline_top.txt | head -20; cat $O/pmc_traffic_f16x3_summary.txt | tail -8; tail -22 $O/pmc_f16x3_summary.txt
