#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_bench; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 3), d["config"].get("exact_path"))
print("roofline", {k: d["roofline"][k] for k in ("kernel_ms", "achieved", "frac", "traffic")})
print("proved", {k: v for k, v in d["proved"].items() if k not in ("what",)})
print("fp32_dense", round(d["fp32_dense"]["value"]), d["fp32_dense"]["roofline"]["frac"])
for p in d.get("matrix", []): print("matrix", p["precision"], p["batch"], p["k_prime"], round(p["ms_per_step"], 3))
for w in d.get("other_workloads", []): print(w["workload"], w["precision"], round(w["ms_per_step"], 4), w.get("graph_replay_ms_per_step"), w.get("proved_calls"), w.get("runs_dense_fp32"))
for l in d.get("full_shards", []): print({k: v for k, v in l.items() if k not in ("recall",)})
print("hr_parity", d["hr_parity"]["parity"], d["hr_parity"]["identical_rows"], d["hr_parity"]["what"][:160])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["sample"])
for k, v in d["exact_fast_path"].items(): print(k, round(v["value"]), v["dense_fp32_fallbacks"], v.get("proved_calls"), v["eps"])
PY
grep -v amdgpu.ids $O/bench.err | tail -5
