"""bench.py's output contract (the driver parses ONE JSON line): checked on the committed line of the last GPU run
(profiles/r02_bench.json) on CPU, and on a live short run on the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
CPU_BASELINE = {"value", "unit", "cores", "kind", "sample"}


def check_line(d, expect_cpu_baseline):
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["higher_is_better"] is True and d["unit"] == "queries/s" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["scaling"] in ("strong", "weak") and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) / d["value"] < 1e-6
    r = d["roofline"]
    assert ROOFLINE <= set(r) and r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    if expect_cpu_baseline:
        c = d["cpu_baseline"]
        assert CPU_BASELINE <= set(c) and c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
    check_line(d, expect_cpu_baseline=True)
    assert d["n_gpus"] == 1 and "amzn-books" in d["config"]["workload"]
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["traffic"] is not None and "profiles/" in d["roofline"]["traffic_source"]
    assert d["ms_per_step_stdev"] >= 0 and d["cpu_baseline"]["physical_cores"] >= 1 and "scaled" not in d["cpu_baseline"]["sample"]
    f = d["fast_path"]["roofline"]
    assert f["bound"] == "mfma" and abs(f["frac"] - f["achieved"] / f["peak"]) < 1e-9 and abs(f["peak"] - 2500 / 3) < 1e-6
    pts = {(p["precision"], p["batch"], p["k_prime"]) for p in d["matrix"]}
    assert {("fp32", 1, 200), ("fp32", 8, 200), ("fp32", 32, 2561), ("f16x3", 1, 200), ("f16x3", 8, 200), ("f16x3", 32, 2561)} <= pts


def test_committed_round3_bench_line_carries_the_full_shard_legs():
    """profiles/r03_bench.json (the default `python bench.py` run at the end of round 3): the contract fields, and BASELINE configs 4 and 5
    timed at the size of one 8-way shard (12.5 M / 125 M items) next to the headline."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench.json")))
    check_line(d, expect_cpu_baseline=True)
    assert d["roofline"]["bound"] == "mfma" and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    legs = d["full_shards"]
    assert [l["variant"][:8] for l in legs] == ["fp32", "f16x3", "f16-exac", "two-pass"] and not any("skipped" in l for l in legs)
    assert all("N=12500000" in l["workload"] for l in legs[:3]) and "N=125000000" in legs[3]["workload"]
    for l in legs:
        assert l["queries_per_s"] > 0 and abs(l["queries_per_s"] - l["batch"] / (l["ms_per_step"] * 1e-3)) < 1e-6 * l["queries_per_s"]
    assert legs[2]["dense_fp32_fallbacks"] == 0 and 0 < legs[3]["hbm_frac_lower_bound"] < 1 and 0 < legs[0]["mfma_frac_lower_bound"] < 1


def test_committed_round4_bench_line_carries_the_quality_half():
    """profiles/r04_bench.json (the default `python bench.py` run at the end of round 4): BASELINE.json's metric is "queries/sec + HR@10/50
    parity" -- the line carries both halves, the reference's own CSV row, and recall@k vs exact for the config-5 shard."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    check_line(d, expect_cpu_baseline=True)
    hp = d["hr_parity"]
    assert hp["parity"] is True and hp["identical_rows"] + hp["rows_differing_only_inside_oracle_ties"] == hp["rows"] == 32
    for key in ("hr@1", "hr@5", "hr@10", "hr@50", "hr@100", "ndcg@10", "mrr"):
        assert hp[key]["rows_differing"] == 0 and hp[key]["hip"] == hp[key]["oracle"]
    csv = d["reference_csv"]
    assert csv["header"] == "HR@1,HR@5,HR@10,HR@50,HR@100,BatchTimeMsAvg,BatchTimeMsDev" and csv["row"].split(",")[:5] == csv["oracle_row"].split(",")
    two_pass = d["full_shards"][3]
    assert "N=125000000" in two_pass["workload"] and 0.3 < two_pass["recall"]["recall@10"] <= 1.0 and 0.2 < two_pass["recall"]["recall@120"] <= 1.0
    for mode in ("f16-exact", "f16x3-exact"):
        e = d["exact_fast_path"][mode]
        assert e["output_identical_to_fp32_path"] is True and e["eps_rigorous"] > e["eps"]     # (round 4 ran the verdicts on the monitored eps; round 5 runs them on the a-priori one)
    assert two_pass["pipelined"]["output_equal_to_unpipelined"] is True and two_pass["pipelined"]["ms_per_step"] <= two_pass["ms_per_step"] * 1.02
    tp = json.load(open(os.path.join(ROOT, "profiles", "r04_two_pass_125m.json")))
    assert tp["roofline"]["bound"] == "hbm" and 0 < tp["roofline"]["frac"] < 1 and 0.3 < tp["recall"]["recall@10"] <= 1.0
    assert tp["pipelined"]["output_equal_to_unpipelined"] is True and tp["ms_per_step"] < 1.65     # config-5 shard: 1.73 ms before the round-4 rebuild


def test_committed_round5_bench_line_is_the_proved_exact_path():
    """profiles/r05_bench.json (the default `python bench.py` run at the end of round 5): `value` is the proved exact path -- every timed
    call proved on the device with the a-priori bound, the output bit-identical to the dense fp32 kernels' -- with the dense fp32 number
    beside it, the roofline of the first-pass kernel against 2 500 / 3 TFLOP/s, HR parity on the whole corpus, and config 4's shard proved
    through the per-pair upper bounds."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    check_line(d, expect_cpu_baseline=True)
    p = d["proved"]
    assert p["is_headline"] is True and p["proved_calls"] == p["timed_calls"] == d["steps"] and p["dense_fp32_fallbacks"] == 0 and p["bound_violations"] == 0
    assert p["output_identical_to_fp32_path"] is True and abs(p["value"] - d["value"]) < 1e-9 and d["dtype"] == "f32" and d["value"] >= 10500      # (10 944 / 11 139 / 11 493 on the round's three boxes: the f16 kernel follows the box's power limit, the fp32 kernels do not)
    assert d["config"]["exact_path"].startswith("proved") and d["fp32_dense"]["value"] < d["value"] and d["fp32_dense"]["roofline"]["frac"] > 0.8
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["peak"] - 2500 / 3) < 1e-6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] is not None
    assert abs(r["kernel_ms"] - p["first_pass_kernel_ms"]) < 1e-9 and r["kernel_ms"] < d["ms_per_step"]
    hp = d["hr_parity"]
    assert hp["parity"] is True and "695762" in hp["what"] and hp["identical_rows"] + hp["rows_differing_only_inside_oracle_ties"] == hp["rows"]
    assert "all 695762" in d["cpu_baseline"]["sample"]
    rows = {(m["precision"], m["batch"], m["k_prime"]): m for m in d["matrix"]}
    assert rows[("proved", 8, 200)]["ms_per_step"] < 0.6 * rows[("fp32", 8, 200)]["ms_per_step"]
    legs = {l["variant"][:8]: l for l in d["full_shards"]}
    c4 = legs["proved"]
    assert "N=12500000" in c4["workload"] and c4["runs_dense_fp32"] is False and c4["bound"] == "per-pair upper bound"
    assert c4["proved_calls"] == c4["rescore_calls"] > 0 and c4["dense_fp32_fallbacks"] == 0 and c4["bound_violations"] == 0
    assert c4["ms_per_step"] < 0.45 * legs["fp32"]["ms_per_step"]


@pytest.mark.gpu
def test_live_bench_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-fast-path", "--no-matrix",
                          "--no-other-workloads", "--no-weights-sweep"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    check_line(d, expect_cpu_baseline=False)
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1
    # round 5: `value` is the PROVED exact path (the module's default) -- every timed call proved on the device with the a-priori bound, output identical
    # to the dense fp32 kernels', which are timed beside it; the roofline is the split-f16 first pass against a third of the f16 MFMA peak
    pr = d["proved"]
    assert pr["is_headline"] is True and pr["output_identical_to_fp32_path"] is True and pr["proved_calls"] == pr["timed_calls"] == 3
    assert pr["dense_fp32_fallbacks"] == 0 and pr["bound_violations"] == 0 and 0 < pr["eps_a_priori"] < 2.0
    assert d["value"] == pr["value"] and d["fp32_dense"]["value"] < d["value"] and d["config"]["prefilter"] == "f16x3, a-priori eps"
    assert abs(d["roofline"]["peak"] - 2500 / 3) < 1e-6 and "F16Unit" in d["roofline"]["kernel"] and d["fp32_dense"]["roofline"]["peak"] < 200
    # the quality half of BASELINE.json's metric ("+ HR@10/50 parity") rides on the line: HIP path vs the CPU oracle chain, row by row
    hp = d["hr_parity"]
    assert hp["parity"] is True and hp["identical_rows"] + hp["rows_differing_only_inside_oracle_ties"] == hp["rows"] == d["config"]["global_batch"]
    assert hp["identical_rows"] >= hp["rows"] - 2
    for key in ("hr@1", "hr@5", "hr@10", "hr@50", "hr@100", "ndcg@10", "mrr"):
        assert hp[key]["rows_differing"] == 0 and hp[key]["hip"] == hp[key]["oracle"]
    assert 0.0 < hp["hr@50"]["hip"] < hp["hr@100"]["hip"] < 1.0      # planted targets: neither trivially 0 nor 1
    csv = d["reference_csv"]                                           # eval_from_checkpoint.py:507-515
    assert csv["header"] == "HR@1,HR@5,HR@10,HR@50,HR@100,BatchTimeMsAvg,BatchTimeMsDev" and len(csv["row"].split(",")) == 7
