#!/usr/bin/env python3
"""How much of "returned ids identical to the reference's modulo ties" rests on the tie rule: for the golden
fixtures with reference top-k ids (F2 at k = 10 / 200 / N for the four BASELINE shapes, F7 = ML-1M and ML-20M at full size), in both
result-producing precisions, count the rows / positions where the HIP path's id differs from the reference's, and how many of those
positions are NOT inside a run of reference scores closer than tol -- at the survey's tol = 1e-5 and at the 2e-5 the tests use.
  python tools/tie_branch_census.py > profiles/r04_tie_branch_census.json"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from tests._fixtures import Fixture, full_size_inputs, tie_branch_census
from tests.test_gpu_parity import SUPPORTED, build_module

dev = torch.device("cuda", 0)
rows = []
for precision in ("fp32", "f16x3"):
    for name in SUPPORTED:
        fx = Fixture(name)
        mol = build_module(fx.cfg, fx.weights, dev, precision)
        X, ids = fx.t("X").to(dev), fx.t("item_ids").to(dev)
        kw = {k: v.to(dev) for k, v in fx.kw.items()}
        with torch.inference_mode():
            tk = rails_amd.MoLBruteForceTopK(mol, X, ids)
            for k in (10, 200, X.shape[1]):
                _, i = tk(fx.t("q").to(dev), k=k, **kw)
                for tol in (1e-5, 2e-5):
                    rows.append({"fixture": f"F2 {name} k={k}", "precision": precision, **tie_branch_census(i, fx.t(f"F2/k{k}/scores"), fx.t(f"F2/k{k}/ids"), tol)})
    for name in ("full_c1_ml1m", "full_c2_ml20m"):
        fx = Fixture(name)
        mol = build_module(fx.cfg, fx.weights, dev, precision)
        X, ids = full_size_inputs(fx)
        kw = {k: v.to(dev) for k, v in fx.kw.items()}
        with torch.inference_mode():
            _, i = rails_amd.MoLBruteForceTopK(mol, X.to(dev), ids.to(dev))(fx.t("q").to(dev), k=200, **kw)
        for tol in (1e-5, 2e-5):
            rows.append({"fixture": f"F7 {name} k=200", "precision": precision, **tie_branch_census(i, fx.t("scores"), fx.t("ids"), tol)})
tot = {tol: {"positions": sum(r["rows"] * r["k"] for r in rows if r["tie_tol"] == tol), "differing": sum(r["positions_differing"] for r in rows if r["tie_tol"] == tol),
             "outside_tie_runs": sum(r["positions_outside_tie_runs"] for r in rows if r["tie_tol"] == tol)} for tol in (1e-5, 2e-5)}
print(json.dumps({"what": __doc__.split("\n  python")[0], "totals": {str(k): v for k, v in tot.items()}, "rows": rows}, indent=1))
