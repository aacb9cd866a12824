#!/usr/bin/env python3
"""Tail of |first pass - fp32| over >= 1e9 (query, item) pairs, for the reference's initialisers and for the xavier_normal_ re-init the
reference's sequential models apply to every >= 2-D parameter (reference modeling/sequential/hstu.py:632-650): what the eps of the
verified modes ("f16-exact", "f16x3-exact", rails_amd/topk_modules.py) has to cover.
  python tools/first_pass_error_tail.py --pairs 1.0e9 --out profiles/r03_first_pass_error_tail.json
"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402  (input generator only)


def build(cfg, w, dev, precision):
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(w, strict=True)
    mol = mol.to(dev).eval()
    mol.precision = precision
    return mol


def xavier_normal_reinit(w, seed):
    """hstu.py:632-650: xavier_normal_ on every parameter it applies to (>= 2-D); the rest keep their values."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in w.items():
        if v.dim() >= 2:
            fan_out, fan_in = v.shape[0], v.shape[1]
            if v.dim() > 2:
                rf = int(torch.tensor(v.shape[2:]).prod())
                fan_out, fan_in = fan_out * rf, fan_in * rf
            std = math.sqrt(2.0 / (fan_in + fan_out))
            out[k] = torch.randn(v.shape, generator=g) * std
        else:
            out[k] = v.clone()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="amzn-books")
    ap.add_argument("--pairs", type=float, default=1.0e9)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    cfg_key, N, _ = bench.WORKLOADS[args.workload]
    cfg = O.CONFIGS[cfg_key]
    dev = torch.device("cuda:0")
    X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).to(dev)
    n_batches = int(math.ceil(args.pairs / (args.batch * N)))
    NB, HI = 4000, 0.4   # histogram of |error| on [0, HI)
    results = {"workload": args.workload, "items": N, "batch": args.batch, "batches": n_batches, "pairs": n_batches * args.batch * N, "runs": []}
    for init in ("reference-init", "xavier_normal-reinit"):
        w = O.synthetic_weights(cfg, seed=0)
        if init != "reference-init":
            w = xavier_normal_reinit(w, seed=1)
        engines = {p: build(cfg, w, dev, None if p == "fp32" else p).engine() for p in ("fp32", "f16x1", "f16x3")}
        idx = {p: e.build_index(X) for p, e in engines.items()}
        stat = {p: {"max": 0.0, "hist": torch.zeros(NB, dtype=torch.float64, device=dev), "over": 0, "sum": 0.0, "min_signed": 0.0} for p in ("f16x1", "f16x3")}
        with torch.inference_mode():
            for b in range(n_batches):
                q = O.synthetic_queries(cfg, args.batch, seed=1000 + b).to(dev)
                uid = torch.arange(args.batch, dtype=torch.int64, device=dev) if cfg.uid_embedding_hash_sizes else None
                ref = engines["fp32"].score_dense(engines["fp32"].query_pack(q, uid)[0], args.batch, idx["fp32"])
                for p in ("f16x1", "f16x3"):
                    got = engines[p].score_dense(engines[p].query_pack(q, uid)[0], args.batch, idx[p])
                    d = got - ref
                    e = d.abs()
                    st = stat[p]
                    st["max"] = max(st["max"], float(e.max()))
                    st["min_signed"] = min(st["min_signed"], float(d.min()))   # the most UNDER-estimated pair (what can drop a true top-k item)
                    st["sum"] += float(e.double().sum())
                    st["over"] += int((e >= HI).sum())
                    st["hist"] += torch.histc(e.float(), bins=NB, min=0.0, max=HI).double()
        for p, st in stat.items():
            cum = torch.cumsum(st["hist"], 0)
            total = float(cum[-1]) + st["over"]

            def quantile(qq):
                target = qq * total
                i = int(torch.searchsorted(cum, torch.tensor(target, dtype=torch.float64, device=dev)))
                return HI if i >= NB else (i + 1) * HI / NB   # upper edge of the bin

            results["runs"].append({
                "init": init, "first_pass": p, "pairs": int(total), "max_abs_err": st["max"], "most_underestimated": st["min_signed"],
                "mean_abs_err": st["sum"] / total, "p99": quantile(0.99), "p99.99": quantile(0.9999), "p99.9999": quantile(0.999999),
                "pairs_at_or_above_0.4": st["over"],
                "default_eps": (7.5e-3 if p == "f16x1" else 5e-5) / cfg.temperature,
            })
            print(json.dumps(results["runs"][-1]))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
