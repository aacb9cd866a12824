#!/usr/bin/env python3
"""First-pass kernel time of the proved step, step by step, after (a) 300 ms of idle, (b) 150 ms of dense fp32 steps with no gap, (c) right after
a run of proved steps: what the f16 kernel's time at the start of a timed region depends on (docs/HISTORY.md R5.4)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402  (input generator only)


def main():
    cfg_key, N, _ = bench.WORKLOADS["amzn-books"]
    cfg = O.CONFIGS[cfg_key]
    dev = torch.device("cuda:0")
    w = O.synthetic_weights(cfg, seed=0)
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(w, strict=True)
    mol = mol.to(dev).eval()
    X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
    ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
    q = O.synthetic_queries(cfg, 32).to(dev)
    n = 30
    with torch.inference_mode():
        proved = bench.brute_force_module(mol, X, ids, "proved")
        dense = bench.brute_force_module(mol, X, ids, "fp32")
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        cur = {"i": None}
        proved._first_pass_hook = lambda wh: (e0 if wh == 0 else e1)[cur["i"]].record() if cur["i"] is not None else None

        def run_proved(tag):
            for i in range(n):
                cur["i"] = i
                proved(q, k=200)
            cur["i"] = None
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in zip(e0, e1)]
            print(f"{tag:48s}", " ".join(f"{v:.3f}" for v in ms[:12]), "... last", f"{ms[-1]:.3f}", flush=True)

        for _ in range(3):
            proved(q, k=200)
            dense(q, k=200)
        torch.cuda.synchronize()
        for rep in range(2):
            time.sleep(0.3)
            run_proved("(a) after 300 ms idle")
            run_proved("(c) right after proved steps (one sync between)")
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.15:
                dense(q, k=200)
            run_proved("(b) right after 150 ms of dense fp32 steps")
            time.sleep(0.02)
            run_proved("(d) after 20 ms idle")
            time.sleep(0.002)
            run_proved("(e) after 2 ms idle")


if __name__ == "__main__":
    main()
