#!/usr/bin/env python3
"""Is the small-corpus step host-bound?  Times CandidateIndex.get_top_k_outputs (plain fp32 path) eagerly and as a captured hipGraph
(torch.cuda.CUDAGraph around the same ctypes launches) for ML-1M, ML-20M and amzn-books at B = 1 / 32, and checks that the replay
returns the same bits.
  python tools/graph_step_probe.py [--steps 200]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, rails_amd
from oracle import mol_oracle as O   # configurations + synthetic inputs only

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
rows = []
for name, B, precision in (("ml-1m", 32, "fp32"), ("ml-20m", 32, "fp32"), ("amzn-books", 1, "fp32"), ("amzn-books", 32, "fp32"),
                           ("ml-1m", 32, "f16x3"), ("amzn-books", 1, "f16x3")):
    cfg_key, N, width = bench.WORKLOADS[name]
    cfg = O.CONFIGS[cfg_key]
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(O.synthetic_weights(cfg, seed=0), strict=True)
    mol = mol.to(dev).eval()
    mol.precision = None if precision == "fp32" else precision
    X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
    ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
    q = O.synthetic_queries(cfg, B).to(dev)
    kw = {}
    if len(cfg.uid_embedding_hash_sizes) > 0:
        g = torch.Generator().manual_seed(3)
        kw["user_ids"] = torch.randint(0, cfg.uid_embedding_hash_sizes[0], (B,), generator=g, dtype=torch.int64).to(dev)
    k, kp = 120, 200
    with torch.inference_mode():
        tk = rails_amd.MoLBruteForceTopK(mol, X, ids)
        cand = rails_amd.CandidateIndex(ids=ids, embeddings=X)
        g = torch.Generator().manual_seed(4)
        inv = torch.randint(1, N + 1, (B, max(width, 1)), generator=g, dtype=torch.int64).to(dev)
        step = lambda: cand.get_top_k_outputs(q, k, kw, tk, inv, truncate_k_prime_to=kp)   # noqa: E731
        for _ in range(5):
            ref_i, ref_s, _ = step()
        torch.cuda.synchronize()

        def timed(fn):
            best = float("inf")
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(a.steps):
                    fn()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / a.steps)
            return best * 1e6

        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        host_us = (time.perf_counter() - t0) / a.steps * 1e6   # enqueue time only (no sync inside the loop)
        torch.cuda.synchronize()
        eager_us = timed(step)
        row = {"workload": name, "B": B, "precision": precision, "eager_us": eager_us, "host_enqueue_us": host_us}
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gi, gs, _ = step()
            graph.replay()
            torch.cuda.synchronize()
            row["graph_identical"] = bool(torch.equal(gi, ref_i) and torch.equal(gs, ref_s))
            row["graph_us"] = timed(graph.replay)
        except Exception as e:   # noqa: BLE001
            row["graph_error"] = repr(e)[:300]
        rows.append(row)
        print(json.dumps(row), flush=True)
print(json.dumps(rows))
