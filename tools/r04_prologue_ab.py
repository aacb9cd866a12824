#!/usr/bin/env python3
"""Query prologue A/B (RAILS_PROLOGUE = 1 per-query kernel / 2 batched MFMA kernels / 3 split kernels), interleaved rounds in one
process: microseconds per call of eng.query_pack for the three real-dataset shapes at B = 1 / 8 / 32 / 128."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from oracle import mol_oracle as O

dev = torch.device("cuda", 0)
for name in ("ml-1m", "ml-20m", "amzn-books"):
    cfg = O.CONFIGS[name]
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(O.synthetic_weights(cfg, seed=0), strict=True)
    mol = mol.to(dev).eval()
    for B in (1, 8, 32, 128):
        q = O.synthetic_queries(cfg, B).to(dev)
        uid = torch.arange(B, dtype=torch.int64, device=dev) if cfg.uid_embedding_hash_sizes else None
        with torch.inference_mode():
            eng = mol.engine()
            n_q = eng.lib.rails_mol_query_pack_floats(rails_amd.engine.C.byref(eng.shape), B)
            buf = torch.empty(n_q, dtype=torch.float32, device=dev)
            times = {m: [] for m in ("1", "2", "3")}
            for rnd in range(9):
                for m in times:
                    os.environ["RAILS_PROLOGUE"] = m
                    eng.query_pack(q, uid, out=buf)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(50):
                        eng.query_pack(q, uid, out=buf)
                    e1.record()
                    torch.cuda.synchronize()
                    times[m].append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"{name:10s} B={B:3d}  per-query {statistics.median(times['1']):6.1f} us   batched {statistics.median(times['2']):6.1f} us   split {statistics.median(times['3']):6.1f} us")
os.environ.pop("RAILS_PROLOGUE", None)
