#!/usr/bin/env python3
"""Step of the proved mode with the pair-gate weights scaled (the per-pair form of the bound on a large corpus): python tools/r06_scale_probe.py <scale> [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from oracle import mol_oracle as O
sc = float(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if len(sys.argv) > 3:
    rails_amd.MoLBruteForceTopK.PROVED_MAX_EPS_PER_PAIR = float(sys.argv[3])      # policy probe: the per-pair form up to this eps
dev = torch.device("cuda:0")
cfg = O.CONFIGS["amzn-books"]; N, B, k, kp = 695762, 32, 120, 200
w = O.synthetic_weights(cfg, seed=0)
p = "_gating_fn._qi_partial_module."
w[p + "1.weight"] = w[p + "1.weight"] * sc; w[p + "3.weight"] = w[p + "3.weight"] * sc
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity)
mol.load_state_dict(w, strict=True); mol = mol.to(dev).eval()
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, B).to(dev)
inv = ids[0, torch.randint(0, N, (B, 61), device=dev)]
with torch.inference_mode():
    tk = rails_amd.MoLBruteForceTopK(mol, X, ids)
    cand = rails_amd.CandidateIndex(ids=ids, embeddings=X)
    for _ in range(5): cand.get_top_k_outputs(q, k, {}, tk, inv, truncate_k_prime_to=kp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): cand.get_top_k_outputs(q, k, {}, tk, inv, truncate_k_prime_to=kp)
    torch.cuda.synchronize()
    print(f"scale {sc}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step", {kk: vv for kk, vv in tk.stats().items() if kk in ("calls", "proved_calls", "fallbacks", "kc", "eps", "bound_kind", "eps_rigorous", "bound_violations")})
