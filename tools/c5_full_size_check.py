#!/usr/bin/env python3
"""Full-size check of the fused coarse top-K' (BASELINE config 5, one 8-way shard): the fused scan against the materialising path
(coarse scores of all N items + exact top-K') on the SAME device table, bit for bit, for B = 32 and B = 128.
  python tools/c5_full_size_check.py [--items 125000000] [--k-prime 1000]
The table is made straight from a device-generated (N, D) embedding table through the item-side MoL projection and the coarse
build; the 160 GB fp32 index of the full step is not needed for this check and is released before the comparison."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, rails_amd
from rails_amd import engine as E
from oracle import mol_oracle as O   # configuration + synthetic weights / queries only

ap = argparse.ArgumentParser()
ap.add_argument("--items", type=int, default=125_000_000)
ap.add_argument("--k-prime", type=int, default=1000)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = O.CONFIGS[bench.WORKLOADS["synthetic-8x8x32"][0]]
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
mol.load_state_dict(O.synthetic_weights(cfg, seed=0), strict=True)
mol = mol.to(dev).eval()
N = a.items
X = torch.empty((1, N, cfg.item_embedding_dim), dtype=torch.float32, device=dev)
g = torch.Generator(device=dev).manual_seed(1000)
for s0 in range(0, N, 8_000_000):
    n0 = min(8_000_000, N - s0)
    X[0, s0 : s0 + n0] = torch.fmod(torch.randn((n0, cfg.item_embedding_dim), generator=g, device=dev), 2.0) * 0.02
ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
out = {"items": N, "k_prime": a.k_prime, "cases": []}
with torch.inference_mode():
    at = rails_amd.MoLAvgTopK(mol, X, ids, avg_top_k=a.k_prime)
    eng = at._bind()
    table = at._table()
    for B in (32, 128):
        q = O.synthetic_queries(cfg, B, seed=40 + B).to(dev)
        _, eq, _ = eng.query_pack(q, None, want_plain=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fs, fp, counts = eng.coarse_topk(eq, table, True, a.k_prime)
        torch.cuda.synchronize(); t_fused = time.perf_counter() - t0
        same_s = same_p = True
        t_mat = 0.0
        for b0 in range(0, B, 32):   # the materialised scores 32 queries at a time (16 GB per slice at 125 M items)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            coarse = eng.coarse_scores(eq[b0 : b0 + 32].contiguous(), table, True)
            rs, rp = E.topk(coarse, a.k_prime)
            torch.cuda.synchronize(); t_mat += time.perf_counter() - t0
            same_s &= bool(torch.equal(fs[b0 : b0 + 32], rs)); same_p &= bool(torch.equal(fp[b0 : b0 + 32], rp))
            del coarse
        out["cases"].append({"B": B, "scores_identical": same_s, "positions_identical": same_p, "candidates_min": int(counts.min()),
                             "candidates_max": int(counts.max()), "capacity": int(eng.coarse_topk_capacity(a.k_prime)),
                             "fused_ms_first_call": t_fused * 1e3, "materialised_ms": t_mat * 1e3})
print(json.dumps(out, indent=1))
sys.exit(0 if all(c["scores_identical"] and c["positions_identical"] for c in out["cases"]) else 1)
