"""Why is the proved step slow at B = 8 inside bench.py's matrix?  Replays the leg with per-call wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, rails_amd
from oracle import mol_oracle as O
cfg = O.CONFIGS["amzn-books"]; N = 695762; dev = torch.device("cuda:0")
w = O.synthetic_weights(cfg, seed=0)
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups, cfg.item_dot_product_groups,
    cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim, cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim,
    cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False, query_nonlinearity=cfg.query_nonlinearity)
mol.load_state_dict(w, strict=True); mol = mol.to(dev).eval()
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, 32).to(dev)
inv = ids[0, torch.randint(0, N, (32, 61), device=dev)]
cand = rails_amd.CandidateIndex(ids=ids, embeddings=X)
import cProfile, pstats
with torch.inference_mode():
    for order in ((1, 8, 32), (8,), (32, 8)):
        tk = bench.brute_force_module(mol, X, ids, "proved")
        for B in order:
            qx, invx = q[:B], inv[:B]
            for _ in range(2):
                cand.get_top_k_outputs(qx, 120, {}, tk, invx, truncate_k_prime_to=200)
            torch.cuda.synchronize()
            walls = []
            for i in range(8):
                t0 = time.perf_counter()
                cand.get_top_k_outputs(qx, 120, {}, tk, invx, truncate_k_prime_to=200)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                walls.append((round((t1 - t0) * 1e3, 3), round((t2 - t0) * 1e3, 3)))
            print("order", order, "B", B, "(host enqueue ms, total ms) per call:", walls, "pad", tk._pad_scale, "kc", tk.rescore_stats.get("kc"))
            if B == 8:
                pr = cProfile.Profile(); pr.enable()
                for i in range(5):
                    cand.get_top_k_outputs(qx, 120, {}, tk, invx, truncate_k_prime_to=200)
                torch.cuda.synchronize(); pr.disable()
                pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
        del tk
