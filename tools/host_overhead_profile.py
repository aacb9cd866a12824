import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, "/root/repo")
import rails_amd
from oracle import mol_oracle as O
dev = torch.device("cuda", 0)
cfg = O.CONFIGS["ml-1m"]; N = 3883; B = 32
w = O.synthetic_weights(cfg, seed=0)
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
mol.load_state_dict(w, strict=True); mol = mol.to(dev).eval()
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, B).to(dev)
kw = {"user_ids": torch.randint(0, 6040, (B,), dtype=torch.int64).to(dev)}
inv = torch.zeros((B, 211), dtype=torch.int64, device=dev)
with torch.inference_mode():
    tk = rails_amd.MoLBruteForceTopK(mol, X, ids)
    cand = rails_amd.CandidateIndex(ids=ids, embeddings=X)
    for _ in range(20): cand.get_top_k_outputs(q, 120, kw, tk, inv, truncate_k_prime_to=200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): cand.get_top_k_outputs(q, 120, kw, tk, inv, truncate_k_prime_to=200)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host issue {1e6*(t1-t0)/2000:.1f} us/step, with drain {1e6*(t2-t0)/2000:.1f} us/step")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): cand.get_top_k_outputs(q, 120, kw, tk, inv, truncate_k_prime_to=200)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
