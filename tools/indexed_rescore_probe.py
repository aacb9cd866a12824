import sys, os, time, torch
sys.path.insert(0, "/root/repo")
import bench, rails_amd
from oracle import mol_oracle as O
cfg_key, N, _ = bench.WORKLOADS["amzn-books"]; cfg = O.CONFIGS[cfg_key]; dev = torch.device("cuda:0")
mol, _ = rails_amd.create_mol_interaction_module(cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim, cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim,
    cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False, query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=None)
mol.load_state_dict(O.synthetic_weights(cfg, seed=0), strict=True); mol = mol.to(dev).eval(); mol.precision = "f16-exact"
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
with torch.inference_mode():
  for B in (32, 8, 128, 1):
    q = O.synthetic_queries(cfg, B).to(dev)
    for mx in (1 << 30, 1024, 1 << 30, 1024):
        for k in (200, 2561):
            tk = rails_amd.MoLBruteForceTopK(mol, X, ids); tk.INDEXED_MAX_CANDIDATES = mx
            for _ in range(5): out = tk(q, k=k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): out = tk(q, k=k)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
            print(f"B={B} INDEXED_MAX={mx} k'={k}: {ms:.3f} ms/step", tk.stats())
