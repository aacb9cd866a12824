"""Kernel time of the fused-selection scoring launch against the dense one (events around the launches; amzn-books, fp32).
RAILS_AMD_LIBRARY selects the build (timing experiments: librails_amd_sel{A,B,C}.so)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402
from rails_amd import engine as E  # noqa: E402

dev = torch.device("cuda:0")
cfg = O.CONFIGS["amzn-books"]
w = O.synthetic_weights(cfg, seed=0)
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups, cfg.item_dot_product_groups,
    cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim, cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim,
    cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False, query_nonlinearity=cfg.query_nonlinearity,
    uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
mol.load_state_dict(w, strict=True)
mol = mol.to(dev).eval()
mol.precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
N, B, k = 695762, 32, 200
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(N, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, B).to(dev)
with torch.inference_mode():
    tk = rails_amd.MoLBruteForceTopK(mol, X, ids)
    eng = tk._bind()
    qpack, _, _ = eng.query_pack(q, None)
    ws = eng._score_topk_workspace(B, dev)
    logits = torch.empty((B, N), dtype=torch.float32, device=dev)
    out_s = torch.empty((B, k), dtype=torch.float32, device=dev); out_i = torch.empty((B, k), dtype=torch.int64, device=dev)

    def fused():
        E._lib.check(eng.lib.rails_mol_score_survivors(C.byref(eng.dense_shape), E._ptr(eng.gate_pack), E._ptr(qpack), B, E._ptr(tk._index.buf), N, k, None, 0,
                                                       E._ptr(ws), ws.numel(), E._stream()), "survivors")

    def select():
        E._lib.check(eng.lib.rails_select_survivors(B, k, None, 0, None, 0, 0, E._ptr(out_s), E._ptr(out_i), E._ptr(ws), ws.numel(), E._stream()), "select")

    def dense():
        eng.score_dense(qpack, B, tk._index, out=logits)

    res = {}
    for name, fn, after in (("dense", dense, None), ("fused", fused, select), ("dense", dense, None), ("fused", fused, select)):
        for _ in range(3):
            fn()
            if after: after()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            if after: after()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res.setdefault(name, []).append(sum(ts) / len(ts))
    print(os.path.basename(os.environ.get("RAILS_AMD_LIBRARY", "librails_amd.so")), mol.precision, {k2: [round(x, 4) for x in v] for k2, v in res.items()})
