"""Survivor statistics of the fused score + select path (rails_mol_score_survivors): keys appended per query, per workgroup
segment, and the bounds at the end of the launch.  python tools/fused_select_stats.py [--workload amzn-books] [--items N] [--k 200]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402
from rails_amd import engine as E  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="amzn-books")
ap.add_argument("--items", type=int, default=695762)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--k", type=int, default=200)
ap.add_argument("--precision", default="fp32")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = O.CONFIGS[a.workload]
w = O.synthetic_weights(cfg, seed=0)
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups, cfg.item_dot_product_groups,
    cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim, cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim,
    cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False, query_nonlinearity=cfg.query_nonlinearity,
    uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
mol.load_state_dict(w, strict=True)
mol = mol.to(dev).eval()
mol.precision = a.precision
N, B, k = a.items, a.batch, a.k
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(N, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, B).to(dev)
kw = {"user_ids": torch.arange(B, dtype=torch.int64, device=dev)} if cfg.uid_embedding_hash_sizes else {}
with torch.inference_mode():
    tk = rails_amd.MoLBruteForceTopK(mol, X, ids)
    eng = tk._bind()
    print("supported:", eng.score_topk_supported(B, N, k))
    qpack, _, _ = eng.query_pack(q, kw.get("user_ids"))
    ws = eng._score_topk_workspace(B, dev)
    E._lib.check(eng.lib.rails_mol_score_survivors(C.byref(eng.dense_shape), E._ptr(eng.gate_pack), E._ptr(qpack), B, E._ptr(tk._index.buf), N, k, None, 0,
                                                   E._ptr(ws), ws.numel(), E._stream()), "survivors")
    torch.cuda.synchronize()
    hdr = (4 * B + 4 + 255) // 256 * 256
    thr = ws[: 4 * B].view(torch.int32)
    status = int(ws[4 * B: 4 * B + 4].view(torch.int32))
    if os.environ.get("RAILS_AMD_LIBRARY", "").endswith("selph.so"):   # debug build (RAILS_SEL_PHASES): workgroup 0's first checkpoint
        print("first checkpoint selection of workgroup 0:", int(ws[4 * B + 4: 4 * B + 8].view(torch.int32)) / 100.0, "us")
        ws[4 * B + 4: 4 * B + 8].zero_()
    lists = ws[hdr:].view(torch.int64).view(B, 256, 128)
    nz = (lists != 0)
    per_q = nz.sum(dim=(1, 2))
    per_seg = nz.sum(dim=2)
    print("status", status, "survivors per query: min/mean/max", int(per_q.min()), float(per_q.float().mean()), int(per_q.max()))
    print("per (query, workgroup) segment: max", int(per_seg.max()), "mean", float(per_seg.float().mean()))
    logits = eng.score_dense(qpack, B, tk._index)
    kth = logits.topk(k, dim=1).values[:, -1]
    keyf = thr.view(torch.uint32) if hasattr(torch, "uint32") else thr
    t = thr.to(torch.int64) & 0xFFFFFFFF
    f = torch.where(t >= 0x80000000, t & 0x7FFFFFFF, (~t) & 0xFFFFFFFF).to(torch.int32).view(torch.float32)
    above = (logits >= f[:, None]).sum(1)
    print("final bounds vs true k-th: bound <= kth everywhere:", bool((f <= kth).all()), "; items above the final bound: mean", float(above.float().mean()))
    # consume the lists so that the workspace is clean
    out_s = torch.empty((B, k), dtype=torch.float32, device=dev); out_i = torch.empty((B, k), dtype=torch.int64, device=dev)
    E._lib.check(eng.lib.rails_select_survivors(B, k, None, 0, None, 0, 0, E._ptr(out_s), E._ptr(out_i), E._ptr(ws), ws.numel(), E._stream()), "select")
    torch.cuda.synchronize()
    r = logits.topk(k, dim=1)
    print("selection equals torch.topk values:", bool(torch.equal(out_s, r.values)), "workspace clean:", int(ws.count_nonzero()) == 0)
