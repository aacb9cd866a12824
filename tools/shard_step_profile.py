#!/usr/bin/env python3
"""Per-rank step of the item-sharded path at world size R, emulated on ONE GPU: rank 0's shard of amzn-books, the real
prologue / scoring / local top-k / pack / merge / filter kernels, and the all-gather replaced by a device copy of the
rank's own message R times (so merge sees R*k' keys).  Everything but the RCCL latency is real: run it under
`rocprofv3 --kernel-trace --stats` to see the fixed per-step costs that bound strong scaling."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from rails_amd import engine as E
from rails_amd.sharded import shard_bounds
from oracle import mol_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--precision", default=None, help="fp32 (default) | f16x3 | f16x3-exact | f16-exact (the module's verified route, proved PER SHARD) | "
                                                 "proved-global (ShardedMoLBruteForceTopK's flow: one proof for all shards, its collectives replaced by copies)")
ap.add_argument("--pipeline", action="store_true", help="exchange (copy + merge + filter) of step i on a second stream, behind step i+1's scoring")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = O.CONFIGS["amzn-books"]
N, B, k, kp, width = 695762, 32, 120, 200, 61
w = O.synthetic_weights(cfg, seed=0)
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
mol.load_state_dict(w, strict=True)
mol = mol.to(dev).eval()
if a.precision and a.precision != "proved-global":
    mol.precision = a.precision
lo, hi = shard_bounds(N, a.world, 0)
X = torch.from_numpy(O.hash_item_table(1, lo, hi - lo, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(lo + 1, hi + 1, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, B).to(dev)
inv = torch.zeros((B, width), dtype=torch.int64, device=dev)
with torch.inference_mode():
    tk = rails_amd.MoLBruteForceTopK(mol, X, ids, exact_mode="dense")     # "fp32" = the dense fp32 kernels (the module's default is the proved mode)
    eng = tk._bind()
    logits = torch.empty((B, hi - lo), dtype=torch.float32, device=dev)

    tk.SPECULATE_MIN_ITEMS = 0
    side = torch.cuda.Stream(dev)

    if a.precision == "proved-global":
        del tk
        rails_amd.MoLBruteForceTopK.SPECULATE_MIN_ITEMS = 0
        tk = rails_amd.MoLBruteForceTopK(mol, X, ids, exact_mode="proved")
        assert tk.shard_can_speculate()
        eps, state = tk._proved_eps(), torch.zeros(8, dtype=torch.float32, device=dev)
        total = kp + max(824, 3 * kp)
        per = -(-total // a.world)
        kc = min((per + int(4 * per ** 0.5) + 32 + 31) // 32 * 32, hi - lo)
        print("global proof: eps", eps, "kc per rank", kc)

    host = torch.zeros(8, dtype=torch.float32).pin_memory()
    call_ws = torch.zeros(8 + 4 * B, dtype=torch.int32, device=dev)
    issued = [0]

    def global_step():
        # round 6: one message (top-k' | ids | m | err), one exchange, merge + verdict + filter in one launch, the verdict read from the pinned mirror
        msg, qp32 = tk.speculate_for_shard(q, kp, kc)
        sp = tk._engine.spec
        off = (B + 32 // sp.query_dot_product_groups - 1) // (32 // sp.query_dot_product_groups) * 32 * sp.dot_product_dimension
        gq = qp32[off : off + B * sp.num_logits]
        gathered = msg.repeat(a.world, 1)                                   # stands for the all-gather
        out = E.merge_candidates_verdict(gathered, a.world, kp, kp, eps, 1.0, gq, sp.num_logits, tk._gate_guard_limit, state, host, call_ws, (inv, k))
        issued[0] += 1
        if not a.pipeline:
            while float(host[5]) < issued[0]:
                pass
            assert int(host.view(torch.int32)[1]) == 0, "the emulated global verdict failed"
        return out

    def local_part():
        if a.precision == "proved-global":
            return None
        if a.precision and a.precision.endswith("-exact"):
            s, top = tk(q, k=min(kp, hi - lo))
        else:
            qpack, _, _ = eng.query_pack(q, None)
            eng.score_dense(qpack, B, tk._index, out=logits)
            s, top = E.topk(logits, min(kp, hi - lo), ids=tk._ids_flat)
        return E.pack_candidates(s, top, kp) if a.world > 1 else (s, top)

    def exchange(msg):
        if a.precision == "proved-global":
            return global_step()
        if a.world > 1:
            gathered = msg.repeat(a.world, 1)
            if E.merge_filter_fusable(kp, inv.shape[1], k):   # the sharded module's route: the seen-id filter inside the merge launch
                return E.merge_candidates_filtered(gathered, a.world, kp, kp, inv, k)
            s, top = E.merge_candidates(gathered, a.world, kp, kp)
        else:
            s, top = msg
        return E.filter_seen_ids(top, s, inv, k)

    def run(n):
        if not a.pipeline:
            for _ in range(n):
                exchange(local_part())
            return
        cur = torch.cuda.current_stream(dev)
        msg = local_part()
        for i in range(n):
            ev = torch.cuda.Event(); ev.record()
            nxt = local_part() if i + 1 < n else None
            side.wait_event(ev)
            with torch.cuda.stream(side):
                exchange(msg)
            msg = nxt
        cur.wait_stream(side)

    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
if a.precision == "proved-global":
    print("verdict state [seen err, redo, eps, err, gap, calls, redone, guard max]:", state.view(torch.float32).tolist(), "redo flag", int(state.view(torch.int32)[1]))
print(f"world={a.world} shard={hi - lo} items precision={a.precision or 'fp32'} pipeline={a.pipeline}: {dt * 1e3:.3f} ms/step (no RCCL latency) -> {B / dt:.0f} q/s if all ranks match")
