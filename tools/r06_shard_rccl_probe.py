#!/usr/bin/env python3
"""Per-rank step of the item-sharded proved path through the REAL module (ShardedMoLBruteForceTopK: exchange stream, events, RCCL all-gather, merge +
global verdict, the host's look at the verdict) on ONE GPU: rank 0's shard of an R-way split of amzn-books in a process group of ONE rank over
backend nccl (ShardedTopK.EXCHANGE_WITH_ONE_RANK), with the per-rank candidate count of the R-way run.  tools/shard_step_profile.py is the
hand-rolled single-stream emulation of the same step; the difference is what the module's streams and RCCL's launch add.
  python tools/r06_shard_rccl_probe.py [--world 8] [--steps 200] [--pipeline]"""
import argparse, os, socket, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
import rails_amd.sharded as S
from oracle import mol_oracle as O   # input generator only

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--pipeline", action="store_true")
ap.add_argument("--depth", type=int, default=1, help="batches submitted ahead of the one whose result is taken (--pipeline)")
ap.add_argument("--exchange-stream", action="store_true", help="ShardedTopK.EXCHANGE_STREAM: the exchange on the module's second stream")
ap.add_argument("--kc", type=int, default=0, help="candidates per rank (default: the R-way run's; one rank alone proves only with the single-device count, 1024)")
ap.add_argument("--cprofile", action="store_true", help="cProfile of the timed loop (host side), top functions by own time")
ap.add_argument("--host-times", action="store_true", help="host time per step inside submit / the all-gather call / the merge call / the wait for the verdict")
a = ap.parse_args()
with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
S.ShardedTopK.EXCHANGE_WITH_ONE_RANK = True
S.ShardedTopK.EXCHANGE_STREAM = a.exchange_stream
cfg = O.CONFIGS["amzn-books"]
N, B, k, kp, width = 695762, 32, 120, 200, 61
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
mol.load_state_dict(O.synthetic_weights(cfg, seed=0), strict=True)
mol = mol.to(dev).eval()
lo, hi = S.shard_bounds(N, a.world, 0)
X = torch.from_numpy(O.hash_item_table(1, lo, hi - lo, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(lo + 1, hi + 1, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, B).to(dev)
inv = torch.zeros((B, width), dtype=torch.int64, device=dev)
rails_amd.MoLBruteForceTopK.SPECULATE_MIN_ITEMS = 0
with torch.inference_mode():
    sh = S.ShardedMoLBruteForceTopK(mol, X, ids, hi - lo)
    total = kp + max(824, 3 * kp)
    per = -(-total // a.world)
    kc = a.kc or min((per + int(4 * per ** 0.5) + 32 + 31) // 32 * 32, hi - lo)
    sh._kc_local = lambda k_: kc            # the R-way run's candidates per rank
    cand = rails_amd.CandidateIndex(ids=ids, embeddings=X)

    def run(n):
        if not a.pipeline:
            for _ in range(n):
                cand.get_top_k_outputs(q, k, {}, sh, inv, truncate_k_prime_to=kp)
            return
        hs = [sh.submit(q, kp) for _ in range(min(a.depth, n))]
        sub = len(hs)
        for i in range(n):
            if sub < n:
                hs.append(sh.submit(q, kp))
                sub += 1
            sh.result(hs.pop(0), seen=(inv, k))

    acc = {}
    if a.host_times:
        def timed(obj, name, tag):
            fn = getattr(obj, name)
            def w(*x, **y):
                t = time.perf_counter()
                r = fn(*x, **y)
                acc[tag] = acc.get(tag, 0.0) + time.perf_counter() - t
                return r
            setattr(obj, name, w)
        from rails_amd import engine as E_
        timed(sh, "submit", "submit"); timed(sh, "_all_gather_rows", "all_gather call"); timed(E_, "merge_candidates_verdict", "merge call")
        timed(sh, "_gp_wait_verdict", "verdict wait"); timed(sh, "result", "result (incl. the three above)")
    run(8)
    acc.clear()
    torch.cuda.synchronize()
    if a.cprofile:
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if a.cprofile:
        pr.disable()
        ps = pstats.Stats(pr)
        ps.sort_stats("tottime").print_stats(28)
        ps.sort_stats("cumulative").print_stats(30)
    st = sh.stats()
print(f"real module, one-rank nccl group, shard of {a.world}: {hi - lo} items, kc {kc}, pipeline={a.pipeline} depth={a.depth} exchange_stream={a.exchange_stream}: {dt * 1e3:.3f} ms/step  "
      f"calls {st['calls']} proved {st.get('proved_calls')} fallbacks {st['fallbacks']} global_proof {st.get('global_proof')} {sh.exchange_info()}")
if acc:
    print("host us per step:", {k_: round(v / a.steps * 1e6, 1) for k_, v in acc.items()})
dist.destroy_process_group()
