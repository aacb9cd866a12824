"""bench.main() with host timers around the engine calls: which call takes the one-off 20-70 ms inside a timed region?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, rails_amd
from rails_amd import engine as E, mol_module as MM
log = []
T0 = time.perf_counter()
def wrap(obj, name, thr=2e-3):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t0
        if dt > thr: log.append((round(t0 - T0, 3), name, round(dt * 1e3, 2)))
        return r
    setattr(obj, name, g)
for n in ("topk", "rescore_select", "rescore_verdict", "filter_seen_ids", "topk_filtered"):
    wrap(E, n)
for n in ("score_dense", "score_indexed", "query_pack_both", "query_pack", "build_index", "gather_index", "__init__"):
    wrap(E.MolEngine, n)
for n in ("_absorb_state", "_proved_eps", "_bound_from_weights", "_proved_applies", "_engine_for_bind", "_forward_rescored", "_forward_fp32_dense", "stats", "rigorous_eps", "_gi_abs_max"):
    wrap(rails_amd.MoLBruteForceTopK, n)
wrap(MM.MoLSimilarity, "engine")
wrap(rails_amd.CandidateIndex, "get_top_k_outputs", 4e-3)
sys.argv = ["bench.py", "--steps", "10", "--warmup", "3", "--no-other-workloads", "--no-cpu-baseline", "--no-hr-parity"]
try:
    bench.main()
finally:
    for rec in log:
        print(rec, file=sys.stderr)
