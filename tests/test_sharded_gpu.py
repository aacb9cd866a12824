"""GPU, two processes: the item-sharded path through the REAL HIP modules and a real process group.

  >= 2 visible devices: one rank per GPU, backend nccl (= RCCL), the all-gather runs on device tensors over xGMI;
  1 visible device (the gpurun box): both ranks share GPU 0, backend gloo, the 102 KB message is staged through
    the host by rails_amd/sharded.py -- everything else (local scoring, local top-k, pack, merge) is the same HIP code.

Oracle: the single-process module over the whole corpus, which the sharded result must equal bit for bit
(rails_amd/sharded.py docstring).  Also runs `python bench.py --gpus 2` the way the driver does (plain python, no launcher).
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, sizes, precision, ret, workload: str = "amzn-books", one_rank_exchange: bool = False):
    """one spawn, every corpus size of `sizes` (an int or a tuple of ints) in turn: the processes' start-up (import, process group) is most of a case's time"""
    import rails_amd
    import rails_amd.sharded

    rails_amd.sharded.ShardedTopK.EXCHANGE_WITH_ONE_RANK = one_rank_exchange
    torch.set_num_threads(8)     # (mp.spawn does not set OMP_NUM_THREADS as torchrun does: two ranks with one OpenMP thread per logical CPU each spend their time spinning)
    from oracle import mol_oracle as O
    from rails_amd import engine as E
    from rails_amd.sharded import ShardedMoLAvgTopK, ShardedMoLBruteForceTopK, shard_bounds
    from tests.test_gpu_parity import build_module

    rails_amd.MoLBruteForceTopK.SPECULATE_MIN_ITEMS = 0     # the shards are small: keep the verified modes on their speculative route
    rails_amd.MoLBruteForceTopK.PROVED_MIN_PAIRS = 0        # (and the proved flow, which the default policy starts at 2^18 pairs per call)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    multi = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if multi else 0)
    torch.cuda.set_device(dev)
    if multi:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for n_items in ((sizes,) if isinstance(sizes, int) else tuple(sizes)):
            cfg = O.CONFIGS[workload]
            mol = build_module(cfg, O.synthetic_weights(cfg, seed=1), dev)
            mol.precision = precision
            B, k, avg_k = 9, 200, 300
            q = O.synthetic_queries(cfg, B, seed=5).to(dev)
            X = torch.from_numpy(O.hash_item_table(7, 0, n_items, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
            ids = (torch.arange(n_items, dtype=torch.int64, device=dev) * 3 + 1).unsqueeze(0)
            lo, hi = shard_bounds(n_items, world, rank)
            with torch.inference_mode():
                # exact: sharded == single-device brute force, bit for bit, on every rank
                sh = ShardedMoLBruteForceTopK(mol, X[:, lo:hi], ids[:, lo:hi], n_items)
                s, i = sh(q, k=k)
                s2, i2 = sh(q, k=k)   # second call: recycled buffers
                full_s, full_i = rails_amd.MoLBruteForceTopK(mol, X, ids)(q, k=k)
                assert torch.equal(s, s2) and torch.equal(i, i2)
                assert torch.equal(s, full_s) and torch.equal(i, full_i), "sharded exact top-k differs from the single-device result"
                # pipelined: batch 2 is submitted before batch 1's exchange is taken -- bit-equal to the plain calls
                q2 = O.synthetic_queries(cfg, B, seed=6).to(dev)
                h1 = sh.submit(q, k)
                h2 = sh.submit(q2, k)
                p1 = sh.result(h1)
                p2 = sh.result(h2)
                r2 = sh(q2, k=k)
                assert torch.equal(p1[0], s) and torch.equal(p1[1], i) and torch.equal(p2[0], r2[0]) and torch.equal(p2[1], r2[1]), "pipelined != unpipelined"
                # CandidateIndex route: the seen-id filter inside the merge launch == forward + filter_seen_ids
                kk = min(120, i.shape[1])
                inv = i[:, torch.randperm(i.shape[1], device=dev)[:61]] if i.shape[1] >= 61 else i[:, :1].repeat(1, 61)
                want_i, want_s = E.filter_seen_ids(i, s, inv, kk)
                got = sh.forward_filtered(q, min(k, n_items), inv, kk)
                assert got is not None and torch.equal(got[0], want_i) and torch.equal(got[1], want_s), "filtered merge != merge + filter"
                cand = rails_amd.CandidateIndex(ids=ids, embeddings=X)      # the harness object holds the whole id table; the module is sharded
                c_i, c_s, _ = cand.get_top_k_outputs(q, kk, {}, sh, inv, truncate_k_prime_to=min(k, n_items))
                assert torch.equal(c_i, want_i) and torch.equal(c_s, want_s)
                provable = bool(sh._global_proof(q))    # (16x16x64: one a-priori eps exceeds PROVED_MAX_EPS -- the first pass writes per-pair upper bounds there)
                assert provable or precision not in (None, "f16x3-exact"), "the global proof did not engage"
                if precision in (None, "f16x3-exact") and provable:
                    # the default exact path and its explicit form prove ONCE for all shards (ShardedMoLBruteForceTopK's global proof): every call
                    # above went through it, was proved or redone, and no observed error exceeded the a-priori bound; with an absurd bound every
                    # verdict fails on every rank alike and the dense fp32 redo -- second exchange, output picked by the flag -- returns the same bits
                    st = sh.stats()
                    assert st.get("global_proof") is True and st["calls"] >= 6 and st["proved_calls"] + st["fallbacks"] == st["calls"] and st["bound_violations"] == 0, st
                    # (16x16x64 at 20 k items: a fifth of the corpus can reach the k-th score under the per-pair bounds -- calls are redone until the margin has grown)
                    assert n_items < 1000 or st["proved_calls"] >= 1 or workload == "synthetic-16x16x64", st
                    before = st["fallbacks"]
                    sh._gp_eps = 1.0e9
                    fs, fi = sh(q, k=k)
                    assert torch.equal(fs, full_s) and torch.equal(fi, full_i), "global proof: the redo differs from the single-device result"
                    got = sh.forward_filtered(q, min(k, n_items), inv, kk)
                    assert torch.equal(got[0], want_i) and torch.equal(got[1], want_s)
                    after = sh.stats()["fallbacks"]
                    if workload == "synthetic-16x16x64":    # (the margin may have grown to the whole shard by now)
                        assert before <= after <= before + 2
                    else:
                        assert after == before + (2 if n_items >= 1000 else 0)     # (a shard that is all candidates leaves nothing outside: proved whatever the bound)
                    sh._gp_eps = sh._local_module._proved_eps()
                if precision in ("f16x3-exact", "f16-exact"):   # ... and both are the fp32 path's result, bit for bit
                    mol32 = build_module(cfg, O.synthetic_weights(cfg, seed=1), dev)
                    f32_s, f32_i = rails_amd.MoLBruteForceTopK(mol32, X, ids)(q, k=k)
                    assert torch.equal(s, f32_s) and torch.equal(i, f32_i), "f16x3-exact sharded top-k differs from the fp32 path"
                if precision not in (None, "f16x3"):      # the two-pass compositions below do not depend on the exact modes
                    ret[(rank, n_items)] = (dist.get_backend(), s.cpu(), i.cpu())
                    continue
                # two-pass: sharded == merge of the per-shard MoLAvgTopK results (exact MoL scores, shard-major ties)
                sa = ShardedMoLAvgTopK(mol, X[:, lo:hi], ids[:, lo:hi], n_items, avg_top_k=avg_k)
                a_s, a_i = sa(q, k=k)
                parts = []
                for r in range(world):
                    l2, h2 = shard_bounds(n_items, world, r)
                    parts.append(rails_amd.MoLAvgTopK(mol, X[:, l2:h2], ids[:, l2:h2], avg_top_k=min(avg_k, h2 - l2))(q, k=min(k, h2 - l2)))
                es, epos = E.topk(torch.cat([p[0] for p in parts], 1), k)
                assert torch.equal(a_s, es) and torch.equal(a_i, torch.gather(torch.cat([p[1] for p in parts], 1), 1, epos))
                # global K': coarse candidates exchanged first -> exactly the single-device MoLAvgTopK, bit for bit
                gk = min(avg_k, n_items)
                sg = ShardedMoLAvgTopK(mol, X[:, lo:hi], ids[:, lo:hi], n_items, avg_top_k=gk, global_k_prime=True)
                g_s, g_i = sg(q, k=min(k, gk))
                o_s, o_i = rails_amd.MoLAvgTopK(mol, X, ids, avg_top_k=gk)(q, k=min(k, gk))
                assert torch.equal(g_s, o_s) and torch.equal(g_i, o_i), "global-K' sharded two-pass differs from the single-device algorithm"
            ret[(rank, n_items)] = (dist.get_backend(), s.cpu(), i.cpu())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", [None, "f16x3", "f16-exact"])      # (None = the default proved mode: its explicit spelling "f16x3-exact" takes the same global-proof route)
def test_two_ranks_through_the_hip_modules(precision):
    world = 2
    sizes = (70_001, 331) if precision is None else (70_001,)   # second case: the last shard is shorter than k
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), sizes, precision, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {(r, n) for r in range(world) for n in sizes}
    for n in sizes:
        assert torch.equal(ret[(0, n)][1], ret[(1, n)][1]) and torch.equal(ret[(0, n)][2], ret[(1, n)][2])   # identical on every rank
        assert ret[(0, n)][0] == ("nccl" if torch.cuda.device_count() >= world else "gloo")


@pytest.mark.parametrize("workload,n", [("amzn-books", 70_001), ("synthetic-16x16x64", 20_003)])
def test_one_rank_over_rccl_runs_the_exchange_path(workload, n):
    """A one-GPU box cannot hold two RCCL ranks (one rank per device), so the two-rank cases above exchange over gloo there.  This case runs the
    SAME checks in a process group of ONE rank over backend nccl with ShardedTopK.EXCHANGE_WITH_ONE_RANK: RCCL's communicator, its all-gather on
    device tensors on the module's exchange stream, the merge / global-verdict launches behind it, the forced redo's second exchange and the
    global-K' two-pass exchange all execute on the part -- the stream semantics the multi-GPU run will meet."""
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(1, _free_port(), (n,), None, ret, workload, True), nprocs=1, join=True)
    assert set(ret.keys()) == {(0, n)} and ret[(0, n)][0] == "nccl"


def test_bench_one_rank_exchange_over_rccl():
    """bench.py's sharded path (RCCL all-gather, all-reduce(MAX) of the timings, barriers, the self-checks) in a process group of one rank over
    backend nccl: the code the driver's multi-GPU run executes, on this box's one GPU (RAILS_BENCH_TEST_ONE_RANK_EXCHANGE)."""
    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RAILS_BENCH_TEST_BACKEND"):
        env.pop(v, None)
    env["RAILS_BENCH_TEST_ONE_RANK_EXCHANGE"] = "1"
    env["MASTER_PORT"] = str(_free_port())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--no-other-workloads", "--items", "200000"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    sh = d["sharded"]
    assert sh["backend"] == "nccl" and sh["rccl_ranks"] == 1 and d["n_gpus"] == 1
    chk = sh["check"]
    assert chk["all_ranks_identical"] is True and chk["merged_equals_unsharded"] is True and chk["nothing_outside_beats_kth"] is True
    assert sh["pipelined"]["output_equal_to_unpipelined"] is True
    pr = d["proved"]
    assert pr["output_identical_to_fp32_path"] is True and pr["bound_violations"] == 0 and pr["timed_calls"] == 3
    assert "sharded_global_proof" in pr          # the ONE-collective proved route ran (its string says what travels)
    assert pr["pipelined"]["output_equal_to_unpipelined"] is True and pr["pipelined"]["timed_calls"] == 3


@pytest.mark.parametrize("precision", [None, "f16x3"])
def test_two_ranks_16x16x64(precision):
    """BASELINE config 4's shape (L = 256: the team kernel of mol_score_wsplit.h) through the same two-rank checks: sharded ==
    single device bit for bit, the verified mode == the fp32 path, pipelined == unpipelined, two-pass compositions."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), 20_003, precision, ret, "synthetic-16x16x64"), nprocs=world, join=True)
    assert set(ret.keys()) == {(0, 20_003), (1, 20_003)}
    assert torch.equal(ret[(0, 20_003)][1], ret[(1, 20_003)][1]) and torch.equal(ret[(0, 20_003)][2], ret[(1, 20_003)][2])


def _bench_two_ranks(*extra):
    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    if torch.cuda.device_count() < 2:
        env["RAILS_BENCH_TEST_BACKEND"] = "gloo"     # both ranks on GPU 0; the message is staged through the host
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--no-other-workloads", *extra], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3
    assert d["sharded"]["rccl_ranks"] == 2 and d["sharded"]["backend"] in ("nccl", "gloo")
    return d


def test_bench_self_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (how the driver may call it) must re-exec itself under
    torch.distributed.run and print one JSON line with n_gpus = 2 -- a line that carries its own correctness evidence (pytest does
    not run on the multi-GPU node): every rank ends with the same result, the merged top-k' equals a one-device recompute over the
    union of the shards' candidates, and no shard holds anything that beats the merged k'-th."""
    d = _bench_two_ranks("--items", "200000")
    assert d["scaling"] == "strong" and d["config"]["n_items"] == 200000 and "roofline" in d
    sh = d["sharded"]
    assert sh["pipelined"]["output_equal_to_unpipelined"] is True
    assert set(sh["phase_ms"]) == {"score", "select_and_pack", "all_gather", "merge_and_filter"}
    chk = sh["check"]
    assert chk["all_ranks_identical"] is True and chk["merged_equals_unsharded"] is True and chk["nothing_outside_beats_kth"] is True
    assert 200 <= chk["union_items"] <= 2 * 32 * 200
    # the proved exact path (the module's default) ran per shard: identical output on every rank; it is the headline only if every
    # rank proved every timed call (a shard's k'-th score sits in a denser part of the distribution: fallbacks are legitimate here)
    pr = d["proved"]
    assert pr["output_identical_to_fp32_path"] is True and pr["bound_violations"] == 0 and pr["timed_calls"] == 2 * 3
    assert pr["proved_calls"] + pr["dense_fp32_fallbacks"] >= pr["timed_calls"] - 2 * 3 and pr["is_headline"] == (pr["proved_calls"] == pr["timed_calls"] and pr["dense_fp32_fallbacks"] == 0)
    pp = pr["pipelined"]       # the same steps with a batch in flight: same output, every call accounted for
    assert pp["output_equal_to_unpipelined"] is True and pp["timed_calls"] == 2 * 3 and pp["proved_calls"] + pp["dense_fp32_fallbacks"] >= pp["timed_calls"] - 2 * 3


def test_bench_two_ranks_on_the_256_logit_shape():
    """BASELINE config 4's shape (16x16x64, the team kernel) through the same self-verifying sharded run."""
    d = _bench_two_ranks("--workload", "synthetic-16x16x64", "--items", "40000", "--no-fast-path")
    chk = d["sharded"]["check"]
    assert chk["all_ranks_identical"] is True and chk["merged_equals_unsharded"] is True and chk["nothing_outside_beats_kth"] is True


def test_bench_two_ranks_two_pass_reports_recall():
    """BASELINE config 5's mode (two-pass: coarse prefilter + MoL rerank) on two ranks: same result on every rank, recall@k against exact
    brute force over the whole sharded corpus on the line."""
    d = _bench_two_ranks("--workload", "synthetic-8x8x32", "--items", "600000", "--two-pass", "500")
    assert d["sharded"]["check"]["all_ranks_identical"] is True
    r = d["recall"]
    assert 0.0 < r["recall@10"] <= 1.0 and 0.0 < r[f"recall@{r['k']}"] <= 1.0 and r["avg_top_k_per_shard"] == 500
