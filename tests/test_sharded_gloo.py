"""CPU, gloo, world_size 2: the item-sharded top-k orchestration (rails_amd/sharded.py) -- shard bounds,
the single all-gather message format, shard-major merge order -- with the oracle standing in for the HIP
local top-k / merge kernels (the GPU tests cover those).  Result must equal the unsharded oracle exactly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mol_oracle as O
from rails_amd.sharded import ShardedMoLAvgTopK, ShardedMoLBruteForceTopK, pack_candidates, shard_bounds, unpack_candidates


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_items: int, k: int, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cfg = O.CONFIGS["amzn-books"]
        w = O.synthetic_weights(cfg, seed=0)
        q = O.synthetic_queries(cfg, 5)
        lo, hi = shard_bounds(n_items, world, rank)
        X = torch.from_numpy(O.hash_item_table(1, lo, hi - lo, cfg.item_embedding_dim)).unsqueeze(0)
        ids = (torch.arange(lo, hi, dtype=torch.int64) * 3 + 1).unsqueeze(0)

        def local_topk(qq, kk, **kw):
            logits = O.mol_logits(cfg, w, qq, X)
            s, pos = O.select_topk_deterministic(logits, kk)
            return s, ids.reshape(-1)[pos]

        def merge(scores, all_ids, kk):
            s, pos = O.select_topk_deterministic(scores, kk)
            return s, torch.gather(all_ids, 1, pos)

        mod = ShardedMoLBruteForceTopK(None, None, ids, n_items, local_topk=local_topk, merge=merge)
        s, i = mod(q, k=k)
        # a second call exercises buffer reuse; results must be identical on every rank
        s2, i2 = mod(q, k=k)
        assert torch.equal(s, s2) and torch.equal(i, i2)
        ret[rank] = (s.clone(), i.clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items,k", [(1000, 50), (65, 40)])   # second case: k > items of the last shard
def test_sharded_topk_equals_unsharded_oracle(n_items, k):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_items, k, ret), nprocs=world, join=True)
    cfg = O.CONFIGS["amzn-books"]
    w = O.synthetic_weights(cfg, seed=0)
    q = O.synthetic_queries(cfg, 5)
    X = torch.from_numpy(O.hash_item_table(1, 0, n_items, cfg.item_embedding_dim)).unsqueeze(0)
    ids = torch.arange(0, n_items, dtype=torch.int64) * 3 + 1
    rs, rpos = O.select_topk_deterministic(O.mol_logits(cfg, w, q, X), k)
    for rank in range(world):
        s, i = ret[rank]
        assert torch.equal(s, rs) and torch.equal(i, ids[rpos])


def _avg_worker(rank: int, world: int, port: int, n_items: int, k: int, avg_k: int, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cfg = O.CONFIGS["amzn-books"]
        w = O.synthetic_weights(cfg, seed=0)
        q = O.synthetic_queries(cfg, 4)
        lo, hi = shard_bounds(n_items, world, rank)
        X = torch.from_numpy(O.hash_item_table(1, lo, hi - lo, cfg.item_embedding_dim)).unsqueeze(0)
        ids = (torch.arange(lo, hi, dtype=torch.int64) * 3 + 1).unsqueeze(0)

        def local_topk(qq, kk, **kw):   # the oracle's MoLAvgTopK on this shard
            s, i, _ = O.avg_topk(cfg, w, qq, X, ids, kk, min(avg_k, hi - lo))
            return s, i

        def merge(scores, all_ids, kk):
            s, pos = O.select_topk_deterministic(scores, kk)
            return s, torch.gather(all_ids, 1, pos)

        mod = ShardedMoLAvgTopK(None, None, ids, n_items, avg_top_k=avg_k, local_topk=local_topk, merge=merge)
        s, i = mod(q, k=k)
        with pytest.raises(ValueError, match="must be larger than k"):
            mod(q, k=avg_k + 1)
        ret[rank] = (s.clone(), i.clone(), *local_topk(q, min(k, hi - lo)))
    finally:
        dist.destroy_process_group()


def test_sharded_two_pass_merges_the_per_shard_reranks():
    """Config 5's shape: MoLAvgTopK per shard + the single all-gather + merge.  Every rank must hold the top-k (by exact MoL
    score, shard-major ties) of the union of the per-shard two-pass results."""
    world, n_items, k, avg_k = 2, 600, 20, 60
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_avg_worker, args=(world, _free_port(), n_items, k, avg_k, ret), nprocs=world, join=True)
    all_s = torch.cat([ret[r][2] for r in range(world)], dim=1)
    all_i = torch.cat([ret[r][3] for r in range(world)], dim=1)
    es, pos = O.select_topk_deterministic(all_s, k)
    for rank in range(world):
        s, i = ret[rank][0], ret[rank][1]
        assert torch.equal(s, es) and torch.equal(i, torch.gather(all_i, 1, pos))


def _global_worker(rank: int, world: int, port: int, n_items: int, k: int, avg_k: int, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cfg = O.CONFIGS["amzn-books"]
        w = O.synthetic_weights(cfg, seed=0)
        q = O.synthetic_queries(cfg, 4)
        lo, hi = shard_bounds(n_items, world, rank)
        X = torch.from_numpy(O.hash_item_table(1, lo, hi - lo, cfg.item_embedding_dim)).unsqueeze(0)
        ids = (torch.arange(lo, hi, dtype=torch.int64) * 3 + 1).unsqueeze(0)

        def coarse_local(qq, **kw):      # pass 1 on this shard, deterministic order (score desc, position asc)
            return O.select_topk_deterministic(O.avg_topk_coarse_scores(cfg, w, qq, X).float(), min(avg_k, hi - lo))

        def rerank_local(qq, idx, kk, **kw):   # pass 2 on a candidate list with holes
            hole = idx < 0
            cand = X.squeeze(0)[idx.clamp_min(0)]
            sc = O.mol_stages(cfg, w, qq, cand)["logits"]
            sc = torch.where(hole, torch.full_like(sc, float("-inf")), sc)
            s, pos = O.select_topk_deterministic(sc, min(kk, idx.shape[1]))
            cid = torch.where(hole, torch.full_like(idx, -1), ids.reshape(-1)[idx.clamp_min(0)])
            return s, torch.gather(cid, 1, pos)

        def merge(scores, all_ids, kk):
            s, pos = O.select_topk_deterministic(scores, kk)
            return s, torch.gather(all_ids, 1, pos)

        mod = ShardedMoLAvgTopK(None, None, ids, n_items, avg_top_k=avg_k, global_k_prime=True, coarse_local=coarse_local,
                                rerank_local=rerank_local, merge=merge)
        ret[rank] = tuple(t.clone() for t in mod(q, k=k))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items,avg_k", [(600, 60), (70, 50)])   # second case: K' exceeds the items of a shard
def test_sharded_two_pass_with_global_k_prime_equals_the_single_device_algorithm(n_items, avg_k):
    """global_k_prime=True: coarse candidates are exchanged first, so exactly the GLOBAL coarse top-K' is reranked -- the result
    must equal MoLAvgTopK over the whole corpus (with the deterministic tie rule at both selections)."""
    world, k = 2, 20
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_global_worker, args=(world, _free_port(), n_items, k, avg_k, ret), nprocs=world, join=True)
    cfg = O.CONFIGS["amzn-books"]
    w = O.synthetic_weights(cfg, seed=0)
    q = O.synthetic_queries(cfg, 4)
    X = torch.from_numpy(O.hash_item_table(1, 0, n_items, cfg.item_embedding_dim)).unsqueeze(0)
    ids = torch.arange(0, n_items, dtype=torch.int64) * 3 + 1
    _, coarse_idx = O.select_topk_deterministic(O.avg_topk_coarse_scores(cfg, w, q, X).float(), avg_k)
    sc = O.mol_stages(cfg, w, q, X.squeeze(0)[coarse_idx])["logits"]
    es, pos = O.select_topk_deterministic(sc, k)
    ei = ids[torch.gather(coarse_idx, 1, pos)]
    for rank in range(world):
        s, i = ret[rank]
        assert torch.equal(i, ei) and torch.allclose(s, es, atol=1e-6)


def test_shard_bounds_cover_the_corpus():
    for n in (0, 1, 7, 695762, 10**9):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= (n + world - 1) // world


def test_message_roundtrip_is_bit_exact():
    g = torch.Generator().manual_seed(0)
    s = torch.randn((3, 4, 7), generator=g)
    s[0, 0, 0] = float("-inf")
    s[1, 1, 1] = -0.0
    i = torch.randint(0, 2**40, (3, 4, 7), generator=g)
    msgs = torch.stack([pack_candidates(s[r], i[r], 9) for r in range(3)])     # k=9 > 7: padded
    us, ui = unpack_candidates(msgs, 9)
    assert us.shape == (4, 27) and ui.shape == (4, 27)
    for r in range(3):
        assert torch.equal(us[:, r * 9 : r * 9 + 7].view(torch.int32), s[r].view(torch.int32))
        assert torch.equal(ui[:, r * 9 : r * 9 + 7], i[r])
        assert bool(torch.isinf(us[:, r * 9 + 7 : r * 9 + 9]).all()) and bool((ui[:, r * 9 + 7 : r * 9 + 9] == -1).all())


class _SpeculatingLocal:
    """Stand-in for a local module with the two-stage form of forward (MoLAvgTopK.submit / result): submit hands out tensors at once;
    when `fail` is set they are WRONG (a fused scan whose verdict says redo) and result returns the right ones as new tensors."""

    def __init__(self, cfg, w, X, ids, fail):
        self.cfg, self.w, self.X, self.ids, self.fail = cfg, w, X, ids, fail
        self.num_items = int(ids.numel())
        self.redone = 0

    def _right(self, q, k):
        s, pos = O.select_topk_deterministic(O.mol_logits(self.cfg, self.w, q, self.X), k)
        return s, self.ids.reshape(-1)[pos]

    def __call__(self, q, k, **kw):
        return self._right(q, k)

    def submit(self, q, k, sorted=True, **kw):
        s, i = self._right(q, k)
        if self.fail:
            return ("speculative", torch.zeros_like(s), torch.zeros_like(i), q, k)
        return ("speculative", s, i, q, k)

    def result(self, handle):
        if self.fail:
            self.redone += 1
            return self._right(handle[3], handle[4])
        return handle[1], handle[2]


def _spec_worker(rank: int, world: int, port: int, n_items: int, k: int, ret):
    from rails_amd.sharded import ShardedTopK

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cfg = O.CONFIGS["amzn-books"]
        w = O.synthetic_weights(cfg, seed=0)
        q = O.synthetic_queries(cfg, 5)
        lo, hi = shard_bounds(n_items, world, rank)
        X = torch.from_numpy(O.hash_item_table(1, lo, hi - lo, cfg.item_embedding_dim)).unsqueeze(0)
        ids = (torch.arange(lo, hi, dtype=torch.int64) * 3 + 1).unsqueeze(0)
        local = _SpeculatingLocal(cfg, w, X, ids, fail=(rank == 1))

        class Sharded(ShardedTopK):
            def _make_local_module(self, mol_module, item_embeddings_shard, item_ids_shard):
                return local

        def merge(scores, all_ids, kk):
            s, pos = O.select_topk_deterministic(scores, kk)
            return s, torch.gather(all_ids, 1, pos)

        mod = Sharded(None, X, ids, n_items, merge=merge)
        h1, h2 = mod.submit(q, k), mod.submit(q, k)      # two batches in flight before either result
        s, i = mod.result(h1)
        s2, i2 = mod.result(h2)
        assert torch.equal(s, s2) and torch.equal(i, i2)
        assert local.redone == (2 if rank == 1 else 0)
        ret[rank] = (s.clone(), i.clone())
    finally:
        dist.destroy_process_group()


def test_sharded_topk_carries_a_speculative_local_handle():
    """ShardedTopK.submit packs what a local module's submit() hands out; result() first asks the local module for the verified
    output and packs again when it differs (rank 1's speculation fails here).  Merged result == the unsharded oracle on both ranks."""
    world, n_items, k = 2, 600, 40
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_spec_worker, args=(world, _free_port(), n_items, k, ret), nprocs=world, join=True)
    cfg = O.CONFIGS["amzn-books"]
    w = O.synthetic_weights(cfg, seed=0)
    q = O.synthetic_queries(cfg, 5)
    X = torch.from_numpy(O.hash_item_table(1, 0, n_items, cfg.item_embedding_dim)).unsqueeze(0)
    ids = torch.arange(0, n_items, dtype=torch.int64) * 3 + 1
    rs, rpos = O.select_topk_deterministic(O.mol_logits(cfg, w, q, X), k)
    for rank in range(world):
        s, i = ret[rank]
        assert torch.equal(s, rs) and torch.equal(i, ids[rpos])
