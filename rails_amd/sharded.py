"""Item-sharded MoL top-k (exact, and the two-pass approximate one): one process per GPU, one small all-gather per batch.

The reference has no sharded retrieval (eval is asserted single-GPU, eval_from_checkpoint.py:554-555); this
is the north-star's 8-GPU path.  Its oracle is "equals single-device brute force over the concatenated
corpus", which holds bit for bit because (a) per-pair arithmetic does not depend on the shard and
(b) selection uses the total order (score desc, global position asc) at both levels.

  rank r owns items [r * ceil(N/R), min(N, (r+1) * ceil(N/R)))      (contiguous item-id ranges)
  queries and MoL weights are replicated (KBs); every rank redoes the query prologue
  per batch: local scoring -> local top-k -> ONE all_gather of B*k*16 bytes -> merge R*k -> k on every rank

ShardedMoLAvgTopK (BASELINE config 5: 1 B items 8-way, coarse prefilter + MoL rerank) has the same shape: every rank
runs MoLAvgTopK on its shard -- coarse top-K' of ITS items, full MoL on those, local top-k -- and the merge is the same,
because what is merged are exact MoL scores.  It reranks R*K' candidates in total (K' per shard), a superset-quality
variant of the single-device algorithm with the same K'; it equals it exactly when R = 1.

ShardedMoLAvgTopK(global_k_prime=True) is the single-device algorithm itself on a sharded corpus:
one more all-gather first exchanges every shard's coarse top-K' (coarse score bits | global position), every rank
selects the GLOBAL coarse top-K' with the same total order (score desc, global position asc), reranks only its own
members of it, and the usual merge follows.  Bit-identical to MoLAvgTopK(avg_top_k = K') over the whole corpus.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import contextlib

import torch
import torch.distributed as dist

from . import engine as E
from .topk_modules import MoLAvgTopK, MoLBruteForceTopK, TopKModule


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    per = (n_items + world_size - 1) // world_size
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def pack_candidates(scores: torch.Tensor, ids: torch.Tensor, k: int) -> torch.Tensor:
    """(B, k_local) fp32 scores + int64 ids -> one (B, 2k) int64 message (score bits | ids); rows shorter than
    k are padded with -inf / id -1 so every rank sends the same size."""
    B, kl = scores.shape
    if kl < k:
        scores = torch.cat([scores, scores.new_full((B, k - kl), float("-inf"))], 1)
        ids = torch.cat([ids, ids.new_full((B, k - kl), -1)], 1)
    return torch.cat([scores.contiguous().view(torch.int32).to(torch.int64), ids], 1)


def unpack_candidates(msg: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(R, B, 2k) gathered messages -> (B, R*k) scores and ids in shard-major order."""
    R, B, _ = msg.shape
    scores = msg[:, :, :k].to(torch.int32).view(torch.float32)
    ids = msg[:, :, k:]
    return scores.permute(1, 0, 2).reshape(B, R * k).contiguous(), ids.permute(1, 0, 2).reshape(B, R * k).contiguous()


def _hip_merge(scores: torch.Tensor, ids: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    return E.topk(scores, k, ids=ids)


class ShardedTopK(TopKModule):
    """forward(query_embeddings, k) -> (scores (B, k), ids (B, k)), identical on every rank: the local module's top-k of
    every shard, merged.  `local_topk` / `merge` default to the HIP kernels; the CPU tests of the collective logic inject
    oracle-backed callables instead (gloo, world_size 2)."""

    EXCHANGE_WITH_ONE_RANK = False

    def _make_local_module(self, mol_module, item_embeddings_shard, item_ids_shard) -> TopKModule:
        raise NotImplementedError

    def __init__(
        self,
        mol_module,
        item_embeddings_shard: Optional[torch.Tensor],
        item_ids_shard: Optional[torch.Tensor],
        n_items_total: int,
        group: Optional[dist.ProcessGroup] = None,
        local_topk: Optional[Callable[..., Tuple[torch.Tensor, torch.Tensor]]] = None,
        merge: Optional[Callable[[torch.Tensor, torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]]] = None,
    ) -> None:
        super().__init__()
        self._group = group
        self._world = dist.get_world_size(group) if dist.is_initialized() else 1
        # one shard = the local module's own result, no exchange -- unless EXCHANGE_WITH_ONE_RANK (a test setting: a one-GPU box runs the whole
        # exchange path -- RCCL all-gather on the exchange stream, merge, global verdict -- in a process group of one rank)
        self._exchange = self._world > 1 or (self.EXCHANGE_WITH_ONE_RANK and dist.is_initialized())
        self._n_total = n_items_total
        if local_topk is None:
            self._local_module = self._make_local_module(mol_module, item_embeddings_shard, item_ids_shard)
            self._n_local = self._local_module.num_items
            local_topk = lambda q, k, **kw: self._local_module(q, k=k, **kw)  # noqa: E731
            self._local_topk_is_module = True
        else:
            self._n_local = int(item_ids_shard.numel())
            self._local_topk_is_module = False
        self._local_topk = local_topk
        self._merge = merge if merge is not None else _hip_merge
        self._xstream = None     # exchange stream (all-gather + merge), created on first GPU use

    def forward(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        with self._inline():     # a plain call has no neighbouring batch to overlap with (MoLAvgTopK.submit)
            return self.result(self.submit(query_embeddings, k, sorted, **kwargs))

    _plain_call = False
    EXCHANGE_STREAM = False    # True: submit / result run the exchange (all-gather + merge) on a second stream, concurrently with the next batch's
                               # scoring.  Measured through the real module in a one-rank nccl group (tools/r06_shard_rccl_probe.py, 8-way shard):
                               # the two stream hand-overs per step cost more than the overlap gains (0.474 against 0.455 ms for the plain call); with
                               # everything on the caller's stream, submit / result two batches ahead keeps the device queue full across the host's look
                               # at the verdict with no hand-over at all.

    @contextlib.contextmanager
    def _inline(self):
        """A plain forward / forward_filtered: there is no neighbouring batch to overlap with, so the local module stays on the caller's stream
        (MoLAvgTopK.inline_calls) and so does the exchange -- every hand-over between streams is an event the GPU waits 5-12 us for (round 6:
        two of them per step on an 8-way shard's 0.44 ms)."""
        local = getattr(self, "_local_module", None)
        saved = self._plain_call
        self._plain_call = True
        try:
            with (local.inline_calls() if hasattr(local, "inline_calls") else contextlib.nullcontext()):
                yield
        finally:
            self._plain_call = saved

    # ---- two-stage form of forward: submit() enqueues this rank's part, result() the exchange ------------------------------
    # A caller that has the next batch at hand calls submit(batch i+1) BEFORE result(batch i): the all-gather and the merge of
    # batch i then run on a second stream while batch i+1's prologue and scoring occupy the first ("overlap
    # it with the next batch's scoring").  Same kernels, same order of arithmetic: the output is bit-equal to forward's.
    def submit(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs):
        """Local scoring + local top-k + pack on the current stream -> handle for result()."""
        if k > self._n_total:
            raise RuntimeError(f"selected index k out of range (k={k}, n={self._n_total})")
        k_local = min(k, self._n_local)
        local = getattr(self, "_local_module", None)
        spec = None   # a local module with its own submit / result (MoLAvgTopK): its speculative output travels on, verified in result()
        if k_local > 0 and local is not None and hasattr(local, "submit") and self._local_topk_is_module:
            spec = local.submit(query_embeddings, k_local, **kwargs)
            s, ids = spec[1], spec[2]
            if spec[0] == "final":
                spec = None
            elif self._exchange and s.is_cuda and isinstance(spec[-1], torch.cuda.Stream):
                # the local call ran on a stream of its own and the pack below reads its output here: join it now (with one shard
                # nothing reads it before result(), and batches overlap)
                torch.cuda.current_stream(s.device).wait_event(spec[4])
        elif k_local > 0:
            s, ids = self._local_topk(query_embeddings, k_local, **kwargs)
        else:  # an empty shard still takes part in the collective
            B = query_embeddings.size(0)
            s = torch.empty((B, 0), dtype=torch.float32, device=query_embeddings.device)
            ids = torch.empty((B, 0), dtype=torch.int64, device=query_embeddings.device)
        if not self._exchange:
            return ("done", s, ids, spec)
        on_gpu = s.is_cuda and self._merge is _hip_merge
        msg = E.pack_candidates(s, ids, k) if on_gpu else pack_candidates(s.float(), ids, k)
        ready = None
        if msg.is_cuda:
            ready = torch.cuda.Event()
            ready.record()
        return ("pending", msg, ready, k, on_gpu, s.dtype, spec)

    def forward_filtered(self, query_embeddings: torch.Tensor, k_prime: int, invalid_ids: torch.Tensor, k: int, **kwargs):
        """CandidateIndex.get_top_k_outputs' body for the sharded modules: the seen-id filter runs inside the merge launch
        (rails_merge_candidates_filtered) -> (top_k_ids (B, k), top_k_scores (B, k)), or None when the sizes are outside the fused path
        or the merge is not the HIP one (the caller then composes forward + filter_seen_ids: same bits)."""
        if not self._exchange:
            local = getattr(self, "_local_module", None)
            return local.forward_filtered(query_embeddings, k_prime, invalid_ids, k, **kwargs) if hasattr(local, "forward_filtered") else None
        if not (query_embeddings.is_cuda and self._merge is _hip_merge and E.merge_filter_fusable(k_prime, invalid_ids.shape[1], k)) or k_prime > self._n_total:
            return None
        with self._inline():
            return self.result(self.submit(query_embeddings, k_prime, **kwargs), seen=(invalid_ids, k))

    def result(self, handle, seen=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """All-gather of the per-shard candidates + merge -> (scores, ids), identical on every rank.  On the GPU both run on this
        module's exchange stream, behind the handle's event; the current stream waits for the merge only.
        seen = (invalid_ids, k): the seen-id filter inside the merge launch -> (ids (B, k), scores (B, k)) instead (GPU merge only)."""
        if handle[0] == "done":
            s, ids = handle[1], handle[2]
            if handle[3] is not None:
                s, ids = self._local_module.result(handle[3])     # the verified output (the same tensors unless the call was redone)
            if seen is not None:
                return E.filter_seen_ids(ids, s, seen[0], seen[1])
            return s, ids
        _, msg, ready, k, on_gpu, dtype, spec = handle
        if spec is not None:
            s, ids = self._local_module.result(spec)
            if s is not spec[1]:                                   # redone on the materialising path: pack again
                msg = E.pack_candidates(s, ids, k) if on_gpu else pack_candidates(s.float(), ids, k)
                if msg.is_cuda:
                    ready = torch.cuda.Event()
                    ready.record()
        if seen is not None and not (msg.is_cuda and on_gpu):
            raise RuntimeError("result(seen=...) needs the HIP merge")
        if not msg.is_cuda:
            gathered = torch.empty((self._world * msg.shape[0], msg.shape[1]), dtype=msg.dtype)
            dist.all_gather_into_tensor(gathered, msg, group=self._group)
            all_s, all_ids = unpack_candidates(gathered.view(self._world, msg.shape[0], msg.shape[1]), k)
            ms, mi = self._merge(all_s, all_ids, k)
            return ms.to(dtype), mi
        cur = torch.cuda.current_stream(msg.device)
        plain = self._plain_call or not self.EXCHANGE_STREAM
        if plain:
            side = cur
            if not self._plain_call and ready is not None:
                cur.wait_event(ready)      # submit() may have run on another stream than the one result() is called on
        else:
            if self._xstream is None:
                self._xstream = torch.cuda.Stream(msg.device)
            side = self._xstream
            side.wait_event(ready)
            msg.record_stream(side)
        with (contextlib.nullcontext() if plain else torch.cuda.stream(side)):
            # concatenated-along-dim-0 output: the layout both RCCL and gloo accept for all_gather_into_tensor
            if dist.get_backend(self._group) == "gloo":   # test setups only: stage through the host
                host = torch.empty((self._world * msg.shape[0], msg.shape[1]), dtype=msg.dtype)
                dist.all_gather_into_tensor(host, msg.cpu(), group=self._group)
                gathered = host.to(msg.device)
            else:
                gathered = torch.empty((self._world * msg.shape[0], msg.shape[1]), dtype=msg.dtype, device=msg.device)
                dist.all_gather_into_tensor(gathered, msg, group=self._group)
            if on_gpu and seen is not None:   # one kernel: rank-major candidates -> exact top-k -> seen-id filter
                mi, ms = E.merge_candidates_filtered(gathered, self._world, k, k, seen[0], seen[1])
                if not plain:
                    seen[0].record_stream(side)
            elif on_gpu:   # one kernel: rank-major candidates -> exact top-k (scores, ids)
                ms, mi = E.merge_candidates(gathered, self._world, k, k)
            else:
                all_s, all_ids = unpack_candidates(gathered.view(self._world, msg.shape[0], msg.shape[1]), k)
                ms, mi = self._merge(all_s, all_ids, k)
            ms = ms.to(dtype)
        if not plain:
            cur.wait_stream(side)
            ms.record_stream(cur)
            mi.record_stream(cur)
        if seen is not None:
            return mi, ms
        return ms, mi

    def exchange_info(self) -> dict:
        """What carried the exchange: backend of the process group and its size (bench.py reports it)."""
        if not dist.is_initialized():
            return {"backend": None, "ranks": 1}
        return {"backend": dist.get_backend(self._group), "ranks": dist.get_world_size(self._group)}


class ShardedMoLBruteForceTopK(ShardedTopK):
    """Exact: identical to MoLBruteForceTopK over the whole corpus, bit for bit (see the module docstring).

    GLOBAL PROOF (round 5).  The single-device module's default exact path -- split-f16 first pass, fp32 re-scoring of the candidates, proof
    with the a-priori bound eps on |first pass - fp32| (topk_modules.MoLBruteForceTopK, f16x3_bound.py) -- proved PER SHARD would need
    every shard to cover what lies within eps of ITS OWN k-th score, which sits in a denser part of the score distribution than the whole
    corpus' (8 shards of amzn-books: thousands of candidates per shard and query instead of ~85).  So the proof is made once, globally:
      rank r:  first pass over its shard -> its kc_r best by first-pass score, m_r = the best first-pass score it leaves outside
               -> fp32 re-scoring of the kc_r -> its best k by (fp32 score, position)
      all:     ONE all-gather of (B, 2k + 2) messages -- the rank's top-k, m_r per row and the largest |fp32 - first pass| it saw ride in the
               same message (round 6; rounds 5 used an all-reduce next to the gather and an always-enqueued second gather for the redo)
               -> merge to the global top-k by fp32 score + verdict per row, e_k - max_r m_r > eps, in ONE launch
    Every item of every shard outside the candidates has a first-pass score <= max_r m_r, hence an fp32 score < e_k: the merged top-k IS the
    dense one.  Candidates per rank: the ~kc items within eps of the global k-th score spread evenly over the shards, so kc_r = kc / R +
    4 sqrt(kc / R) + 32 (doubled after a failed verdict, halved again after PAD_DECAY_CALLS proved calls: both decided from the verdicts,
    which every rank sees alike).  A failed verdict (crowded scores, a skewed shard, a violated guard) is the same on every rank -- its
    inputs are the gathered bytes -- and is read by the host from the pinned mirror the merge kernel writes; only then do the ranks run
    the dense fp32 kernels over their shards and a second exchange.  Used when EVERY rank's local module is bound in proved mode (an
    all-reduce at the first call); otherwise the per-shard path above."""

    GLOBAL_PROOF = True
    PAD_DECAY_CALLS = 64          # after this many consecutive proved calls a doubled candidate margin is halved again
    VERDICT_TIMEOUT_S = 120.0

    def _make_local_module(self, mol_module, item_embeddings_shard, item_ids_shard) -> TopKModule:
        # the size-dependent choices of the proved flow (one eps or per-pair bounds, candidate margins) are made for the SHARD size every rank
        # computes alike -- the last shard may be shorter, and ranks must agree on the form of the bound
        return MoLBruteForceTopK(mol_module, item_embeddings_shard, item_ids_shard, bound_kind_items=-(-self._n_total // max(self._world, 1)))

    # ---- collectives on small tensors (host-staged only on a gloo group: test setups) --------------------------------------------
    def _all_reduce(self, t: torch.Tensor, op) -> torch.Tensor:
        if t.is_cuda and dist.get_backend(self._group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self._group)
            return h.to(t.device)
        dist.all_reduce(t, op=op, group=self._group)
        return t

    def _all_gather_rows(self, msg: torch.Tensor) -> torch.Tensor:
        self._gp_collectives = getattr(self, "_gp_collectives", 0) + 1
        if msg.is_cuda and dist.get_backend(self._group) == "gloo":
            host = torch.empty((self._world * msg.shape[0], msg.shape[1]), dtype=msg.dtype)
            dist.all_gather_into_tensor(host, msg.cpu(), group=self._group)
            return host.to(msg.device)
        out = torch.empty((self._world * msg.shape[0], msg.shape[1]), dtype=msg.dtype, device=msg.device)
        dist.all_gather_into_tensor(out, msg, group=self._group)
        return out

    def _global_proof(self, query_embeddings: torch.Tensor) -> bool:
        """Decided collectively, once per binding of the local module (all ranks reach this at the same call)."""
        local = getattr(self, "_local_module", None)
        if not (self.GLOBAL_PROOF and self._exchange and dist.is_initialized() and isinstance(local, MoLBruteForceTopK) and self._local_topk_is_module
                and self._merge is _hip_merge and query_embeddings.is_cuda):
            return False
        eng = local._bind()
        if getattr(self, "_gp_engine", None) is not eng:
            from . import f16x3_bound as FB

            mine = local.shard_can_speculate()
            flags = torch.tensor([1.0 if mine else 0.0, -(local._gi_abs_max() if mine else 0.0)], dtype=torch.float32, device=query_embeddings.device)
            flags = self._all_reduce(flags, dist.ReduceOp.MIN)          # min of the flags, max of max |gi| (negated)
            self._gp_on = bool(flags[0].item() > 0.5)
            gi_max = -float(flags[1].item())
            self._gp_guard_limit = min(FB.GATE_GUARD / gi_max, 3.0e38) if gi_max > 0.0 else 3.0e38
            self._gp_engine = eng
            self._gp_eps = local._proved_eps() if self._gp_on else None
            self._gp_state = torch.zeros(8, dtype=torch.float32, device=query_embeddings.device)
            self._gp_host = torch.zeros(8, dtype=torch.float32).pin_memory()
            self._gp_host_f = self._gp_host.numpy()                         # views of the same pinned words: a poll is a plain memory read,
            self._gp_host_i = self._gp_host.view(torch.int32).numpy()       # not a tensor index + conversion (2-3 us each, between the batches)
            self._gp_call = torch.zeros(8 + 4 * 256, dtype=torch.int32, device=query_embeddings.device)      # arrival counter + one 16-byte word per row
            self._gp_issued = 0           # verdicts enqueued so far (the host mirror's call counter reaches it when the last one has landed)
            self._gp_pad = 1
            self._gp_streak = 0
            self._gp_stats = {"calls": 0, "fallbacks": 0, "proved_calls": 0, "bound_violations": 0}
        return self._gp_on

    def stats(self) -> dict:
        """Counters of the global proof (calls, proved_calls, fallbacks, bound_violations, kc per rank) merged over the local module's."""
        local = self._local_module
        out = local.stats()
        if getattr(self, "_gp_on", False):
            out.update(self._gp_stats)
            out["global_proof"] = True
        return out

    def _kc_local(self, k: int) -> int:
        upper = self._local_module._upper_poly() is not None       # per-pair upper bounds: more items can reach the k-th score (topk_modules._forward_rescored)
        floor, per_k = self._local_module._per_pair_pad() if upper else MoLBruteForceTopK.PAD_ONE_EPS
        total = k + max(floor, per_k * k) * self._gp_pad
        per = -(-total // self._world)
        kc = per + int(4.0 * per ** 0.5) + 32
        kc = (kc + E.TILE_ITEMS - 1) // E.TILE_ITEMS * E.TILE_ITEMS
        return max(1, min(kc, 16384, self._n_local))

    def submit(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs):
        B = query_embeddings.size(0)
        per_shard = -(-self._n_total // max(self._world, 1))
        # (the 4 GiB logit policy: the first pass wants the whole (B, N_shard) matrix -- beyond it every rank alike takes the per-shard path, which chunks)
        if (not self._global_proof(query_embeddings) or not MoLBruteForceTopK.speculation_pays(B, per_shard) or B * per_shard * 4 > MoLBruteForceTopK.MAX_LOGIT_BYTES
                or not MoLBruteForceTopK.FUSED_TAIL):
            return super().submit(query_embeddings, k, sorted, **kwargs)
        if k > self._n_total:
            raise RuntimeError(f"selected index k out of range (k={k}, n={self._n_total})")
        local = self._local_module
        with local.one_bind():
            kc = self._kc_local(k)
            msg, qpack32 = local.speculate_for_shard(query_embeddings, k, kc, **kwargs)
        ready = None
        if not self._plain_call:      # (a plain call's exchange follows on this very stream; result() of an explicit submit may be called on another)
            ready = torch.cuda.Event()
            ready.record()
        self._gp_stats["kc"] = kc
        return ("gproof", msg, ready, k, qpack32, query_embeddings, kwargs, sorted)

    def result(self, handle, seen=None) -> Tuple[torch.Tensor, torch.Tensor]:
        if handle[0] != "gproof":
            return super().result(handle, seen)
        _, msg, ready, k, qpack32, query_embeddings, kwargs, sorted_ = handle
        local = self._local_module
        sp = local._engine.spec
        B = query_embeddings.size(0)
        off = (B + 32 // sp.query_dot_product_groups - 1) // (32 // sp.query_dot_product_groups) * 32 * sp.dot_product_dimension
        gq = qpack32[off : off + B * sp.num_logits]
        cur = torch.cuda.current_stream(msg.device)
        plain = self._plain_call or not self.EXCHANGE_STREAM          # the exchange stays on the caller's stream (always for a plain call: _inline)
        if plain:
            side = cur
            if not self._plain_call and ready is not None:
                cur.wait_event(ready)      # submit() may have run on another stream than the one result() is called on
        else:
            if self._xstream is None:
                self._xstream = torch.cuda.Stream(msg.device)
            side = self._xstream
            side.wait_event(ready)
            for t in (msg, qpack32):
                t.record_stream(side)
        fuse = seen is not None and E.merge_filter_fusable(k, seen[0].shape[1], seen[1])
        if self._gp_call.numel() < 8 + 4 * B:
            self._gp_call = torch.zeros(8 + 4 * B, dtype=torch.int32, device=msg.device)
        with (contextlib.nullcontext() if plain else torch.cuda.stream(side)):
            # ONE exchange: the (B, 2k + 2) messages carry every rank's top-k, the best first-pass score it left outside its candidates and the
            # largest |fp32 - first pass| it saw; merge, verdict and the seen-id filter are one launch behind it
            gathered = self._all_gather_rows(msg)
            if fuse and not plain:
                seen[0].record_stream(side)
            out = E.merge_candidates_verdict(gathered, self._world, k, k, self._gp_eps, 1.0, gq, sp.num_logits, self._gp_guard_limit, self._gp_state,
                                             self._gp_host, self._gp_call, seen if fuse else None)
            if seen is not None and not fuse:
                if not plain:
                    seen[0].record_stream(side)
                out = E.filter_seen_ids(out[1], out[0], seen[0], seen[1])
        self._gp_issued += 1
        if not plain:
            cur.wait_stream(side)
            out[0].record_stream(cur)
            out[1].record_stream(cur)
        # Every rank computes the verdict from the same gathered bytes, so all of them raise or clear REDO alike; the host reads it from the pinned
        # mirror the merge kernel writes (with submit / result pipelining batch i + 1 is already enqueued: the device does not idle) and only a
        # failed verdict -- crowded scores, a skewed shard, a violated guard -- costs more: the dense fp32 kernels over the shards and their
        # own exchange, issued here, by every rank.
        redo = self._gp_wait_verdict()
        st = self._gp_stats
        st["calls"] += 1
        if float(self._gp_host_f[0]) > self._gp_eps:
            st["bound_violations"] += 1
        elif not redo:
            st["proved_calls"] += 1
        st["guard_max"] = float(self._gp_host_f[7])
        if not redo:
            self._gp_streak += 1
            if self._gp_pad > 1 and self._gp_streak >= self.PAD_DECAY_CALLS:
                self._gp_pad //= 2
                self._gp_streak = 0
            if seen is not None:
                return out[0], out[1].to(query_embeddings.dtype)
            return out[0].to(query_embeddings.dtype), out[1]
        st["fallbacks"] += 1
        self._gp_streak = 0
        if self._gp_pad < 64:
            self._gp_pad *= 2
        # the dense fp32 kernels over this shard (the local module's resident fp32 index) and the plain exchange of the per-shard top-k
        k_local = min(k, self._n_local)
        if k_local > 0:
            with local.one_bind():
                s, ids = local._forward_fp32_dense(query_embeddings, k_local, **kwargs)
        else:
            s = torch.empty((B, 0), dtype=torch.float32, device=msg.device)
            ids = torch.empty((B, 0), dtype=torch.int64, device=msg.device)
        msg2 = E.pack_candidates(s.float(), ids, k)
        ready2 = torch.cuda.Event()
        ready2.record()
        return super().result(("pending", msg2, ready2, k, True, query_embeddings.dtype, None), seen)

    def _gp_wait_verdict(self) -> bool:
        """Spin on the pinned mirror's call counter until the verdict enqueued last has landed -> its REDO flag."""
        import time

        h = self._gp_host_f
        want = float(self._gp_issued)
        t0 = None
        n = 0
        while h[5] < want:
            n += 1
            if (n & 1023) == 0:
                if t0 is None:
                    t0 = time.perf_counter()
                elif time.perf_counter() - t0 > self.VERDICT_TIMEOUT_S:
                    raise RuntimeError("the item-sharded verdict did not arrive (a kernel or the exchange failed)")
        return int(self._gp_host_i[1]) != 0

    def exchange_info(self) -> dict:
        info = super().exchange_info()
        info["collectives_per_proved_step"] = 1
        info["collectives_issued"] = getattr(self, "_gp_collectives", 0)
        return info

    def forward_filtered(self, query_embeddings: torch.Tensor, k_prime: int, invalid_ids: torch.Tensor, k: int, **kwargs):
        if self._exchange and self._global_proof(query_embeddings) and MoLBruteForceTopK.speculation_pays(query_embeddings.size(0), -(-self._n_total // self._world)) and k_prime <= self._n_total:
            with self._inline():
                return self.result(self.submit(query_embeddings, k_prime, **kwargs), seen=(invalid_ids, k))
        return super().forward_filtered(query_embeddings, k_prime, invalid_ids, k, **kwargs)


class ShardedMoLAvgTopK(ShardedTopK):
    """Two-pass approximate top-k on an item-sharded corpus (BASELINE config 5): MoLAvgTopK(avg_top_k) per shard, then
    the same single all-gather + merge.  `avg_top_k` is PER SHARD; k <= avg_top_k as in the reference
    (rails/indexing/mol_top_k.py:383-386)."""

    def __init__(self, mol_module, item_embeddings_shard, item_ids_shard, n_items_total: int, avg_top_k: int,
                 global_k_prime: bool = False, shard_offset: Optional[int] = None,
                 coarse_local: Optional[Callable[..., Tuple[torch.Tensor, torch.Tensor]]] = None,
                 rerank_local: Optional[Callable[..., Tuple[torch.Tensor, torch.Tensor]]] = None, **kwargs) -> None:
        """global_k_prime: exchange coarse candidates first so that exactly the global coarse top-K' is reranked (see the module
        docstring); `shard_offset` = global position of this shard's first item (default: the contiguous split of shard_bounds).
        `coarse_local(q, **kw) -> (scores, local positions)` / `rerank_local(q, local positions with -1 holes, k, **kw) ->
        (scores, ids)` default to the HIP module's methods; the CPU test of the collective logic injects oracle callables."""
        self._avg_top_k = avg_top_k
        self._global = global_k_prime
        if coarse_local is not None and "local_topk" not in kwargs:
            kwargs["local_topk"] = lambda q, k, **kw: (_ for _ in ()).throw(RuntimeError("global_k_prime path only"))   # noqa: E731
        super().__init__(mol_module, item_embeddings_shard, item_ids_shard, n_items_total, **kwargs)
        rank = dist.get_rank(self._group) if dist.is_initialized() else 0
        self._offset = shard_offset if shard_offset is not None else shard_bounds(n_items_total, self._world, rank)[0]
        if self._global and self._exchange and shard_offset is not None:
            # ties are broken by the slot in the rank-major concatenation: that is the global position order only when the shards
            # are disjoint position ranges in rank order
            spans = [None] * self._world
            dist.all_gather_object(spans, (int(self._offset), int(self._offset + self._n_local)), group=self._group)
            if any(spans[r][1] > spans[r + 1][0] for r in range(self._world - 1)):
                raise ValueError(f"global_k_prime needs shard ranges that ascend with the rank without overlap, got {spans}")
        self._coarse_local = coarse_local if coarse_local is not None else (lambda q, **kw: self._local_module.coarse_candidates(q, **kw))
        self._rerank_local = rerank_local if rerank_local is not None else (lambda q, idx, k, **kw: self._local_module.rerank_masked(q, idx, k, **kw))

    def _make_local_module(self, mol_module, item_embeddings_shard, item_ids_shard) -> TopKModule:
        return MoLAvgTopK(mol_module, item_embeddings_shard, item_ids_shard, avg_top_k=min(self._avg_top_k, int(item_ids_shard.numel())))

    def _gather(self, msg: torch.Tensor) -> torch.Tensor:
        """(B, W) -> (world * B, W), rank-major (host-staged only for device tensors on a gloo group: test setups)."""
        if msg.is_cuda and dist.get_backend(self._group) == "gloo":
            host = torch.empty((self._world * msg.shape[0], msg.shape[1]), dtype=msg.dtype)
            dist.all_gather_into_tensor(host, msg.cpu(), group=self._group)
            return host.to(msg.device)
        out = torch.empty((self._world * msg.shape[0], msg.shape[1]), dtype=msg.dtype, device=msg.device)
        dist.all_gather_into_tensor(out, msg, group=self._group)
        return out

    def forward_filtered(self, query_embeddings: torch.Tensor, k_prime: int, invalid_ids: torch.Tensor, k: int, **kwargs):
        if k_prime > self._avg_top_k or (self._global and self._exchange):
            return None   # forward's own checks / the global-K' exchange: the caller composes forward + filter_seen_ids
        return super().forward_filtered(query_embeddings, k_prime, invalid_ids, k, **kwargs)

    def forward(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        if k > self._avg_top_k:
            raise ValueError(f"avg_top_k ({self._avg_top_k}) must be larger than k ({k})")
        if not self._global or not self._exchange:
            return super().forward(query_embeddings, k, sorted, **kwargs)
        K = self._avg_top_k
        if K > self._n_total:
            raise RuntimeError(f"selected index k out of range (k={K}, n={self._n_total})")
        B = query_embeddings.size(0)
        dev = query_embeddings.device
        # (1) local coarse top-K' -> message (coarse score bits | GLOBAL position), K' slots, short shards padded with (-inf, -1)
        if self._n_local > 0:
            cs, cpos = self._coarse_local(query_embeddings, **kwargs)
            cpos = cpos + self._offset
        else:
            cs = torch.empty((B, 0), dtype=torch.float32, device=dev)
            cpos = torch.empty((B, 0), dtype=torch.int64, device=dev)
        on_gpu = cs.is_cuda and self._merge is _hip_merge
        msg = E.pack_candidates(cs, cpos, K) if on_gpu else pack_candidates(cs.float(), cpos, K)
        gathered = self._gather(msg)
        # (2) the global coarse top-K' (same total order on every rank: score desc, global position asc)
        if on_gpu and self._world * K <= 16384:   # one kernel (its lists are sorted in LDS)
            _, gpos = E.merge_candidates(gathered, self._world, K, K)
        elif on_gpu:                              # long lists: the general top-k over the shard-major concatenation (same tie rule)
            all_s, all_p = unpack_candidates(gathered.view(self._world, B, 2 * K), K)
            _, gpos = E.topk(all_s, K, ids=all_p)
        else:
            all_s, all_p = unpack_candidates(gathered.view(self._world, B, 2 * K), K)
            _, gpos = self._merge(all_s, all_p, K)
        # (3) rerank my members of it (others become holes), local top-k, and the usual exchange of exact MoL scores
        mine = (gpos >= self._offset) & (gpos < self._offset + self._n_local)
        local_idx = torch.where(mine, gpos - self._offset, gpos.new_full((), -1))
        if self._n_local > 0:
            s, ids = self._rerank_local(query_embeddings, local_idx, k, **kwargs)
        else:
            s = torch.full((B, k), float("-inf"), dtype=torch.float32, device=dev)
            ids = torch.full((B, k), -1, dtype=torch.int64, device=dev)
        msg2 = E.pack_candidates(s, ids, k) if on_gpu else pack_candidates(s.float(), ids, k)
        gathered2 = self._gather(msg2)
        if on_gpu:
            ms, mi = E.merge_candidates(gathered2, self._world, k, k)
        else:
            all_s, all_i = unpack_candidates(gathered2.view(self._world, B, 2 * k), k)
            ms, mi = self._merge(all_s, all_i, k)
        return ms.to(query_embeddings.dtype), mi
