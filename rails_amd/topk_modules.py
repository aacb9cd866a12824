"""Host-side mirror of the reference's top-k modules and candidate index.

  TopKModule            reference rails/indexing/candidate_index.py:24-42
  MoLBruteForceTopK     reference rails/indexing/mol_top_k.py:84-130   (exact)
  MoLAvgTopK            reference rails/indexing/mol_top_k.py:296-429  (two-pass approximate)
  CandidateIndex        reference indexing/candidate_index.py:30-185
  get_top_k_module      reference indexing/utils_rails.py:25-233

Unlike the reference, which keeps the raw (1, N, D) table and re-projects every item on every call
(mol_top_k.py:118-122), the modules here build the tile-packed item index once at construction
(rebuilt automatically if the MoL module's parameters change) and per call run: query prologue ->
fused scoring -> exact top-k -> id gather, all HIP.
"""
from __future__ import annotations

import abc
from typing import Dict, Optional, Tuple

import math

import contextlib

import torch

from . import engine as E
from .mol_module import MoLSimilarity


class TopKModule(torch.nn.Module):
    def __setattr__(self, name, value):
        """Plain Python state (counters, flags, cached engines, handles: a dozen assignments per call) goes straight to the instance dict:
        torch.nn.Module.__setattr__ walks its parameter / buffer / module registries for every assignment (1-2.5 us each -- host time that sits
        between the batches wherever a call ends with the host's look at a verdict).  Tensors, modules and names already registered keep the
        nn.Module path."""
        if not isinstance(value, (torch.Tensor, torch.nn.Module)):
            d = self.__dict__
            if name not in d.get("_parameters", ()) and name not in d.get("_buffers", ()) and name not in d.get("_modules", ()):
                d[name] = value
                return
        super().__setattr__(name, value)

    @abc.abstractmethod
    def forward(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (top_k_scores (B, k), top_k_ids (B, k))."""


class MoLTopKModule(TopKModule):
    """Common state of the MoL top-k modules (reference mol_top_k.py:29-81): borrows `item_embeddings`
    (1, N, D) and `item_ids` (1, N); owns the packed index."""

    def __init__(self, mol_module: MoLSimilarity, item_embeddings: torch.Tensor, item_ids: torch.Tensor) -> None:
        super().__init__()
        self._mol_module: MoLSimilarity = mol_module
        if item_embeddings.dim() != 3 or item_embeddings.shape[0] != 1:
            raise ValueError(f"item_embeddings must be (1, N, D), got {tuple(item_embeddings.shape)}")
        self._item_embeddings: torch.Tensor = item_embeddings
        self._item_ids: torch.Tensor = item_ids
        self._ids_flat: torch.Tensor = item_ids.reshape(-1).to(device=item_embeddings.device, dtype=torch.int64).contiguous()
        self._engine: Optional[E.MolEngine] = None
        self._call_eng: Optional[E.MolEngine] = None
        self._index: Optional[E.MolIndex] = None
        self._scratch: Dict[tuple, torch.Tensor] = {}   # internal buffers recycled across calls (never returned)
        self._bind()

    def _buf(self, tag: str, numel: int, dtype: torch.dtype) -> torch.Tensor:
        key = (tag, numel, dtype)
        t = self._scratch.get(key)
        if t is None:
            if len(self._scratch) > 16:
                self._scratch.clear()
            t = torch.empty(numel, dtype=dtype, device=self._item_embeddings.device)
            self._scratch[key] = t
        return t

    @property
    def mol_module(self) -> MoLSimilarity:
        return self._mol_module

    @property
    def num_items(self) -> int:
        return self._item_embeddings.shape[1]

    def _bind(self) -> E.MolEngine:
        if self._call_eng is not None:   # inside one_bind(): the parameters were checked when the call began
            return self._call_eng
        eng = self._engine_for_bind()
        if eng is not self._engine:  # first use, or the module's parameters changed
            self._engine = eng
            self._index = eng.build_index(self._item_embeddings[0])
        return eng

    def _engine_for_bind(self) -> E.MolEngine:
        return self._mol_module.engine()

    @contextlib.contextmanager
    def one_bind(self):
        """One look at the module's parameters for a whole call: `engine()` compares 2 x 48 (pointer, version) pairs, ~10 us, and a
        two-pass call asks for the engine six times -- host time that sits between the batches of a plain loop."""
        outer = self._call_eng
        if outer is None:
            self._call_eng = self._bind()
        try:
            yield
        finally:
            self._call_eng = outer

    def all_logits(self, query_embeddings: torch.Tensor, **kwargs) -> torch.Tensor:
        """(B, N) fp32 MoL logits against the whole corpus."""
        eng = self._bind()
        qpack, _, _ = eng.query_pack(query_embeddings, kwargs.get("user_ids"))
        return eng.score_dense(qpack, query_embeddings.size(0), self._index)

    def _all_logits_scratch(self, query_embeddings: torch.Tensor, **kwargs) -> torch.Tensor:
        """Same, into recycled internal buffers (the result is consumed by the top-k before the next call)."""
        eng = self._bind()
        B = query_embeddings.size(0)
        n_q = eng.lib.rails_mol_query_pack_floats(E.C.byref(eng.shape), B)
        qpack, _, _ = eng.query_pack(query_embeddings, kwargs.get("user_ids"), out=self._buf("qpack", n_q, torch.float32))
        logits = self._buf("logits", B * self._index.n_items, torch.float32).view(B, self._index.n_items)
        return eng.score_dense(qpack, B, self._index, out=logits)


    def _score_at(self, eng, qpack: torch.Tensor, batch: int, positions: torch.Tensor) -> torch.Tensor:
        """(B, K) full-MoL logits of per-row candidates given as VALID positions of this module's index.  fp32 engines read the
        candidates in place (rails_mol_score_indexed: no gathered copy, one launch; the same bits as gather + score_candidates);
        the f16 builds, which have no indexed instantiation, gather a per-row index of the candidates first."""
        K = positions.shape[1]
        if eng.score_indexed_supported(batch, K):
            rows = self._index_rows(eng)
            if rows is not None:           # the row-major copy: a candidate's bytes in whole cache lines (half the time of the tile-packed reads; same bits)
                return eng.score_indexed_rows(qpack, batch, rows, self._index.n_items, positions)
            return eng.score_indexed(qpack, batch, self._index, positions)   # any K: the kernel masks the ragged last tile
        cand, kp = eng.gather_index(self._index, positions)
        return eng.score_candidates(qpack, batch, cand, kp)[:, :K]


    RERANK_ROWS_COPY_MAX_BYTES = 8 << 30     # fp32 indexes up to this size get a row-major copy for the candidate re-scoring of the rerank paths (0: never)
    _rows_cache = None

    def _index_rows(self, eng) -> Optional[torch.Tensor]:
        """The row-major copy of this module's fp32 index (rails_mol_index_rows_build), built at the first rerank; None where it does not apply
        (f16 index formats, indexes beyond RERANK_ROWS_COPY_MAX_BYTES -- a 125 M-item shard -- or too little free memory)."""
        c = self._rows_cache
        if c is not None and c[0] is eng and c[1] is self._index:
            return c[2]
        rows = None
        need = self._index.buf.numel() * 4
        if getattr(eng, "precision", None) == "fp32" and self._index.buf.is_cuda and 0 < need <= self.RERANK_ROWS_COPY_MAX_BYTES:
            free, _ = torch.cuda.mem_get_info(self._index.buf.device)
            if free > 2 * need:
                rows = eng.build_index_rows(self._index)
        self._rows_cache = (eng, self._index, rows)
        return rows


class MoLBruteForceTopK(MoLTopKModule):
    def __init__(self, mol_module: MoLSimilarity, item_embeddings: torch.Tensor, item_ids: torch.Tensor, exact_mode: Optional[str] = None,
                 bound_kind_items: Optional[int] = None) -> None:
        """exact_mode (not in the reference's signature, rails/indexing/mol_top_k.py:84-97): "proved" | "dense", default EXACT_MODE.
        bound_kind_items: the corpus size the proved flow's size-dependent choices are made for (the item-sharded wrapper passes the shard size
        every rank computes alike); default: this module's own corpus."""
        self.bound_kind_items = bound_kind_items
        if exact_mode not in (None, "proved", "dense"):
            raise ValueError(f"exact_mode must be 'proved' or 'dense', got {exact_mode!r}")
        self._index32: Optional[E.MolIndex] = None          # precision "f16x3-exact": dense fp32 index (candidate gather, fallback)
        self._index32_engine = None
        self.keep_dense_fp32_index: Optional[bool] = self.KEEP_DENSE_FP32_INDEX
        self.rescore_stats = {"calls": 0, "fallbacks": 0, "audited": 0, "mismatches": 0}
        self._probe_pool: Optional[torch.Tensor] = None
        self._ok_host: Optional[torch.Tensor] = None
        self._recent: list = []       # verdicts of the last speculative calls
        self._calib_engine = None
        self._err_seen = 0.0          # largest |first pass - fp32| ever seen on re-scored candidates and probes (reset only with the engine)
        self._risk_pool: Optional[torch.Tensor] = None   # positions of the highest-norm items (probed every call)
        self._risk_rows: Optional[torch.Tensor] = None   # the probe rows the rotation over them produces, precomputed
        self.audit_every: int = int(self.AUDIT_EVERY)    # > 0: every n-th speculative call is also run on the dense fp32 path and compared
        self._audit_stream = None
        self._debug_first_pass_bias = None
        self._verdict_state: Optional[torch.Tensor] = None
        self._state_pending = None
        self._pad_scale = 1           # candidate margin multiplier, doubled when a verification fails
        self._state_direct = False    # True: the verdict state reaches the host through the finish kernel's own stores (no copy, no event)
        self._pause_left = 0
        self.exact_mode: str = exact_mode or self.EXACT_MODE
        self._proved_choice = None    # (fp32 engine the choice was made for, precision to bind or None)
        self._gate_guard_limit: Optional[float] = None
        self._ok_event = None
        self._probe_n = -1
        super().__init__(mol_module=mol_module, item_embeddings=item_embeddings, item_ids=item_ids)

    # ---- which arithmetic an exact top-k runs in -----------------------------------------------------------------------------------
    # A module whose MoL precision is the default (fp32) returns the fp32 kernels' bits.  "proved" (the default): the split-f16 kernels
    # pick the candidates, the fp32 kernels re-score them, and the a-priori bound of rails_amd/f16x3_bound.py on |first pass - fp32|
    # proves per call, on the device, that nothing outside the candidates can belong to the result -- the dense fp32 path's output bit for
    # bit at ~2.3 x its speed; a call that cannot be proved (crowded scores, a guard of the bound violated) is redone on the dense fp32
    # kernels behind the verdict.  "dense": the fp32 kernels over the whole corpus, always.  RAILS_EXACT_MODE overrides the default.
    # The proved mode needs both index formats resident (2 x the fp32 index bytes); corpora where that does not fit, corpora below
    # SPECULATE_MIN_ITEMS and modules whose bound is infinite (see the guards in f16x3_bound.py) run "dense".
    EXACT_MODE = __import__("os").environ.get("RAILS_EXACT_MODE", "proved")
    if EXACT_MODE not in ("proved", "dense"):
        raise ValueError(f"RAILS_EXACT_MODE must be 'proved' or 'dense', got {EXACT_MODE!r}")
    PROVED_MAX_EPS = 2.0          # a module whose a-priori bound exceeds this many logit units is not worth a second index: the items within eps of the
                                  # k-th score run into the tens of thousands (16x16x64: eps = 2.9; profiles/r05_proved_candidate_census.json)
    PROVED_MAX_EPS_PER_PAIR = 10.0 # ... up to this eps (at the a-priori |cl| <= 1/tau) the bound is applied PER PAIR instead: the pairs of a corpus sit at a third of
                                  # 1/tau and the bound is quadratic in it (_bound_kind "upper": 16x16x64 at random init, 3.0; trained weights of the other shapes).
                                  # Round 6 (tools/r06_scale_probe.py, amzn-books, pair-gate weights x s): x 3 (eps 8.6) proves 44 / 45 calls at 3 904 candidates,
                                  # 2.78 ms per batch against 6.4 dense; x 4 (eps 15.2) proves none even at 16 384 candidates -- hence 10, not 8
    PAD_ONE_EPS = (824, 3)        # candidates beyond k: max(floor, per_k * k) (doubled after a failed verdict) -- under one eps ...
    PAD_PER_PAIR = (1848, 1)      # ... and under per-pair upper bounds (sized for a 12.5 M-item shard of 16x16x64: 730-900 items can reach the 200-th score;
                                  # amzn-books at k' = 2 561: 5 152 candidates prove every call where one eps needs 10 272)
    # Corpora up to PER_PAIR_MAX_ITEMS items take the per-pair form whatever their one eps is: their first pass is short, so the UPPER build's extra
    # work costs microseconds, and the tighter bound proves with half the candidates (kc = 512 at k' = 200: a cheaper selection, half the re-scoring).
    # amzn-books shape, proved step per-pair / one eps: 20 k items 0.173 / 0.195 ms, 32 k 0.219 / 0.255, 65 k 0.350 / 0.368, 131 k 0.590 / 0.603
    # (695 k: 2.80 / 2.75 -- one eps wins there); ML-20M (27 278 items), which one eps cannot prove at all: 0.179 ms against 0.239 dense.
    PER_PAIR_MAX_ITEMS = 196608
    PAD_PER_PAIR_SMALL = (312, 1)
    bound_kind_items: Optional[int] = None
    # The default mode speculates where the first pass saves more than the verification costs (~70 us of launches): from B x N = 2^18 (query, item)
    # pairs on.  Same-box steps, proved flow / dense fp32 kernels (tools/r06_probe_p.sh): amzn-books 695 762 items B = 1 0.285 / 0.322 ms, 400 k items
    # B = 1 0.198 / 0.211, 200 k B = 1 0.131 / 0.125, B = 2 0.145 / 0.184; ML-20M (27 278 items) B = 2 0.100 / 0.057, B = 8 0.113 / 0.094, B = 16 0.113 / 0.146.
    PROVED_MIN_BATCH = 1
    PROVED_MIN_PAIRS = 1 << 18

    @classmethod
    def speculation_pays(cls, batch: int, n_items: int) -> bool:
        return batch >= cls.PROVED_MIN_BATCH and batch * n_items >= cls.PROVED_MIN_PAIRS

    def _engine_for_bind(self) -> E.MolEngine:
        mol = self._mol_module
        base = mol.engine()
        if self.exact_mode != "proved" or base.precision != "fp32" or base.exact is not None:
            return base
        if self._proved_choice is None or self._proved_choice[0] is not base:
            self._proved_choice = (base, "f16x3-exact" if self._proved_applies(base) else None)
        want = self._proved_choice[1]
        return mol.engine(want, _params_as_checked=True) if want else base     # (the second look at the same parameters in the same breath)

    def _proved_applies(self, base: E.MolEngine) -> bool:
        spec, N = base.spec, self._item_embeddings.shape[1]
        if N < self.SPECULATE_MIN_ITEMS or N > 0xFFFFFFFF or not self._item_embeddings.is_cuda:
            return False
        if not base.lib.rails_mol_shape_supported(E.C.byref(spec.to_c("f16x3"))):
            return False
        if self._bound_kind(spec, base.lib) is None:
            return False
        from . import arith_check

        if not arith_check.device_ok(self._item_embeddings.device):     # the bound's hypotheses about the part, re-measured on THIS device (once per process)
            return False
        if self.keep_dense_fp32_index is False:
            return False
        need = base.lib.rails_mol_index_floats(E.C.byref(base.shape), N) * 4
        free, _ = torch.cuda.mem_get_info(self._item_embeddings.device)
        held = self._index.buf.numel() * 4 if self._index is not None else 0     # a rebind: the old index is dropped first
        return free + held > 2 * need + min(32 * N * 4, self.MAX_LOGIT_BYTES) + (1 << 30)

    def _bound_from_weights(self, spec) -> Dict[str, float]:
        """rails_amd/f16x3_bound.py for this module's pair-gate weights; {"eps": inf} where a guard of the bound fails."""
        from . import f16x3_bound as FB

        if (not spec.dot_product_l2_norm or spec.gating_combination_type != "glu_silu" or spec.gating_qi_hidden_dim <= 0
                or not (spec.gating_query_fn and spec.gating_item_fn)):
            return {"eps": math.inf}
        g = self._mol_module._gating_fn._qi_partial_module
        lin = [m for m in g.modules() if isinstance(m, torch.nn.Linear)]
        if len(lin) != 2:
            return {"eps": math.inf}
        zeros = lambda n: torch.zeros(n)
        b1 = lin[0].bias if lin[0].bias is not None else zeros(lin[0].out_features)
        b2 = lin[1].bias if lin[1].bias is not None else zeros(lin[1].out_features)
        return FB.first_pass_bound(lin[0].weight, b1, lin[1].weight, b2, spec.temperature, spec.dot_product_dimension,
                                   spec.query_dot_product_groups, spec.item_dot_product_groups)

    def _bound_kind(self, spec, lib) -> Optional[str]:
        """How the proved flow bounds |first pass - fp32| for this shape and these weights:
          "eps"    one a-priori eps for every pair (f16x3_bound.first_pass_bound), where it is at most PROVED_MAX_EPS;
          "upper"  a per-pair bound, quadratic in the pair's largest |cross logit|, added to the first-pass logit by the kernel itself
                   (f16x3_bound.upper_bound_poly, rails_mol_score_dense_upper): where one eps is too coarse and the shape has the kernel;
          None     neither (infinite bound, or too coarse without the kernel): the module runs the dense fp32 kernels."""
        eps = self._bound_from_weights(spec).get("eps", math.inf)
        small = self._policy_items() <= self.PER_PAIR_MAX_ITEMS
        if (eps > self.PROVED_MAX_EPS or small) and eps <= self.PROVED_MAX_EPS_PER_PAIR and lib.rails_mol_score_dense_upper_supported(E.C.byref(spec.to_c("f16x3"))):
            return "upper"
        if eps <= self.PROVED_MAX_EPS:
            return "eps"
        return None

    def _policy_items(self) -> int:
        """The corpus size the size-dependent choices of the proved flow are made for: this module's own, or -- set by the item-sharded wrapper,
        the same on every rank -- the shard size (ranks must agree on the form of the bound)."""
        return int(self.bound_kind_items or self._item_embeddings.shape[1])

    def _per_pair_pad(self) -> Tuple[int, int]:
        return self.PAD_PER_PAIR_SMALL if self._policy_items() <= self.PER_PAIR_MAX_ITEMS else self.PAD_PER_PAIR

    def _upper_poly(self, k: Optional[int] = None) -> Optional[Tuple[float, float, float]]:
        """(ub2, ub1, ub0) when the bound engine's first pass writes per-pair UPPER BOUNDS of the fp32 logits (_bound_kind "upper"), else None.
        With k: also for a CALL of an engine whose form is the one eps, when the call wants PER_PAIR_MIN_K results or more -- the candidates a
        large k needs under one eps (k' = 2 561 on amzn-books: 10 272) cost more to re-score and sort than the UPPER build adds to the first pass
        (3.19 -> 2.9 ms per batch); the verdict of such a call runs with eps = 0 on the same state."""
        eng = self._engine
        c = self._upper_poly_cache
        if c is None or c[0] is not eng:
            c = self._upper_poly_fill(eng)
        if c[1] is not None or k is None or k < self.PER_PAIR_MIN_K:
            return c[1]
        return c[2]

    def _upper_poly_fill(self, eng):
        """-> (engine, the polynomial if the engine's form is per-pair, the polynomial if the shape has the UPPER build at all)"""
        poly = any_poly = None
        kind = self._bound_kind(eng.spec, eng.lib) if (eng.exact is not None and eng.dense_precision == "f16x3") else None
        if kind is not None and eng.score_dense_upper_supported() and self._bound_from_weights(eng.spec).get("eps", math.inf) <= self.PROVED_MAX_EPS_PER_PAIR:
            from . import f16x3_bound as FB

            lin = [m for m in self._mol_module._gating_fn._qi_partial_module.modules() if isinstance(m, torch.nn.Linear)]
            sp = eng.spec
            zeros = lambda n: torch.zeros(n)      # noqa: E731
            res = FB.upper_bound_poly(lin[0].weight, lin[0].bias if lin[0].bias is not None else zeros(lin[0].out_features), lin[1].weight,
                                      lin[1].bias if lin[1].bias is not None else zeros(lin[1].out_features), sp.temperature, sp.dot_product_dimension,
                                      sp.query_dot_product_groups, sp.item_dot_product_groups)
            any_poly = res["poly"]
            self._upper_poly_info = res
            if kind == "upper":
                poly = any_poly
        self._upper_poly_cache = (eng, poly, any_poly)
        return self._upper_poly_cache

    PER_PAIR_MIN_K = 1024         # calls for at least this many results take per-pair bounds whatever the engine's form (see _upper_poly)
    _upper_poly_cache = None
    _upper_poly_info = None

    def all_logits(self, query_embeddings: torch.Tensor, **kwargs) -> torch.Tensor:
        """(B, N) fp32 MoL logits against the whole corpus -- of the module's OWN precision (the proved mode's internal split-f16
        engine is not the module's precision: its fp32 companion answers)."""
        eng = self._bind()
        if eng.exact is not None and self._mol_module.engine() is not eng:
            ex = eng.exact
            qpack, _, _ = ex.query_pack(query_embeddings, kwargs.get("user_ids"))
            return ex.score_dense(qpack, query_embeddings.size(0), self._dense_fp32_index())
        return super().all_logits(query_embeddings, **kwargs)

    def forward(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        eng = self._bind()
        if eng.exact is not None:
            return self._forward_rescored(query_embeddings, k, **kwargs)
        B, N = query_embeddings.size(0), self._index.n_items
        if B * N * 4 > self.MAX_LOGIT_BYTES and k <= self.CHUNK_ITEMS:
            return self._forward_chunked(query_embeddings, k, **kwargs)
        logits = self._all_logits_scratch(query_embeddings, **kwargs)
        ws = self._buf("topk_ws", E._lib.load().rails_topk_workspace_bytes(logits.shape[0], logits.shape[1], k), torch.uint8)
        scores, ids = E.topk(logits, k, ids=self._ids_flat, sorted=sorted, workspace=ws)
        return scores.to(query_embeddings.dtype), ids

    def forward_filtered(self, query_embeddings: torch.Tensor, k_prime: int, invalid_ids: torch.Tensor, k: int, **kwargs):
        """CandidateIndex.get_top_k_outputs' body for this module: top-k' + id map + seen-id filter with the filter fused into the final
        selection launch (rails_topk_filtered on the dense logits) -> (top_k_ids (B, k), top_k_scores (B, k)),
        or None when the sizes / the precision route are outside the fused path (the caller then composes forward +
        filter_seen_ids: same bits)."""
        eng = self._bind()
        B, N = query_embeddings.size(0), self._index.n_items
        if B * N * 4 > self.MAX_LOGIT_BYTES or not E.topk_filter_fusable(N, k_prime, invalid_ids.shape[1], k):
            return None
        if eng.exact is not None:
            # the default mode's small calls (speculation_pays) run the dense fp32 kernels (_forward_rescored): keep the filter fused into their
            # selection launch as the plain fp32 module does
            ex = eng.exact
            if not (not self.speculation_pays(B, N) and self._mol_module.engine() is not eng and eng.dense_precision == "f16x3"
                    and self._index32 is not None and self._index32_engine is ex):
                if not (self.FUSED_TAIL and eng.dense_precision == "f16x3" and k <= k_prime <= N):
                    return None
                # the proved flow: the filter runs inside its finish launch (and inside the redo's selection)
                r = self._forward_rescored(query_embeddings, k_prime, _seen=(invalid_ids, k), **kwargs)
                if r[0] == "filtered":
                    return r[1], r[2]
                return E.filter_seen_ids(r[1], r[0], invalid_ids, k)
            n_q = ex.lib.rails_mol_query_pack_floats(E.C.byref(ex.shape), B)
            qpack32, _, _ = ex.query_pack(query_embeddings, kwargs.get("user_ids"), out=self._buf("qpack32", n_q, torch.float32))
            logits = ex.score_dense(qpack32, B, self._index32, out=self._buf("logits", B * N, torch.float32).view(B, N))
            ws = self._buf("topk_ws", E._lib.load().rails_topk_workspace_bytes(B, N, k_prime), torch.uint8)
            ids, scores = E.topk_filtered(logits, k_prime, self._ids_flat, invalid_ids, k, workspace=ws)
            return ids, scores.to(query_embeddings.dtype)
        logits = self._all_logits_scratch(query_embeddings, **kwargs)
        ws = self._buf("topk_ws", E._lib.load().rails_topk_workspace_bytes(B, N, k_prime), torch.uint8)
        ids, scores = E.topk_filtered(logits, k_prime, self._ids_flat, invalid_ids, k, workspace=ws)
        return ids, scores.to(query_embeddings.dtype)

    MAX_LOGIT_BYTES = 4 << 30      # larger (B, N) logit matrices are never materialised: the corpus is scored in chunks
    CHUNK_ITEMS = 1 << 23          # 8 Mi items per chunk (a multiple of the tile): 1 GiB of logits at B = 32

    def _forward_chunked(self, query_embeddings: torch.Tensor, k: int, _engine=None, _index=None, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """Exact top-k of a corpus whose (B, N) logits would not fit comfortably (a 125 M-item shard at B = 32 is 16 GB): score
        CHUNK_ITEMS at a time into one recycled buffer, keep each chunk's top-k, merge.  Same result as the one-pass path bit for
        bit: chunks are position ranges in order and every list is sorted (score desc, position asc), so the final top-k over
        the chunk-major concatenation breaks ties by position as well (the item-sharded merge's argument, rails_amd/sharded.py)."""
        eng = _engine if _engine is not None else self._bind()
        index = _index if _index is not None else self._index
        B, N, C = query_embeddings.size(0), index.n_items, self.CHUNK_ITEMS
        n_q = eng.lib.rails_mol_query_pack_floats(E.C.byref(eng.shape), B)
        qpack, _, _ = eng.query_pack(query_embeddings, kwargs.get("user_ids"), out=self._buf("qpack", n_q, torch.float32))
        buf = self._buf("logits_chunk", B * min(C, N), torch.float32)
        ws = self._buf("topk_ws", E._lib.load().rails_topk_workspace_bytes(B, min(C, N), min(k, C)), torch.uint8)
        part_s, part_p = [], []
        for lo in range(0, N, C):
            n = min(C, N - lo)
            logits = eng.score_dense(qpack, B, index.items(lo, lo + n), out=buf[: B * n].view(B, n))
            s, p = E.topk(logits, min(k, n), workspace=ws)
            part_s.append(s)
            part_p.append(p + lo)
        scores, pos = E.topk(torch.cat(part_s, 1), k, ids=torch.cat(part_p, 1))
        return scores.to(query_embeddings.dtype), self._ids_flat[pos]

    # ---- precision "f16x3-exact": speculate with the f16x3 kernels, verify in fp32 ---------------------------------------
    KEEP_DENSE_FP32_INDEX: Optional[bool] = None   # None: when memory allows; True / False: always / never (candidates' rows are rebuilt)
    RESCORE_EPS_PER_INV_TEMPERATURE = 5e-5   # eps = this / temperature: 1e-3 on logits in [-20, 20], 30 x the largest
                                             # |f16x3 - fp32| seen over 22 M pairs (profiles/r02_bench.json fast_path)
    # first pass on the one-product f16 kernels ("f16-exact"): |s16 - s32| up to 4.3e-2 on amzn-books, 1.7e-2 on ML-20M
    # (tools/single_f16_probe.py) -> default eps = 0.15 on logits in [-20, 20], twice the candidate margin of the f16x3 first pass
    RESCORE_EPS_PER_INV_TEMPERATURE_F16X1 = 7.5e-3

    def _forward_rescored(self, query_embeddings: torch.Tensor, k: int, _seen=None, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """The fp32 brute-force result -- same scores, same ids, same tie order -- at the f16x3 kernel's speed.
          1. f16x3 logits s16 over the whole index; the top Kc = k + max(64, k/4) of them (rounded up to whole tiles) are the
             candidates, m = the smallest candidate's s16.  Every other item has s16 <= m.
          2. the candidates are gathered from the dense fp32 index (kept next to the f16x3 one when memory allows; otherwise
             their raw rows go through the fp32 index build) and scored by the fp32 kernel: e32, their exact fp32 logits (the
             same arithmetic per (query, item) pair as the dense fp32 path, hence the same bits).
          3. rails_rescore_select: top-k of e32 by (score desc, corpus position asc) -- the dense path's total order.
        The result is the dense fp32 top-k iff no item outside the candidates can reach the k-th exact score e_k, i.e. iff
        e_k > m + eps where |s16 - s32| <= eps.  Step 3 returns e_k - m per row and the largest |e32 - s16| over the row's Kc
        candidates and 32 more items drawn at random from the whole corpus; eps is the calibrated default or SAFETY x the largest
        such error this module has seen, whichever is larger.  A row that does not clear eps -> the call is redone on the dense
        fp32 index (and later calls take more candidates).  One (B x 8)-byte device-to-host copy per call."""
        eng = self._bind()
        ex = eng.exact
        self._absorb_state()
        B, N = query_embeddings.size(0), self._index.n_items
        if k > N:
            raise RuntimeError(f"selected index k out of range (k={k}, n={N})")
        single = eng.dense_precision == "f16x1"
        # The bound on |first pass - fp32|: A PRIORI for the f16x3 first pass (rails_amd/f16x3_bound.py: the call is then PROVED to
        # return the dense fp32 result whenever its verdict clears); monitored and empirical for the one-product first pass, whose a-priori
        # bound is vacuous.  A module whose a-priori bound is infinite (a guard fails) does not speculate.
        eps_proved = None if single else self._proved_eps()
        upper = None if single else self._upper_poly(k)         # per-pair upper bounds instead of one eps (then the verdict's eps is 0)
        if upper is not None and eps_proved is not None and math.isfinite(eps_proved) and self._upper_poly() is None:
            eps_proved = 0.0          # a large-k call of an engine whose own form is the one eps
        if eps_proved is not None and not math.isfinite(eps_proved):
            self.rescore_stats["unprovable_calls"] = self.rescore_stats.get("unprovable_calls", 0) + 1
            return self._forward_fp32_dense(query_embeddings, k, **kwargs)
        if eps_proved is not None and not self.speculation_pays(query_embeddings.size(0), self._index.n_items) and self._mol_module.engine() is not eng:
            # too few (query, item) pairs for the first pass to save what the verification costs (PROVED_MIN_PAIRS) -- the default mode takes
            # the dense kernels there (an explicit "f16x3-exact" precision keeps speculating)
            return self._forward_fp32_dense(query_embeddings, k, **kwargs)
        if eps_proved is not None:
            # candidates: every item within eps of the k-th score must be among them.  amzn-books, eps = 0.9-1.0: 470-680 items at k = 200,
            # 6 000-7 500 at k = 2 561 (128 queries; profiles/r05_proved_candidate_census.json); rails_topk costs the same 80-90 us from
            # 544 to 1 536 candidates per row of 700 k scores, so the margin starts generous.  A failed verdict doubles it
            # (per-pair upper bounds on a 12.5 M-item shard of 16x16x64: 730-900 items can reach the 200-th score; tools/r05_c4_census.py)
            floor, per_k = self._per_pair_pad() if upper is not None else self.PAD_ONE_EPS
            pad = max(floor, per_k * k) * self._pad_scale
            kc = min((k + pad + E.TILE_ITEMS - 1) // E.TILE_ITEMS * E.TILE_ITEMS, 16384)
        else:
            pad = (max(128, k // 2) if single else max(64, k // 4)) * self._pad_scale
            kc = (k + pad + E.TILE_ITEMS - 1) // E.TILE_ITEMS * E.TILE_ITEMS
            if k <= 384:
                kc = min(kc, 512)       # rails_topk's two-launch path ends at k = 512; beyond it a selection costs five reads of the logits
        oversize = B * N * 4 > self.MAX_LOGIT_BYTES      # the 4 GiB logit policy comes first: no route below may materialise (B, N)
        if kc >= N or k == 0 or kc > 16384 or k + E.TILE_ITEMS > kc or N > 0xFFFFFFFF or N < self.SPECULATE_MIN_ITEMS or self._speculation_paused():
            return self._forward_fp32_dense(query_embeddings, k, **kwargs)
        if oversize:                              # the speculative pass wants the whole (B, N) s16 matrix
            rows = self.MAX_LOGIT_BYTES // (N * 4)
            if rows >= 1:                         # ... of a slice of the batch at a time (per-row payloads are sliced with it)
                parts = []
                for b0 in range(0, B, rows):
                    kw = {key: (v[b0 : b0 + rows] if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B else v) for key, v in kwargs.items()}
                    parts.append(self._forward_rescored(query_embeddings[b0 : b0 + rows], k, **kw))
                return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)
            return self._forward_fp32_dense(query_embeddings, k, **kwargs)   # one row is too long: fp32, in corpus chunks
        if (eps_proved is not None and self.FUSED_TAIL and self.DEVICE_VERDICT and self._index32 is not None and self._index32_engine is ex):
            return self._forward_proved(query_embeddings, k, kc, eps_proved, upper, _seen, **kwargs)
        # one prologue writes the query pack in both formats: f16 hi/lo for the first pass, fp32 for the re-scoring
        n_q = eng.lib.rails_mol_query_pack_floats(E.C.byref(eng.shape), B)
        qpack16, qpack32 = eng.query_pack_both(query_embeddings, kwargs.get("user_ids"), self._buf("qpack", n_q, torch.float32),
                                               self._buf("qpack32", n_q, torch.float32))
        s16 = self._buf("logits", B * N, torch.float32).view(B, N)
        ws = self._buf("topk_ws", E._lib.load().rails_topk_workspace_bytes(B, N, kc), torch.uint8)
        hook = self._first_pass_hook        # measurement only (bench.py: events around the dominant launch, on its stream)
        if hook is not None:
            hook(0)
        if upper is not None:
            eng.score_dense_upper(qpack16, B, self._index, upper, out=s16)
        else:
            eng.score_dense(qpack16, B, self._index, out=s16)
        if hook is not None:
            hook(1)
        if self._debug_first_pass_bias is not None:   # tests only: (positions, delta) -- the first pass is made to under-score these items
            s16[:, self._debug_first_pass_bias[0]] -= self._debug_first_pass_bias[1]
        c16, pos = E.topk(s16, kc, workspace=ws)
        # two more tiles per query of probes (random + highest-norm items), re-scored too, so that the MONITORED bound |s16 - s32| <= eps is
        # watched outside the candidates as well (the a-priori bound needs no watching: the candidates' own errors are still compared
        # with it, and one above it is reported as a violation of the arithmetic model)
        if eps_proved is None:
            pos = torch.cat([pos, *self._probes(B, N)], dim=1)
        if self._index32 is not None and B * pos.shape[1] <= self.INDEXED_MAX_CANDIDATES and ex.score_indexed_supported(B, pos.shape[1]):
            # (_rescore_candidates below is this branch + the gather fallback, for the sharded flow)
            # read the candidates in place from the fp32 index: one launch less and no gathered copy.  A candidate's 1 280 bytes are 80
            # pieces of 16 bytes in the tile-packed index, each in its own cache line, whoever fetches them -- the gather kernel paid
            # that amplification AND wrote and re-read the copy.  Measured with the GEMM1 lookahead of the independent-wave kernels in
            # (tools/indexed_rescore_probe.py, amzn-books, `f16-exact`): k' = 200  B = 8 / 32 / 128: 0.501 / 1.59 / 5.81 -> 0.494 /
            # 1.59 / 5.76 ms; k' = 2561: 0.631 / 1.93 / 6.88 -> 0.610 / 1.835 / 6.55 ms.  (Before the lookahead the in-place reads cost
            # more than the copy beyond B x Kc = 1024 candidates: B = 32 1.65 -> 1.69 ms; INDEXED_MAX_CANDIDATES keeps the switch.)
            e32 = self._rescore(ex, qpack32, B, pos)
        else:
            if self._index32 is not None:
                cand, _ = ex.gather_index(self._index32, pos)
            else:
                cand = ex.build_index(self._item_embeddings[0].index_select(0, pos.reshape(-1)))
            e32 = ex.score_candidates(qpack32, B, cand, pos.shape[1])
        scores, ids, _, stats = E.rescore_select(e32, c16, pos, self._ids_flat, N, k, approx_dense=s16, one_sided=upper is not None)
        self.rescore_stats["calls"] += 1
        self.rescore_stats["kc"] = kc
        # The bound eps on |s16 - s32|: never below the calibrated default, and SAFETY x the largest error this module has seen on its
        # candidates and probes (this call included) -- a model whose weights make the first pass coarser widens its own margin
        # instead of failing the monitor forever.  The row passes when its k-th exact score clears the best non-candidate by eps.
        default = (self.RESCORE_EPS_PER_INV_TEMPERATURE_F16X1 if single else self.RESCORE_EPS_PER_INV_TEMPERATURE) / eng.spec.temperature
        safety = self.SAFETY_F16X1 if single else self.SAFETY_F16X3
        guard, guard_limit = None, 0.0
        if eps_proved is not None:
            # eps = the a-priori bound (safety 1: an observed error above it -- a violation of the model -- still widens the margin); the
            # one data-dependent hypothesis of the bound, max |gq'| max |gi| <= gate_guard, is checked by the verdict kernel on the
            # batch's gq' rows (behind the Eq fragments in the fp32 query pack)
            default, safety = eps_proved, 1.0
            sp = eng.spec
            off = (B + 32 // sp.query_dot_product_groups - 1) // (32 // sp.query_dot_product_groups) * 32 * sp.dot_product_dimension
            guard, guard_limit = qpack32[off : off + B * sp.num_logits], self._gate_guard_limit
        if self._index32 is not None and self._index32_engine is ex and self.DEVICE_VERDICT:
            # Verdict and fallback ON THE DEVICE: rails_rescore_verdict folds the row stats into the calibration state and writes the
            # REDO flag; the dense fp32 pass and its top-k are enqueued behind it under that flag as their launch predicate (no-ops
            # unless the verification failed) and overwrite (scores, ids).  The host never waits; it looks at a snapshot of the
            # state when the NEXT call starts (statistics, candidate margin, pause logic).
            state = self._state()
            E.rescore_verdict(stats, state, default, safety, guard, guard_limit)
            redo = state.view(torch.int32)[1:2]
            l32 = ex.score_dense(qpack32, B, self._index32, out=self._buf("logits", B * N, torch.float32).view(B, N), run_if=redo)
            E.topk(l32, k, ids=self._ids_flat, workspace=ws, out=(scores, ids), run_if=redo)
            self._state_host.copy_(state, non_blocking=True)
            self._state_event.record()
            self._state_pending = (k, kc)
            self._state_direct = False
        else:
            err, gap = self._read_stats(stats)
            if err == err and err != float("inf"):
                self._err_seen = max(err, self._err_seen)   # never forgotten: a rare outlier keeps the margin wide until the engine changes
            eps = max(default, safety * self._err_seen)
            good = err == err and err != float("inf") and gap > eps
            if guard is not None and good:      # the bound's data-dependent guard, on the host in this (index-less) variant
                gmax = float(guard.abs().max())
                self.rescore_stats["guard_max"] = max(self.rescore_stats.get("guard_max", 0.0), gmax)
                good = gmax <= guard_limit
            if eps_proved is not None:
                self._count_proved(1 if good else 0)
            self.rescore_stats["eps"] = eps
            self._note_verdict(good, k, kc)
            if not good:
                return self._forward_fp32_dense(query_embeddings, k, **kwargs)
        if self.audit_every > 0 and self.rescore_stats["calls"] % self.audit_every == 0:
            self._audit(query_embeddings, k, scores, ids, **kwargs)
        return scores.to(query_embeddings.dtype), ids

    # ---- the proved flow with the fused tail (round 6) ----------------------------------------------------------------------------
    # first pass -> rails_candidates_select (threshold selection: one histogram + one compaction launch, one launch for short rows) ->
    # fp32 re-scoring of the counted candidates -> rails_candidates_finish (sort, top-k, verdict, seen-id filter, calibration state written
    # to the device AND straight into pinned host memory) -> the dense redo under the verdict's launch predicate: 8-9 launches where the
    # exact-kc selection + separate verdict / filter / state copy took 19 (amzn-books B = 32: 0.19 -> ~0.1 ms behind the first pass).
    FUSED_TAIL = __import__("os").environ.get("RAILS_FUSED_TAIL", "1") != "0"
    _cand = None          # {(B, cap): [workspace, positions, first-pass scores, fp32 scores]}
    _cand_dirty = False

    def _cand_buffers(self, B: int, cap: int):
        """(workspace, positions, first-pass scores, fp32 scores) of a (batch, candidate cap): zero-initialised once (the workspace must be; stale
        positions past a row's count must be valid), kept for the last few shapes (callers that alternate batch sizes do not reallocate)."""
        pool = self._cand
        if pool is None:
            pool = self._cand = {}
        c = pool.get((B, cap))
        if c is None:
            if len(pool) >= 4:
                pool.pop(next(iter(pool)))
            dev = self._item_embeddings.device
            c = pool[(B, cap)] = [E.candidates_workspace(B, dev), torch.zeros((B, cap), dtype=torch.int64, device=dev),
                                  torch.zeros((B, cap), dtype=torch.float32, device=dev), torch.zeros((B, cap), dtype=torch.float32, device=dev)]
        elif self._cand_dirty:       # an exception between select and finish left counts / histograms behind
            for v in pool.values():
                v[0].zero_()
        self._cand_dirty = False
        return c[0], c[1], c[2], c[3]

    def _score_range(self, upper) -> Tuple[float, float]:
        """The a-priori range of the first-pass logits (the histogram's bins): a softmax mixture of cross logits in [-1/tau, 1/tau] (l2-normalised
        components: a guard of the bound) plus, for the UPPER builds, the per-pair bound at the largest |cross logit|.  Scores outside are clamped
        into the end bins -- any monotone bin function is valid, the range only sets the resolution."""
        c = 1.02 / float(self._engine.spec.temperature)
        hi = c + ((upper[0] * c + upper[1]) * c + upper[2] if upper is not None else 0.0)
        return -c, hi

    def _forward_proved(self, query_embeddings: torch.Tensor, k: int, kc: int, eps_proved: float, upper, seen, **kwargs):
        eng = self._engine
        ex = eng.exact
        B, N = query_embeddings.size(0), self._index.n_items
        sp = eng.spec
        n_q = eng.lib.rails_mol_query_pack_floats(E.C.byref(eng.shape), B)
        qpack16, qpack32 = eng.query_pack_both(query_embeddings, kwargs.get("user_ids"), self._buf("qpack", n_q, torch.float32),
                                               self._buf("qpack32", n_q, torch.float32))
        s16 = self._buf("logits", B * N, torch.float32).view(B, N)
        hook = self._first_pass_hook        # measurement only (bench.py: events around the dominant launch, on its stream)
        if hook is not None:
            hook(0)
        if upper is not None:
            eng.score_dense_upper(qpack16, B, self._index, upper, out=s16)
        else:
            eng.score_dense(qpack16, B, self._index, out=s16)
        if hook is not None:
            hook(1)
        if self._debug_first_pass_bias is not None:   # tests only: (positions, delta) -- the first pass is made to under-score these items
            s16[:, self._debug_first_pass_bias[0]] -= self._debug_first_pass_bias[1]
        cap = min(kc, N)
        ws, pos, a16, e32 = self._cand_buffers(B, cap)
        self._cand_dirty = True
        lo, hi = self._score_range(upper)
        E.candidates_select(s16, cap, lo, hi, ws, pos, a16)
        if self._rows32 is not None and ex.score_indexed_supported(B, cap):
            ex.score_indexed_rows(qpack32, B, self._rows32, N, pos, counts=ws, out=e32)
        else:
            e32 = self._rescore(ex, qpack32, B, pos)       # every slot (the slots past a row's count hold earlier candidates: valid positions, ignored below)
        off = (B + 32 // sp.query_dot_product_groups - 1) // (32 // sp.query_dot_product_groups) * 32 * sp.dot_product_dimension
        guard = qpack32[off : off + B * sp.num_logits]
        state = self._state()
        fuse = seen is not None and k <= 512 and seen[0].shape[1] <= 256 and E.topk_filter_fusable(N, k, seen[0].shape[1], seen[1])
        scores, ids, f_i, f_s = E.candidates_finish(e32, a16, pos, cap, ws, self._ids_flat, N, k, eps_proved, 1.0, upper is not None, guard, sp.num_logits,
                                                    self._gate_guard_limit, state, self._state_host, seen if fuse else None)
        self._cand_dirty = False
        self.rescore_stats["calls"] += 1
        self.rescore_stats["kc"] = kc
        # the redo: the dense fp32 kernels behind the verdict, no-ops unless it failed (the host never waits)
        redo = state.view(torch.int32)[1:2]
        l32 = ex.score_dense(qpack32, B, self._index32, out=s16, run_if=redo)
        tws = self._buf("topk_ws", E._lib.load().rails_topk_workspace_bytes(B, N, k), torch.uint8)
        if fuse:
            E.topk_filtered(l32, k, self._ids_flat, seen[0], seen[1], workspace=tws, out=(f_i, f_s), run_if=redo)
        else:
            E.topk(l32, k, ids=self._ids_flat, workspace=tws, out=(scores, ids), run_if=redo)
        self._state_pending = (k, kc)
        self._state_direct = True
        if fuse:
            return "filtered", f_i, f_s.to(query_embeddings.dtype)
        if self.audit_every > 0 and self.rescore_stats["calls"] % self.audit_every == 0:
            self._audit(query_embeddings, k, scores, ids, **kwargs)
        return scores.to(query_embeddings.dtype), ids

    # ---- the proved flow split for an item-sharded corpus (rails_amd/sharded.py) --------------------------------------------------
    def shard_can_speculate(self) -> bool:
        """True iff this (local) module is bound in proved mode with both index formats resident: what ShardedMoLBruteForceTopK needs from
        EVERY rank before it runs the global proof."""
        eng = self._bind()
        return eng.exact is not None and eng.dense_precision == "f16x3" and self._index32 is not None and self._index32_engine is eng.exact \
            and self._proved_eps() is not None and math.isfinite(self._proved_eps())

    def speculate_for_shard(self, query_embeddings: torch.Tensor, k: int, kc: int, **kwargs):
        """The proved flow on THIS shard without a verdict: first pass over the shard, threshold selection of at most kc candidates by first-pass
        score, fp32 re-scoring, the best min(k, #candidates) by (fp32 score, position).
        -> (msg (B, 2k + 2) int64: [k score words | k ids | m | err] per row, the fp32 query pack): m = the smallest first-pass score among the
        candidates -- every item outside them scores below it (-inf when the whole shard is a candidate, +inf when nothing could be selected) --,
        err = the largest |fp32 - first pass| over the row's candidates (inf: a NaN).  The caller proves globally, after ONE all-gather of the
        messages: every item of every shard outside the candidates has s16 <= max over ranks of m, so the merged fp32 top-k is the dense one iff
        its k-th score exceeds that by eps (rails_merge_candidates_verdict)."""
        eng = self._bind()
        ex = eng.exact
        B, N = query_embeddings.size(0), self._index.n_items
        dev = query_embeddings.device
        n_q = eng.lib.rails_mol_query_pack_floats(E.C.byref(eng.shape), B)
        # packs of their own: with submit / result pipelining the verdict of batch i reads its gate rows while batch i + 1's prologue runs
        qpack16 = self._buf("qpack", n_q, torch.float32)
        qpack32 = torch.empty(n_q, dtype=torch.float32, device=dev)
        eng.query_pack_both(query_embeddings, kwargs.get("user_ids"), qpack16, qpack32)
        msg = torch.empty((B, 2 * k + 2), dtype=torch.int64, device=dev)
        if N == 0:      # an empty shard still takes part in the exchange: nothing to offer, nothing left outside
            msg[:, :k] = int(torch.tensor(float("-inf")).view(torch.int32)) & 0xFFFFFFFF
            msg[:, k : 2 * k] = -1
            msg[:, 2 * k] = int(torch.tensor(float("-inf")).view(torch.int32)) & 0xFFFFFFFF
            msg[:, 2 * k + 1] = 0
            return msg, qpack32
        cap = min(max(kc, 1), N)
        s16 = self._buf("logits", B * N, torch.float32).view(B, N)
        hook = self._first_pass_hook
        if hook is not None:
            hook(0)
        upper = self._upper_poly()
        if upper is not None:
            eng.score_dense_upper(qpack16, B, self._index, upper, out=s16)
        else:
            eng.score_dense(qpack16, B, self._index, out=s16)
        if hook is not None:
            hook(1)
        ws, pos, a16, e32 = self._cand_buffers(B, cap)
        self._cand_dirty = True
        lo, hi = self._score_range(upper)
        E.candidates_select(s16, cap, lo, hi, ws, pos, a16)
        if self._rows32 is not None and ex.score_indexed_supported(B, cap):
            ex.score_indexed_rows(qpack32, B, self._rows32, N, pos, counts=ws, out=e32)
        else:
            e32 = self._rescore(ex, qpack32, B, pos)
        E.candidates_finish(e32, a16, pos, cap, ws, self._ids_flat, N, k, 0.0, 1.0, upper is not None, None, 0, 0.0, None, None, msg=msg)
        self._cand_dirty = False
        self.rescore_stats["calls"] += 1
        self.rescore_stats["kc"] = kc
        return msg, qpack32

    ROWS_COPY_MAX_BYTES = 8 << 30      # the row-major copy of the fp32 index is kept for indexes up to this size (0: never)
    _rows32 = None

    def _rescore(self, ex: E.MolEngine, qpack32: torch.Tensor, B: int, pos: torch.Tensor) -> torch.Tensor:
        """fp32 logits of per-row candidates `pos` (positions of the resident fp32 index), whichever way is cheapest here -- same bits each way:
        in place from the row-major copy, in place from the tile-packed index, or gathered tiles (the 256-logit team kernel)."""
        if ex.score_indexed_supported(B, pos.shape[1]):
            if self._rows32 is not None:
                return ex.score_indexed_rows(qpack32, B, self._rows32, self._index32.n_items, pos)
            return ex.score_indexed(qpack32, B, self._index32, pos)
        cand, _ = ex.gather_index(self._index32, pos)
        return ex.score_candidates(qpack32, B, cand, pos.shape[1])

    INDEXED_MAX_CANDIDATES = 1 << 30   # rails_mol_score_indexed instead of gather + score_candidates up to this many (B x Kc) candidates (was 1024)
    DEVICE_VERDICT = True     # False: the host reads the verdict (one event spin per call) -- kept for deployments without a resident fp32 index

    def _note_verdict(self, good: bool, k: int, kc: int) -> None:
        self._recent.append(good)
        if not good:
            self.rescore_stats["fallbacks"] += 1
            if self._in_proved_mode():
                if kc < 16384:
                    self._pad_scale *= 2      # proved mode: the candidates must cover everything within eps of the k-th score
            elif self._pad_scale < 4 and not (k <= 384 and kc >= 512):
                self._pad_scale *= 2          # crowded scores or a coarse first pass: more candidates from the next call on

    def _state(self) -> torch.Tensor:
        if self._verdict_state is None:
            dev = self._item_embeddings.device
            self._verdict_state = torch.zeros(8, dtype=torch.float32, device=dev)
            self._state_host = torch.zeros(8, dtype=torch.float32).pin_memory()
            self._state_event = torch.cuda.Event()
            self._state_pending = None
            self._state_seen = (0.0, 0.0)     # (calls, redone calls) already folded into rescore_stats
        return self._verdict_state

    def _absorb_state(self, wait: bool = False) -> None:
        """Fold the device verdict of the PREVIOUS call(s) into rescore_stats / the candidate margin / the pause logic.  Non-blocking
        unless `wait` (stats()): a snapshot that has not landed yet is picked up by a later call."""
        if self._verdict_state is None or self._state_pending is None:
            return
        if self._state_direct:
            # the finish kernel writes the state into the pinned host words itself, the call counter last: a snapshot is whole when the
            # counter reads the same before and after it
            if wait:
                torch.cuda.current_stream(self._verdict_state.device).synchronize()
            c0 = float(self._state_host[5])
            h = self._state_host.tolist()
            if h[5] != c0 or (h[5] <= self._state_seen[0] and not wait):
                return
        else:
            if wait:
                self._state_event.synchronize()
            elif not self._state_event.query():
                return
            h = self._state_host
        calls, redone = float(h[5]), float(h[6])
        new_calls = max(0, int(calls - self._state_seen[0]))
        new_redone = min(new_calls, max(0, int(redone - self._state_seen[1])))
        self._state_seen = (calls, redone)
        self._err_seen = max(self._err_seen, float(h[0]))
        self.rescore_stats["eps"] = float(h[2])
        self.rescore_stats["guard_max"] = max(self.rescore_stats.get("guard_max", 0.0), float(h[7]))
        k, kc = self._state_pending
        self._state_pending = None
        for i in range(new_calls):
            self._note_verdict(i >= new_redone, k, kc)
        if self._in_proved_mode():
            self._count_proved(new_calls - new_redone)

    _first_pass_hook = None
    _proved_eps_cache = None      # (engine, eps as the float32 handed to the verdict or None: monitored mode)

    def _in_proved_mode(self) -> bool:
        c = self._proved_eps_cache
        return c is not None and c[0] is self._engine and c[1] is not None and self._engine.dense_precision != "f16x1"

    def _proved_eps(self) -> Optional[float]:
        """The a-priori bound for the bound engine, rounded UP to a float32 (the verdict compares in fp32: gap = fl(e_k - m) > eps, one
        rounding of relative 2^-24 on a gap of at most 2 / tau -- covered by the 2^-16 relative slack added here), or inf when a
        guard fails.  Computed once per engine; also fixes the device-side guard limit GATE_GUARD / max |gi| from the item index."""
        eng = self._engine
        if self._proved_eps_cache is not None and self._proved_eps_cache[0] is eng:
            return self._proved_eps_cache[1]
        from . import f16x3_bound as FB

        eps = float(self._bound_from_weights(eng.spec).get("eps", math.inf))
        if math.isfinite(eps):
            if self._upper_poly() is not None:
                eps = 0.0          # the first pass writes upper bounds of the fp32 logits: the verdict is e_k > m itself (strict: ties with an outsider are redone)
            else:
                eps32 = torch.tensor(eps * (1.0 + 2.0 ** -16), dtype=torch.float32)
                eps = float(torch.nextafter(eps32, torch.tensor(float("inf"))))
            gi_max = self._gi_abs_max()
            self._gate_guard_limit = min(FB.GATE_GUARD / gi_max, 3.0e38) if gi_max > 0.0 else 3.0e38
            if not math.isfinite(gi_max):
                eps = math.inf
        self._proved_eps_cache = (eng, eps)
        self.rescore_stats["proved_calls"] = 0
        self.rescore_stats["bound_violations"] = 0
        return eps

    def _gi_abs_max(self) -> float:
        """max |gi| over the corpus, from the item-gate rows of the tile-packed index (fp32 in both formats; padding rows are zero).
        Once per index: metadata of the bound, not part of the scoring path."""
        idx = self._index
        tiles = (idx.n_items + E.TILE_ITEMS - 1) // E.TILE_ITEMS
        if tiles == 0:
            return 0.0
        tf = idx.buf.numel() // tiles
        gi_f = E.TILE_ITEMS * self._engine.spec.num_logits
        view = idx.buf.view(tiles, tf)[:, tf - gi_f :]
        parts = [torch.stack(torch.aminmax(view[t0 : t0 + (1 << 18)])) for t0 in range(0, tiles, 1 << 18)]    # chunks of 8 M items; no host wait per chunk
        m = torch.stack(parts)                      # (chunks, 2): per-chunk (min, max); a NaN anywhere propagates
        lo, hi = float(m[:, 0].min()), float(m[:, 1].max())
        if lo != lo or hi != hi:
            return math.inf
        return max(-lo, hi, 0.0)

    def _count_proved(self, n_clear: int) -> None:
        """Calls whose verdict cleared count as PROVED while no observed |first pass - fp32| has exceeded the a-priori bound."""
        eps = self._proved_eps_cache[1]
        if self._err_seen > eps:
            self.rescore_stats["bound_violations"] = self.rescore_stats.get("bound_violations", 0) + 1
            return
        self.rescore_stats["proved_calls"] = self.rescore_stats.get("proved_calls", 0) + max(0, n_clear)

    def stats(self) -> Dict[str, float]:
        """rescore_stats brought up to date with the device-side verdicts and the audit counter (synchronises), plus the a-priori
        bound of rigorous_eps() next to the empirical eps the verdicts use."""
        self._absorb_state(wait=True)
        out = self.audit_summary()
        eng = self._bind()
        if eng.exact is not None:
            out.update(self.rigorous_eps())
        return out

    def rigorous_eps(self) -> Dict[str, float]:
        """The A-PRIORI bound on |first pass - fp32 logit| that holds for EVERY (query, item) pair, from the pair-gate weights alone
        (rails_amd/f16x3_bound.py, which states the arithmetic model and the propagation; oracle/f16x3_bound.py restates it and
        tests/test_f16x3_bound_cpu.py checks it against float64 evaluations of both arithmetics).
          eps_rigorous          the bound for this module's first pass: finite for the f16x3 pass of a glu_silu module with a hidden pair-gate
                                layer and l2-normalised components (random-init amzn-books: ~0.7 on logits in [-20, 20], against an observed
                                maximum of 3e-5 -- the bound adds absolute values where the real roundings cancel); infinite otherwise,
                                and for the one-product pass "f16-exact", whose operands carry 11 bits (its bound is the trivial 2 / tau).
          eps_rigorous_usable   True iff the verified top-k of this module RUNS on that bound: the verdicts then compare e_k - m with
                                eps_rigorous itself and a cleared call is proved, not merely monitored.
          eps_default           the calibrated empirical eps of the monitored modes, for comparison."""
        eng = self._bind()
        cached = getattr(self, "_rig_cache", None)
        if cached is not None and cached[0] is eng:
            return cached[1]
        single = eng.dense_precision == "f16x1"
        inv_tau = 1.0 / float(eng.spec.temperature)
        default = (self.RESCORE_EPS_PER_INV_TEMPERATURE_F16X1 if single else self.RESCORE_EPS_PER_INV_TEMPERATURE) * inv_tau
        if single:
            out = {"eps_rigorous": 2.0 * inv_tau, "eps_default": default, "eps_rigorous_usable": False}
        else:
            terms = self._bound_from_weights(eng.spec)
            bound = float(terms.get("eps", math.inf))
            out = {"eps_rigorous": bound, "eps_default": default, "eps_rigorous_usable": bool(math.isfinite(bound) and eng.exact is not None),
                   "eps_rigorous_terms": {k: v for k, v in terms.items() if k != "eps"}}
            if eng.exact is not None and self._item_embeddings.is_cuda:
                from . import arith_check

                out["arithmetic_model_on_device"] = arith_check.report(self._item_embeddings.device)     # H1-H3 re-measured on this device (worst error / bound)
            if eng.exact is not None and self._upper_poly() is not None:
                # one eps for every pair is too coarse for this shape: the first pass adds a per-pair bound (quadratic in the pair's largest
                # |cross logit|) to its logit and the verdict compares upper bounds with exact scores, eps = 0
                info = self._upper_poly_info or {}
                out.update({"bound_kind": "per-pair upper bound", "upper_bound_poly": list(self._upper_poly()), "eps_of_max_abs_cl": info.get("eps_of_c")})
        self._rig_cache = (eng, out)
        return out

    # Shadow audit: every AUDIT_EVERY-th verified call is ALSO run on the dense fp32 path and compared bit for bit; the counts are in
    # rescore_stats["audited" / "mismatches"] (bench.py reports them).  0 = off.  RAILS_AUDIT_EVERY overrides the default.
    AUDIT_EVERY = int(__import__("os").environ.get("RAILS_AUDIT_EVERY", "0"))

    def _audit(self, query_embeddings: torch.Tensor, k: int, scores: torch.Tensor, ids: torch.Tensor, **kwargs) -> None:
        """Dense fp32 top-k of the same batch on a side stream (after the verified result is complete), compared on the device; the
        mismatch count is accumulated in a device counter and read when rescore_stats is next summarised (audit_summary())."""
        cur = torch.cuda.current_stream(query_embeddings.device)
        if self._audit_stream is None:
            self._audit_stream = torch.cuda.Stream(query_embeddings.device)
            self._audit_bad = torch.zeros(1, dtype=torch.int64, device=query_embeddings.device)
        side = self._audit_stream
        side.wait_stream(cur)
        for t in (query_embeddings, scores, ids):
            t.record_stream(side)
        with torch.cuda.stream(side):
            ref_s, ref_i = self._forward_fp32_dense(query_embeddings, k, _private=True, **kwargs)
            bad = (ref_i != ids).any() | (ref_s != scores.to(ref_s.dtype)).any()
            self._audit_bad += bad.to(torch.int64)
        cur.wait_stream(side)     # the recycled scratch buffers of the next call must not race with the audit
        self.rescore_stats["audited"] += 1

    def audit_summary(self) -> Dict[str, int]:
        """rescore_stats with the device-side mismatch counter folded in (one synchronising read)."""
        if self._audit_stream is not None:
            self.rescore_stats["mismatches"] = int(self._audit_bad.item())
        return dict(self.rescore_stats)

    SAFETY_F16X3 = 8.0      # eps >= SAFETY x the running maximum of |s16 - s32| over the re-scored candidates and probes
    SAFETY_F16X1 = 3.0

    # Speculation pays on large corpora only: below SPECULATE_MIN_ITEMS the fixed cost of the verification (~0.1 ms) exceeds what
    # the faster first pass saves (ML-20M, 27 278 items: fp32 step 0.26 ms; amzn-books shape at 16 384 items: 0.19 against 0.21 ms of GPU time,
    # at 20 000: 0.173 with per-pair bounds, at 32 768: 0.219 against 0.343 ms), so the exact modes run the dense fp32 kernels there.
    # It also needs scores that are not crowded around the k-th place: when more than a quarter of the last 16 speculative calls
    # had to be redone, the next 256 calls go straight to the dense fp32 path, then speculation is tried again.
    SPECULATE_MIN_ITEMS = 1 << 14

    def _speculation_paused(self) -> bool:
        if self._pause_left == 0 and len(self._recent) >= 16:
            bad = self._recent.count(False)
            self._recent.clear()
            if bad > 4:
                self._pause_left = 256
        if self._pause_left > 0:
            self._pause_left -= 1
            self.rescore_stats["paused_calls"] = self.rescore_stats.get("paused_calls", 0) + 1
            return True
        return False

    def _read_stats(self, stats: torch.Tensor) -> Tuple[float, float]:
        """(rows, 2) per-row [max |exact - approx|, k-th exact - min candidate approx] -> (largest error, smallest margin) on the
        host: an async copy into pinned memory and a spin on its event.  (A blocking read parks the thread on an interrupt-driven
        wait whose wake-up cost 0.1-0.5 ms per call here; the result is due in microseconds.)"""
        B = stats.shape[0]
        if self._ok_host is None or self._ok_host.shape[0] < B:
            self._ok_host = torch.empty((max(B, 64), 2), dtype=torch.float32).pin_memory()
            self._ok_event = torch.cuda.Event()
        self._ok_host[:B].copy_(stats, non_blocking=True)
        self._ok_event.record()
        while not self._ok_event.query():
            pass
        h = self._ok_host[:B]
        err, gap = float(h[:, 0].max()), float(h[:, 1].min())
        if bool(torch.isnan(h).any()):
            err = float("inf")
        return err, gap

    RISK_POOL = 4096      # highest-norm items of the corpus kept as a probe pool
    RISK_ALWAYS = 16      # ... the top of it is probed on every call

    def _probes(self, B: int, N: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Two (B, 32) blocks of corpus positions re-scored next to the candidates so that |first pass - fp32| is watched OUTSIDE them too:
        32 drawn uniformly from the whole corpus (row block `call mod 64` of a pool drawn once per (B, N)), the RISK_ALWAYS items
        of largest embedding norm on every call, and 16 more rotating through the RISK_POOL highest-norm items -- large inputs make
        large gate pre-activations, which is where a reduced-precision first pass is furthest off."""
        dev = self._item_embeddings.device
        pool = self._probe_pool
        if pool is None or pool.shape[1] != B or self._probe_n != N:
            g = torch.Generator(device=dev).manual_seed(0x5EED)
            pool = self._probe_pool = torch.randint(0, N, (64, B, E.TILE_ITEMS), generator=g, device=dev, dtype=torch.int64)
            self._probe_n = N
            self._risk_pool = None
        if self._risk_pool is None:
            # row norms of the raw item table, in slices (monitoring metadata computed once per corpus; not part of the scoring path)
            X, best_v, best_i = self._item_embeddings[0], None, None
            for lo in range(0, N, 1 << 22):
                nv = torch.linalg.vector_norm(X[lo : lo + (1 << 22)].float(), dim=1)
                kk = min(self.RISK_POOL, nv.numel())
                v, i = torch.topk(nv, kk)
                i = i + lo
                if best_v is not None:
                    v, i = torch.cat([best_v, v]), torch.cat([best_i, i])
                    v, sel = torch.topk(v, min(self.RISK_POOL, v.numel()))
                    i = i[sel]
                best_v, best_i = v, i
            self._risk_pool = best_i.to(torch.int64)
            self._risk_rows = None
        if self._risk_rows is None:
            # every row the rotation can produce, once: (phases, 32) positions = the RISK_ALWAYS items + 16 of the rest, window start
            # (phase * 16) mod len(rest).  Per call the probes are then two views -- no index arithmetic on the device (the arange /
            # add / remainder / index / cat chain of the first version was seven small launches, ~40 us of a 1.7 ms step).
            risk = self._risk_pool
            n_always = min(self.RISK_ALWAYS, risk.numel())
            rest = risk[n_always:] if risk.numel() > n_always else risk
            n_rot = E.TILE_ITEMS - n_always
            m = max(rest.numel(), 1)
            phases = m // math.gcd(m, n_rot)
            idx = (torch.arange(phases, device=dev)[:, None] * n_rot + torch.arange(n_rot, device=dev)[None, :]) % m
            self._risk_rows = torch.cat([risk[:n_always].unsqueeze(0).expand(phases, -1), rest[idx]], dim=1).contiguous()
        call = self.rescore_stats["calls"]
        return pool[call % 64], self._risk_rows[call % self._risk_rows.shape[0]].unsqueeze(0).expand(B, -1)

    def _dense_fp32_index(self) -> E.MolIndex:
        ex = self._engine.exact
        if self._index32 is None or self._index32_engine is not ex:
            self._index32, self._index32_engine = ex.build_index(self._item_embeddings[0]), ex
        return self._index32

    def _forward_fp32_dense(self, query_embeddings: torch.Tensor, k: int, _private: bool = False, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """The exact-fp32 brute force of this module's corpus (fallback of a failed verification, small corpora, the audit).
        Same policies as the plain fp32 module: never more than MAX_LOGIT_BYTES of logits (corpus chunks beyond that), and with
        keep_dense_fp32_index False -- or when _bind declined the second index for lack of memory -- no resident fp32 index either:
        each chunk's index is rebuilt from the raw rows and dropped."""
        ex = self._engine.exact
        B, N = query_embeddings.size(0), self._index.n_items
        have32 = self._index32 is not None and self._index32_engine is ex
        if have32 and B * N * 4 <= self.MAX_LOGIT_BYTES:
            n_q = ex.lib.rails_mol_query_pack_floats(E.C.byref(ex.shape), B)
            qpack32, _, _ = ex.query_pack(query_embeddings, kwargs.get("user_ids"), out=None if _private else self._buf("qpack32", n_q, torch.float32))
            out = None if _private else self._buf("logits", B * N, torch.float32).view(B, N)
            ws = None if _private else self._buf("topk_ws", E._lib.load().rails_topk_workspace_bytes(B, N, k), torch.uint8)
            scores, ids = E.topk(ex.score_dense(qpack32, B, self._index32, out=out), k, ids=self._ids_flat, workspace=ws)
            return scores.to(query_embeddings.dtype), ids
        if have32:
            return self._forward_chunked(query_embeddings, k, _engine=ex, _index=self._index32, **kwargs)
        # no resident fp32 index: temporary per-chunk indexes (CHUNK_ITEMS rows at a time), merged like _forward_chunked
        C = min(self.CHUNK_ITEMS, max(1, self.MAX_LOGIT_BYTES // (4 * max(B, 1))))
        C = max(E.TILE_ITEMS, C // E.TILE_ITEMS * E.TILE_ITEMS)
        qpack32, _, _ = ex.query_pack(query_embeddings, kwargs.get("user_ids"))
        part_s, part_p = [], []
        for lo in range(0, N, C):
            n = min(C, N - lo)
            idx = ex.build_index(self._item_embeddings[0, lo : lo + n])
            s_, p_ = E.topk(ex.score_dense(qpack32, B, idx), min(k, n))
            part_s.append(s_)
            part_p.append(p_ + lo)
            del idx
        scores, pos = E.topk(torch.cat(part_s, 1), k, ids=torch.cat(part_p, 1))
        return scores.to(query_embeddings.dtype), self._ids_flat[pos]

    def _bind(self) -> E.MolEngine:
        eng = super()._bind()
        if eng.exact is not None and self._calib_engine is not eng:   # a new engine (other weights or precision): calibrate afresh
            self._calib_engine = eng
            self._err_seen, self._pad_scale, self._pause_left = 0.0, 1, 0
            self._recent.clear()
            self._verdict_state = None
            self._state_pending = None
        if eng.exact is not None and self._index32_engine is not eng.exact and self.keep_dense_fp32_index is not False:
            # precision "f16x3-exact": a dense fp32 index next to the f16x3 one makes the candidates a gather (10 us) instead of
            # an index build of their raw rows (160 us), and is the fallback's index.  Same bytes again; skipped (None) when
            # less than twice that is free, or when keep_dense_fp32_index is set to False.
            need = self._index.buf.numel() * 4
            free, _ = torch.cuda.mem_get_info(self._item_embeddings.device)
            self._index32 = None
            self._index32_engine = eng.exact
            if self.keep_dense_fp32_index or free > 2 * need:
                self._index32 = eng.exact.build_index(self._item_embeddings[0])
            # ... and, where the candidates are re-scored in place and a third copy is small change (ROWS_COPY_MAX_BYTES), the same index
            # row-major: a candidate's bytes then come in whole cache lines (score_indexed_rows: 8 x fewer bytes than the tile-packed reads)
            self._rows32 = None
            if (self._index32 is not None and self.ROWS_COPY_MAX_BYTES > 0 and need <= self.ROWS_COPY_MAX_BYTES and eng.exact.score_indexed_supported(32, 1024)):
                free, _ = torch.cuda.mem_get_info(self._item_embeddings.device)
                if free > 2 * need:
                    self._rows32 = eng.exact.build_index_rows(self._index32)
        return eng

def _pinned_word(module) -> torch.Tensor:
    """One int32 in pinned host memory, zeroed on the device by the first launch of the call that takes it.  A word goes back to the module's
    free list only when its verdict has been read (_release_words): a call of many slices, or several batches in flight, never share one."""
    free = module.__dict__.setdefault("_flag_free", [])
    return free.pop() if free else torch.zeros(1, dtype=torch.int32).pin_memory()


def _release_words(module, words) -> None:
    free = module.__dict__.setdefault("_flag_free", [])
    for w in words:
        if not w.is_cuda and len(free) < 64:
            free.append(w)


def _verdicts_clear(pending: list, module=None) -> bool:
    """pending: int32 verdict words of fused scans (1 = a candidate count left its range), read after everything that depends on them is
    enqueued.  Words in PINNED HOST memory (the component scans write theirs there: the kernels store through the device-visible address)
    are read after a spin on an event -- no copy launch, no blocking sync (a blocking read parks the thread on an interrupt whose wake-up
    costs 20-100 us of idle GPU per batch); device words cost one synchronising read each."""
    if not pending:
        return True
    if any(not b.is_cuda for b in pending) and torch.cuda.is_available():
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
    clear = all(int(b.item() if b.is_cuda else b.numpy()[0]) == 0 for b in pending)     # (pinned words: a plain memory read through the numpy view)
    if module is not None:
        _release_words(module, pending)
    return clear


class MoLAvgTopK(MoLTopKModule):
    DEVICE_REDO_BYTES = 1 << 30   # materialised score matrices up to this size are kept as the device-side redo buffer of a fused scan;
                                  # beyond it (a 125 M-item shard: 16 GB) the counts are read on the host after the call is enqueued --
    DEVICE_REDO_FREE_FRACTION = 0.0    # > 0: larger buffers too, when they fit this fraction of the free memory.  Measured on a full config-5 shard (16 GB
                                       # buffer, round 5): the dozen predicated no-op launches of the redo (grids sized for 125 M items) cost as much as the
                                       # host's look at the verdict word (0.884 ms per batch either way) and the stream overlap of submit / result is lost
                                       # (0.79 -> 0.85 ms pipelined): off

    REDO_TOPK_TWO_LAUNCHES = 49152 * 24576      # n * K' up to which the predicated redo's top-K' is two launches (rails_topk's two-level plan: chunks of
                                                # <= 49 152 scores, <= 24 576 winners' keys); beyond it the radix route's nine

    def _device_redo_fits(self, nbytes: int, n_items: Optional[int] = None) -> bool:
        """May a (B, N) redo buffer of `nbytes` live on the device?  Decided once per size (the buffer is recycled across calls).
        n_items: the redo also has to be SHORT when it does not run -- every predicated launch of it is a 4.7 us no-op behind each call, and a
        top-K' beyond the two-level plan is nine of them (amzn-books, K' = 4 000: 47 us of a 0.28 ms call); such calls leave the verdict to
        the host (its look costs less than that)."""
        if n_items is not None and n_items * self._avg_top_k > self.REDO_TOPK_TWO_LAUNCHES:
            return False
        if nbytes <= self.DEVICE_REDO_BYTES:
            return True
        if self.DEVICE_REDO_BYTES <= 0 or self.DEVICE_REDO_FREE_FRACTION <= 0.0:     # tests: "as if it did not fit"
            return False
        memo = self.__dict__.setdefault("_redo_fit_memo", {})
        if nbytes not in memo:
            free, _ = torch.cuda.mem_get_info(self._item_embeddings.device)
            memo[nbytes] = nbytes <= self.DEVICE_REDO_FREE_FRACTION * free
        return memo[nbytes]

    """Two-pass approximate top-k (reference rails/indexing/mol_top_k.py:296-429): a bf16 dot product of the
    P_Q-summed query components against the P_X-averaged item components picks `avg_top_k` candidates per query,
    which are then scored with the full MoL and cut to k.  Spans keep the reference's profiler names."""

    def __init__(self, mol_module: MoLSimilarity, item_embeddings: torch.Tensor, item_ids: torch.Tensor, avg_top_k: int) -> None:
        super().__init__(mol_module=mol_module, item_embeddings=item_embeddings, item_ids=item_ids)
        self._avg_top_k: int = avg_top_k
        self.fused_coarse_min_items: int = 262144    # below this the (B, N) scores are small and one launch chain shorter
        self._coarse_engine = None
        self._coarse_table = None
        self._coarse_prefilter = None
        self._verdict_pool: list = []
        self._side_streams = None
        self._side_turn = 0

    OVERLAP_BATCHES = True            # submit(): speculative calls alternate between two streams of the module (see submit)
    PREFILTER_MIN_ITEMS = 4_000_000   # the int8 copy of the coarse table pays where the streaming pass is bound by HBM reads

    def _table(self) -> torch.Tensor:
        eng = self._bind()
        if self._coarse_engine is not eng:
            self._coarse_engine = eng
            self._coarse_table = eng.build_coarse_table(self._index, self._item_embeddings[0])
            self._coarse_prefilter = eng.build_coarse_prefilter(self._coarse_table) if self._coarse_table.shape[0] >= self.PREFILTER_MIN_ITEMS else None
        return self._coarse_table

    PREFILTER_MAX_FIRED = 0.35        # fraction of (tile, query tile) blocks passing the integer bound beyond which the copy is dropped
    PREFILTER_CHECK_CALLS = (2, 64)   # the header's statistics are read (16 bytes, one sync) after this many calls, then every so many

    def _prefilter(self, count: bool = True) -> Optional[torch.Tensor]:
        """The int8 copy of the coarse table (None: not built, or dropped).  count = False: a look that is not a scan (statistics, bench
        probes) and must not advance the check schedule."""
        self._table()
        pre = self._coarse_prefilter
        if pre is not None and count:
            # a table whose one scale is set by a few outliers makes most tiles pass the bound: still exact, but the pass then reads
            # both copies.  The select scans keep (fired, tested) counts in the header; looked at now and then -- WITHOUT a host wait:
            # the 16 bytes are copied to pinned memory behind the launches already enqueued and read by a later call, once the copy
            # has landed (a blocking read here stalled the submit / result pipeline on call 3 and on every 64th call)
            self._prefilter_calls = getattr(self, "_prefilter_calls", 0) + 1
            pend = getattr(self, "_prefilter_pending", None)
            first, every = self.PREFILTER_CHECK_CALLS
            if pend is not None:
                if pend[1].query():
                    fired, tested = (int(v) for v in pend[0])
                    self._prefilter_pending = None
                    if tested > 0 and fired > self.PREFILTER_MAX_FIRED * tested:
                        self._coarse_prefilter = pre = None
            elif self._prefilter_calls == first + 1 or self._prefilter_calls % every == 0:
                host = torch.empty(2, dtype=torch.int64).pin_memory()
                host.copy_(pre[32:48].view(torch.int64), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._prefilter_pending = (host, ev)
        return pre

    def prefilter_stats(self) -> Optional[dict]:
        """(tile, query tile) blocks of the int8 select scans so far: how many passed the integer bound, how many were tested."""
        if self._coarse_prefilter is None:
            return None
        fired, tested = (int(v) for v in self._coarse_prefilter[32:48].view(torch.int64).cpu())
        return {"fired": fired, "tested": tested, "fraction": fired / tested if tested else 0.0}

    def _coarse_topk(self, query_embeddings: torch.Tensor, average_queries: bool, pending: Optional[list] = None, **kwargs):
        eng = self._bind()
        table = self._table()
        qpack, eq, _ = eng.query_pack(query_embeddings, kwargs.get("user_ids"), want_plain=True)
        return qpack, self._coarse_topk_from_eq(eq, average_queries, pending)

    def _coarse_topk_from_eq(self, eq: torch.Tensor, average_queries: bool, pending: Optional[list] = None, with_scores: bool = False):
        """(B, P_Q, d) query components -> (B, avg_top_k) positions of the coarse top-K', best first
        (with_scores: -> (scores, positions), the bf16 coarse scores as fp32).
        The fused scan's result is exact iff every row collected between K' and `capacity` candidates.  With `pending` (a
        list) the check is DEFERRED: the positions are returned at once, the scan's device verdict word (1 = out of range) is
        appended, and the caller reads it after it has enqueued everything that depends on the positions (speculate, then
        verify: the GPU never waits for the host in mid-pipeline).  Without it the check is made here (one 4-byte D2H copy)."""
        eng = self._bind()
        table = self._table()
        n = table.shape[0]
        if self._avg_top_k > n:
            raise RuntimeError(f"selected index k out of range (k={self._avg_top_k}, n={n})")
        if eq.shape[0] > 128:   # the scan keeps ceil(B / 32) query tiles in LDS: larger batches go in slices
            parts = [self._coarse_topk_from_eq(eq[b0 : b0 + 128], average_queries, pending, with_scores) for b0 in range(0, eq.shape[0], 128)]
            return (torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)) if with_scores else torch.cat(parts, dim=0)
        # large corpora: fused scan + threshold select, no (B, N) score matrix (16 GB per 125 M-item shard at B = 32).
        # Same scores and the same exact top-K' as the materialising path below -- when every query's candidate count
        # landed inside [K', capacity]; the check costs one 128-byte device-to-host copy.
        if n >= self.fused_coarse_min_items and self._avg_top_k <= 4096 and not getattr(self, "_no_fused", False):
            on_device = self._device_redo_fits(eq.shape[0] * n * 4, n)
            # a verdict the HOST reads (no device redo, the caller defers the look): the word lives in pinned host memory and the kernels write it
            # there themselves -- no 4-byte copy behind the call's last launch (round 6, as the component scans)
            word = _pinned_word(self) if (not on_device and pending is not None) else None
            fused = eng.coarse_topk(eq, table, average_queries, self._avg_top_k, with_flag=True, prefilter=self._prefilter(), flag=word)
            if fused is not None:
                # bad: 1 iff some row's candidate count is outside [K', capacity] -- raised by the call's key-selection launch
                sc, idx, counts, bad = fused
                if on_device:
                    # the redo ON THE DEVICE: the materialising scan and its top-K' are enqueued under that flag as their launch
                    # predicate and overwrite (sc, idx) -- no-ops unless a count was out of range; nothing for the host to wait
                    # for (the (B, N) score buffer is recycled across calls)
                    coarse = eng.coarse_scores(eq, table, average_queries, out=self._buf("coarse_all", eq.shape[0] * n, torch.float32).view(eq.shape[0], n), run_if=bad)
                    E.topk(coarse, self._avg_top_k, out=(sc, idx), run_if=bad)
                    return (sc, idx) if with_scores else idx
                if pending is not None:      # the caller reads the verdict word after it has enqueued everything that follows
                    pending.append(bad)
                    return (sc, idx) if with_scores else idx
                if int(bad.item()) == 0:
                    return (sc, idx) if with_scores else idx
        coarse = eng.coarse_scores(eq, table, average_queries)
        sc, idx = E.topk(coarse, self._avg_top_k)
        return (sc, idx) if with_scores else idx

    def rerank(self, qpack: torch.Tensor, batch: int, cand_idx: torch.Tensor, k: int):
        """Full MoL on per-row candidates (positions, (B, K')) -> exact top-min(k, K') among them."""
        eng = self._bind()
        scores = self._score_at(eng, qpack, batch, cand_idx)
        return E.topk_candidates(scores, min(k, cand_idx.shape[1]), cand_idx, self._ids_flat)

    def _enqueue(self, query_embeddings: torch.Tensor, k: int, pending: list, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """One pass of forward's launches; the device verdicts of the fused scan (int32 words, 1 = redo) are appended to `pending`."""
        # the reference's four profiler spans (mol_top_k.py:350-382), around the launches that do the same work here (entered only
        # under a profiler: a span costs ~9 us of host time, four of them more than the GPU needs for the launches they bracket)
        span = torch.profiler.record_function if torch.autograd._profiler_enabled() else (lambda name: contextlib.nullcontext())
        with span("avg_top_k_scoring"):
            qpack, idx = self._coarse_topk(query_embeddings, average_queries=False, pending=pending, **kwargs)
        eng = self._bind()
        with span("avg_topk_selection"):
            pass    # the reference gathers the candidates' embeddings here; they are read in place by the scoring launch below
        with span("filtered_scoring"):
            cand_scores = self._score_at(eng, qpack, query_embeddings.size(0), idx)
        with span("final_topk"):
            scores, ids = E.topk_candidates(cand_scores, min(k, idx.shape[1]), idx, self._ids_flat)   # top-k + gather + id lookup, one launch
        return scores.to(query_embeddings.dtype), ids

    # ---- two-stage form of forward ---------------------------------------------------------------------------------------------
    # The fused coarse scan is exact unless a candidate count left its range (heavy ties at the threshold), which only the device
    # knows when the launches are enqueued.  forward() = result(submit()): submit enqueues the whole call on that assumption and
    # copies the scan's verdict word to pinned host memory behind it; result waits for THAT copy (not for the stream), and redoes
    # the call on the materialising path in the rare other case.  A caller with the next batch at hand calls submit(batch i + 1)
    # before result(batch i): the host's look at the verdict then costs the GPU nothing (bench.py --two-pass reports both rates).
    # The handle submit() returns is OPAQUE: the tensors inside it may still be in flight on a stream of the module's own and are
    # joined to the caller's stream only by result() -- read them through result(), never through the handle, and do not drop a handle
    # without calling result() (ShardedTopK.submit, which needs the scores earlier, waits on the handle's event explicitly).
    def submit(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs):
        if k > self._avg_top_k:  # the reference raises after doing the work (mol_top_k.py:383-386)
            raise ValueError(f"avg_top_k ({self._avg_top_k}) must be larger than k ({k})")
        # Calls that will be speculative (the (B, N) redo buffer does not fit: large shards) alternate between two streams of the
        # module's own: the latency-bound launches of one batch (prologue, sample, threshold, key selection, rerank: ~0.15 ms of
        # a 0.85 ms call on a 125 M-item shard) then run under the table scan of the neighbouring batch.  The caller's stream joins
        # a call's stream in result().  Calls that share the module's redo buffers (small corpora) stay on the caller's stream.
        side = None
        if self.OVERLAP_BATCHES and query_embeddings.is_cuda and not self._device_redo_fits(query_embeddings.size(0) * self.num_items * 4, self.num_items):
            if self._side_streams is None:
                self._side_streams = [torch.cuda.Stream(query_embeddings.device), torch.cuda.Stream(query_embeddings.device)]
            side = self._side_streams[self._side_turn]
            self._side_turn ^= 1
            side.wait_stream(torch.cuda.current_stream(query_embeddings.device))     # the inputs are ready where the caller stands
            query_embeddings.record_stream(side)
            for v in kwargs.values():          # user_ids and the like: read on the call's stream after the caller may have let go of them
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(side)
        pending: list = []
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()), self.one_bind():
            scores, ids = self._enqueue(query_embeddings, k, pending, **kwargs)
            if not pending:
                host = None
            elif all(not bad.is_cuda for bad in pending):
                host = list(pending)           # already in pinned host memory (written by the scans themselves): nothing to copy
            else:
                pool = self._verdict_pool      # pinned words go back to the pool in result(): no host allocation per call
                host = pool.pop() if pool and pool[-1].numel() == len(pending) else torch.empty(len(pending), dtype=torch.int32, pin_memory=True)
                for j, bad in enumerate(pending):
                    host[j : j + 1].copy_(bad, non_blocking=True)
            done = torch.cuda.Event() if (pending or side is not None) else None
            if done is not None:
                done.record()
        if not pending and side is None:
            return ("final", scores, ids)
        return ("speculative", scores, ids, host, done, query_embeddings, k, kwargs, side)

    def result(self, handle) -> Tuple[torch.Tensor, torch.Tensor]:
        if handle[0] == "final":
            return handle[1], handle[2]
        _, scores, ids, host, done, query_embeddings, k, kwargs, side = handle
        if side is not None:           # the caller's stream takes over the outputs
            cur = torch.cuda.current_stream(scores.device)
            cur.wait_event(done)
            scores.record_stream(cur)
            ids.record_stream(cur)
        redo = False
        if host is not None:
            while not done.query():      # spin: the word is microseconds away, and a blocking wait parks the thread on an interrupt whose
                pass                     # wake-up costs 50-100 us of GPU idle per batch (see MoLBruteForceTopK._read_stats)
            if isinstance(host, list):
                redo = any(int(w.numpy()[0]) != 0 for w in host)
                _release_words(self, host)
            else:
                redo = int(host.max()) != 0
                if len(self._verdict_pool) < 8:
                    self._verdict_pool.append(host)
        if not redo:
            return scores, ids
        self._no_fused = True      # redo this call on the materialising path
        try:
            return self._enqueue(query_embeddings, k, [], **kwargs)
        finally:
            self._no_fused = False

    def forward(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        with self.inline_calls():      # nothing to overlap with: stay on the caller's stream (the hand-over between streams costs ~50 us)
            return self.result(self.submit(query_embeddings, k, sorted, **kwargs))

    def forward_filtered(self, query_embeddings: torch.Tensor, k_prime: int, invalid_ids: torch.Tensor, k: int, **kwargs):
        """CandidateIndex.get_top_k_outputs' body: forward(k_prime) + the seen-id filter, with the filter enqueued BEFORE the host looks
        at the scan's verdict word (it then runs while the host is on its way back) -> (top_k_ids, top_k_scores); None where this
        does not apply (subclasses with their own forward, k_prime beyond K': the caller composes the two calls)."""
        if type(self).forward is not MoLAvgTopK.forward or k_prime > self._avg_top_k or k_prime < k:
            return None
        with self.inline_calls():
            h = self.submit(query_embeddings, k_prime, **kwargs)
        out = E.filter_seen_ids(h[2], h[1], invalid_ids, k)
        scores, ids = self.result(h)
        return out if scores is h[1] else E.filter_seen_ids(ids, scores, invalid_ids, k)

    @contextlib.contextmanager
    def inline_calls(self):
        """submit() inside this block stays on the caller's stream (what a plain forward wants)."""
        saved = self.OVERLAP_BATCHES
        self.OVERLAP_BATCHES = False
        try:
            yield
        finally:
            self.OVERLAP_BATCHES = saved

    def coarse_candidates(self, query_embeddings: torch.Tensor, **kwargs):
        """Pass 1 on this module's items: -> (coarse scores (B, K'), positions (B, K')), best first (ties by position)."""
        eng = self._bind()
        _, eq, _ = eng.query_pack(query_embeddings, kwargs.get("user_ids"), want_plain=True)
        return self._coarse_topk_from_eq(eq, False, None, with_scores=True)

    def rerank_masked(self, query_embeddings: torch.Tensor, cand_idx: torch.Tensor, k: int, **kwargs):
        """Pass 2 on a candidate list with holes: positions < 0 are not this module's items; they score -inf and come back
        with id -1.  -> exact top-min(k, K') (scores, ids)."""
        eng = self._bind()
        qpack, _, _ = eng.query_pack(query_embeddings, kwargs.get("user_ids"))
        hole = cand_idx < 0
        scores = self._score_at(eng, qpack, query_embeddings.size(0), cand_idx.clamp_min(0))   # holes read item 0 and are overwritten below
        scores = torch.where(hole, scores.new_full((), float("-inf")), scores)
        ids = torch.where(hole, cand_idx.new_full((), -1), self._ids_flat[cand_idx.clamp_min(0)])
        return E.topk(scores, min(k, cand_idx.shape[1]), ids=ids)

    def topk_ids(self, query_embeddings: torch.Tensor, sorted: bool = True, **kwargs) -> torch.Tensor:
        """Coarse candidates only, with the P_Q-averaged query (reference mol_top_k.py:398-429) -> positions."""
        return self._coarse_topk(query_embeddings, average_queries=True, **kwargs)[1]


class _ComponentCandidates:
    """Per-component candidate generation shared by MoLNaiveTopK and MoLCombTopK."""

    UNION_CAP = 16384          # candidates per query the rerank sorts and ranks in one workgroup's LDS (rails_sort_rows_i64 / rails_topk)
    UNION_HARD_CAP = 1 << 20   # beyond UNION_CAP (16x16x64 with k_per_group >= 75: 256 * 75 = 19 200) the two integer / float sorts of the
                               # rerank go through torch.sort on the device -- the reference "just runs" there (mol_top_k.py:260, :518);
                               # scoring and duplicate masking stay on the HIP kernels

    def _check_union_size(self, n_candidates: int) -> None:
        if n_candidates > self.UNION_HARD_CAP:
            raise NotImplementedError(
                f"{type(self).__name__}: {n_candidates} candidates per query exceed the rerank capacity of {self.UNION_HARD_CAP} "
                "(P_Q * P_X * k_per_group [+ avg_top_k]); use a smaller k_per_group for this shape")

    def _component_table(self) -> torch.Tensor:
        eng = self._bind()
        if getattr(self, "_comp_engine", None) is not eng:
            self._comp_engine = eng
            self._comp_table = eng.build_component_table(self._index, self._item_embeddings[0])
        return self._comp_table

    def _component_topk(self, eq: torch.Tensor, k_per_group: int, pending: Optional[list] = None) -> torch.Tensor:
        """-> (B, P_Q * P_X * k_per_group) positions: top k_per_group items of every (query group, item group) pair.
        `pending`: deferred validity check of the fused scan, as in MoLAvgTopK._coarse_topk_from_eq."""
        eng = self._bind()
        table = self._component_table()      # (P_X, N, d), item-group-major
        n = table.shape[1]
        if k_per_group > n:
            raise RuntimeError(f"selected index k out of range (k={k_per_group}, n={n})")
        # the component scans keep the fragments of all B * P_Q query rows in LDS and (the fused form) eight row tiles of running maxima in
        # registers (four at d = 128): batches beyond 256 (128) query rows go in slices
        max_b = max(1, (128 if eng.spec.dot_product_dimension >= 128 else 256) // eng.spec.query_dot_product_groups)
        if eq.shape[0] > max_b:
            return torch.cat([self._component_topk(eq[b0 : b0 + max_b], k_per_group, pending) for b0 in range(0, eq.shape[0], max_b)], dim=0)
        # large corpora: fused scan + threshold select, no (B*P_Q*P_X, N) score matrix (5.7 GB at amzn-books, B = 32);
        # identical to the materialising path below whenever every row's candidate count is inside [k, capacity]
        if n >= getattr(self, "fused_component_min_items", 262144) and not getattr(self, "_no_fused", False):
            rows = eq.shape[0] * eng.spec.query_dot_product_groups * eng.spec.item_dot_product_groups
            on_device = rows * n * 4 <= MoLAvgTopK.DEVICE_REDO_BYTES
            if on_device:
                flag = self._buf("redo_flag_c", 1, torch.int32)      # (zeroed by the call's first launch)
            else:
                # the verdict word in pinned host memory, written by the kernels themselves: the caller spins on an event and reads it (no 4-byte
                # copy launch, no blocking .item(): ~30 us of every Naive / Comb call at amzn-books)
                flag = _pinned_word(self)
            fused = eng.component_topk(eq, table, k_per_group, flag)
            if fused is not None:
                sc_c, pos, counts = fused
                if on_device:     # redo on the device under the flag, as in MoLAvgTopK._coarse_topk_from_eq
                    scores = eng.component_scores(eq, table, out=self._buf("component_all", rows * n, torch.float32).view(rows, n), run_if=flag)
                    E.topk(scores, k_per_group, out=(sc_c, pos), run_if=flag)
                    return pos.view(eq.shape[0], -1)
                if pending is not None:      # a device verdict word (1 = redo), read by the caller once everything is enqueued
                    pending.append(flag)
                    return pos.view(eq.shape[0], -1)
                if int(flag.item()) == 0:
                    return pos.view(eq.shape[0], -1)
        scores = eng.component_scores(eq, table)
        _, pos = E.topk(scores, k_per_group)
        return pos.view(eq.shape[0], -1)

    def _filter_inside(self, n_candidates: int, invalid_ids: torch.Tensor, k: int):
        """get_top_k_outputs over a module that returns ALL its candidates ranked (masked duplicates last): the first k unseen entries of that
        list are the first k unseen of its top k + width -- at most `width` scored candidates are seen, and a masked duplicate (-32767.0)
        outranks a scored one only where fewer than k + width scored ones exist, in both forms alike.  -> (invalid_ids, k) where
        rails_topk_candidates_filtered takes the sizes, else None (the caller ranks everything and filters after)."""
        if invalid_ids is None or invalid_ids.dim() != 2 or getattr(self, "NO_FILTER_FUSION", False):
            return None
        width = invalid_ids.shape[1]
        if not E.topk_candidates_filterable(n_candidates, min(k + width, n_candidates), width, k):
            return None
        return (invalid_ids, k)

    UNSORTED_RERANK = True     # get_top_k_outputs: rank the candidates where the scans left them (rails_rerank_topk_filtered) instead of sorting their positions first

    def _rerank_union(self, qpack: torch.Tensor, batch: int, all_indices: torch.Tensor, sorted: bool, seen=None, pending: Optional[list] = None):
        """sort -> gather -> full MoL -> mask duplicates with -32767.0 -> top-k over ALL candidates
        (the reference overwrites k with the candidate count, mol_top_k.py:260 / :518).
        seen = (invalid_ids, k) (get_top_k_outputs, sizes checked by the caller): instead of the full ranking, the first k unseen entries of it
        -- the top k + width of the candidates with the seen-id filter inside the selection launch -> (ids, scores).  Where the fused scans'
        verdict word is pending anyway (pinned host memory, read by the caller once everything is enqueued), the integer sort goes too: the
        candidates are scored in the order the scans left them and ranked by (score, position) keys, first copy of every position; a row with
        fewer than k + width distinct positions raises the same word and the call is redone on the sorted form."""
        eng = self._bind()
        big = all_indices.shape[1] > self.UNION_CAP
        word = next((b for b in (pending or []) if not b.is_cuda), None)
        if seen is not None and word is not None and self.UNSORTED_RERANK and not getattr(self, "_no_fused", False):
            invalid_ids, k_out = seen
            pos = all_indices.to(torch.int64).contiguous()
            scores = self._score_at(eng, qpack, batch, pos)
            ws = self._buf("rerank_keys", pos.numel() * 8, torch.uint8)
            return E.rerank_topk_filtered(scores, min(k_out + invalid_ids.shape[1], pos.shape[1]), pos, self._ids_flat, invalid_ids, k_out, word, ws)
        sorted_idx = torch.sort(all_indices.to(torch.int64), dim=1).values if big else E.sort_rows(all_indices)
        k = sorted_idx.shape[1]
        scores = self._score_at(eng, qpack, batch, sorted_idx)
        E.mask_sorted_duplicates(sorted_idx, scores, -32767.0)
        if seen is not None:
            invalid_ids, k_out = seen
            return E.topk_candidates_filtered(scores, min(k_out + invalid_ids.shape[1], k), sorted_idx, self._ids_flat, invalid_ids, k_out)
        if big:   # full ranking of more than 16 384 candidates: stable descending sort = (score desc, column asc), rails_topk's tie rule
            vals, order = torch.sort(scores, dim=1, descending=True, stable=True)
            return vals, torch.gather(self._ids_flat[sorted_idx], 1, order)
        return E.topk_candidates(scores, k, sorted_idx, self._ids_flat)


class MoLNaiveTopK(MoLTopKModule, _ComponentCandidates):
    """Reference rails/indexing/mol_top_k.py:133-293 (the FAISS branch is out of scope).  Returns
    (B, P_Q * P_X * k_per_group) columns whatever `k` is, as the reference does."""

    def __init__(self, mol_module: MoLSimilarity, item_embeddings: torch.Tensor, item_ids: torch.Tensor, k_per_group: int, use_faiss: bool = False) -> None:
        if use_faiss:
            raise NotImplementedError("use_faiss=True (FAISS-GPU IVF index) is out of scope")
        super().__init__(mol_module=mol_module, item_embeddings=item_embeddings, item_ids=item_ids)
        self._k_per_group: int = k_per_group
        self._use_faiss: bool = False
        self._check_union_size(mol_module._query_dot_product_groups * mol_module._item_dot_product_groups * k_per_group)

    def forward(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        scores, ids = self._ranked(query_embeddings, sorted, None, kwargs)
        return scores.to(query_embeddings.dtype), ids

    def forward_filtered(self, query_embeddings: torch.Tensor, k_prime: int, invalid_ids: torch.Tensor, k: int, **kwargs):
        """CandidateIndex.get_top_k_outputs' body (the module returns all its candidates whatever k_prime is, the filter keeps the first k unseen):
        -> (top_k_ids, top_k_scores), or None where the filter does not fit the selection launch (_filter_inside)."""
        mol = self._mol_module
        seen = self._filter_inside(mol._query_dot_product_groups * mol._item_dot_product_groups * self._k_per_group, invalid_ids, k)
        if seen is None:
            return None
        ids, scores = self._ranked(query_embeddings, True, seen, kwargs)
        return ids, scores.to(query_embeddings.dtype)

    def _ranked(self, query_embeddings: torch.Tensor, sorted: bool, seen, kwargs):
        eng = self._bind()
        qpack, eq, _ = eng.query_pack(query_embeddings, kwargs.get("user_ids"), want_plain=True)
        for attempt in range(2):     # speculate on the fused scans, verify after everything is enqueued
            pending: list = []
            all_indices = self._component_topk(eq, self._k_per_group, pending)
            out = self._rerank_union(qpack, query_embeddings.size(0), all_indices, sorted, seen, pending)
            if _verdicts_clear(pending, self):
                break
            self._no_fused = True
        self._no_fused = False
        return out


class MoLCombTopK(MoLAvgTopK, _ComponentCandidates):
    """Reference rails/indexing/mol_top_k.py:432-551: per-component candidates + the averaged-query coarse
    candidates, reranked together.  Returns (B, P_Q * P_X * k_per_group + avg_top_k) columns."""

    def __init__(self, mol_module: MoLSimilarity, item_embeddings: torch.Tensor, item_ids: torch.Tensor, avg_top_k: int, k_per_group: int) -> None:
        super().__init__(mol_module=mol_module, item_embeddings=item_embeddings, item_ids=item_ids, avg_top_k=avg_top_k)
        self._k_per_group: int = k_per_group
        self._check_union_size(mol_module._query_dot_product_groups * mol_module._item_dot_product_groups * k_per_group + avg_top_k)

    def forward(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        scores, ids = self._ranked(query_embeddings, sorted, None, kwargs)
        return scores.to(query_embeddings.dtype), ids

    def forward_filtered(self, query_embeddings: torch.Tensor, k_prime: int, invalid_ids: torch.Tensor, k: int, **kwargs):
        """As MoLNaiveTopK.forward_filtered, over the component candidates + the averaged-query candidates."""
        mol = self._mol_module
        seen = self._filter_inside(mol._query_dot_product_groups * mol._item_dot_product_groups * self._k_per_group + self._avg_top_k, invalid_ids, k)
        if seen is None:
            return None
        ids, scores = self._ranked(query_embeddings, True, seen, kwargs)
        return ids, scores.to(query_embeddings.dtype)

    def _ranked(self, query_embeddings: torch.Tensor, sorted: bool, seen, kwargs):
        eng = self._bind()
        qpack, eq, _ = eng.query_pack(query_embeddings, kwargs.get("user_ids"), want_plain=True)
        for attempt in range(2):
            pending: list = []
            comp = self._component_topk(eq, self._k_per_group, pending)
            avg_idx = self._coarse_topk_from_eq(eq, average_queries=True, pending=pending)
            out = self._rerank_union(qpack, query_embeddings.size(0), torch.cat([comp, avg_idx], dim=1), sorted, seen, pending)
            if _verdicts_clear(pending, self):
                break
            self._no_fused = True
        self._no_fused = False
        return out


class MIPSTopKModule(TopKModule):
    """Reference rails/indexing/mips_top_k.py:23-38."""

    def __init__(self, item_embeddings: torch.Tensor, item_ids: torch.Tensor) -> None:
        super().__init__()
        self._item_embeddings: torch.Tensor = item_embeddings
        self._item_ids: torch.Tensor = item_ids


class MIPSBruteForceTopK(MIPSTopKModule):
    """Dot-product brute force (reference rails/indexing/mips_top_k.py:41-81): MFMA scan + exact top-k, all HIP."""

    def __init__(self, item_embeddings: torch.Tensor, item_ids: torch.Tensor) -> None:
        super().__init__(item_embeddings=item_embeddings, item_ids=item_ids)
        if item_embeddings.dim() != 3 or item_embeddings.shape[0] != 1:
            raise ValueError(f"item_embeddings must be (1, N, D), got {tuple(item_embeddings.shape)}")
        del self._item_embeddings
        self._index = E.MipsIndex(item_embeddings[0])
        self._ids_flat = item_ids.reshape(-1).to(device=item_embeddings.device, dtype=torch.int64).contiguous()

    def forward(self, query_embeddings: torch.Tensor, k: int, sorted: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        logits = self._index.score(query_embeddings)
        scores, ids = E.topk(logits, k, ids=self._ids_flat, sorted=sorted)
        return scores.to(query_embeddings.dtype), ids


class CandidateIndex(object):
    """Reference indexing/candidate_index.py:30-185 (`filter_invalid_ids` / `apply_object_filter` are never
    called by any entry point of the reference and are not provided)."""

    def __init__(self, ids: torch.Tensor, embeddings: torch.Tensor, invalid_ids: Optional[torch.Tensor] = None, debug_path: Optional[str] = None) -> None:
        super().__init__()
        self._ids: torch.Tensor = ids
        self._embeddings: torch.Tensor = embeddings
        self._invalid_ids: Optional[torch.Tensor] = invalid_ids
        self._debug_path: Optional[str] = debug_path

    @property
    def ids(self) -> torch.Tensor:
        return self._ids

    @property
    def num_objects(self) -> int:
        return self._ids.size(1)

    @property
    def embeddings(self) -> torch.Tensor:
        return self._embeddings

    def get_top_k_outputs(
        self,
        query_embeddings: torch.Tensor,
        k: int,
        aux_payloads: Dict[str, torch.Tensor],
        top_k_module: TopKModule,
        invalid_ids: Optional[torch.Tensor],
        r: int = 1,
        return_embeddings: bool = False,
        truncate_k_prime_to: Optional[int] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        """-> (top_k_ids (B, k), top_k_scores (B, k), None).  Note: ids first, as in the reference."""
        bind = getattr(getattr(top_k_module, "_local_module", top_k_module), "one_bind", None)
        with bind() if bind is not None else contextlib.nullcontext():     # one look at the model's parameters per call (MoLTopKModule.one_bind)
            return self._get_top_k_outputs(query_embeddings, k, aux_payloads, top_k_module, invalid_ids, r, return_embeddings, truncate_k_prime_to)

    def _get_top_k_outputs(self, query_embeddings, k, aux_payloads, top_k_module, invalid_ids, r, return_embeddings, truncate_k_prime_to):
        if return_embeddings:
            # the reference's own branch is broken (undefined `top_k_indices`, candidate_index.py:182)
            raise NotImplementedError("return_embeddings=True is not supported")
        max_num_invalid_ids = invalid_ids.size(1) if invalid_ids is not None else 0
        k_prime = min(k + max_num_invalid_ids, self.num_objects)
        if truncate_k_prime_to is not None:
            k_prime = min(k_prime, truncate_k_prime_to)
        if invalid_ids is not None and k <= k_prime and hasattr(top_k_module, "forward_filtered"):
            fused = top_k_module.forward_filtered(query_embeddings, k_prime, invalid_ids, k, **aux_payloads)
            if fused is not None:
                return fused[0], fused[1], None
        top_k_prime_scores, top_k_prime_ids = top_k_module(query_embeddings=query_embeddings, k=k_prime, **aux_payloads)
        if invalid_ids is not None:
            if top_k_prime_ids.shape[1] < k:
                # the reference fails in .view(-1, k) here
                raise RuntimeError(f"shape '[-1, {k}]' is invalid: only {top_k_prime_ids.shape[1]} candidates per row")
            top_k_ids, top_k_scores = E.filter_seen_ids(top_k_prime_ids, top_k_prime_scores, invalid_ids, k)
        else:
            top_k_scores, top_k_ids = top_k_prime_scores, top_k_prime_ids
        return top_k_ids, top_k_scores, None


_BUILT = {"MoLBruteForceTopK": lambda mol, x, ids: MoLBruteForceTopK(mol_module=mol, item_embeddings=x, item_ids=ids)}
for _k in (5, 10, 25, 50, 75, 100):
    _BUILT[f"MoLNaiveTopK{_k}"] = (lambda kk: lambda mol, x, ids: MoLNaiveTopK(mol_module=mol, item_embeddings=x, item_ids=ids, k_per_group=kk))(_k)
for _kg, _ka in ((1, 100), (1, 500), (5, 100), (5, 200), (5, 500), (10, 100), (10, 500), (50, 500), (50, 1000), (100, 1000)):
    _BUILT[f"MoLCombTopK{_kg}_{_ka}"] = (lambda g, a: lambda mol, x, ids: MoLCombTopK(mol_module=mol, item_embeddings=x, item_ids=ids, avg_top_k=a, k_per_group=g))(_kg, _ka)
_NO_MOL = {"MIPSBruteForceTopK": lambda x, ids: MIPSBruteForceTopK(item_embeddings=x, item_ids=ids)}
for _k in (100, 200, 500, 1000, 2000, 2500, 3000, 4000):
    _BUILT[f"MoLAvgTopK{_k}"] = (lambda kk: lambda mol, x, ids: MoLAvgTopK(mol_module=mol, item_embeddings=x, item_ids=ids, avg_top_k=kk))(_k)
# accepted by the reference's factory but out of scope here: FAISS-GPU IVF candidate generation
_KNOWN_UNBUILT = ["MoLNaiveFaissTopK5"]


def get_top_k_module(top_k_method: str, model: torch.nn.Module, item_embeddings: torch.Tensor, item_ids: torch.Tensor) -> TopKModule:
    """String -> module factory; `model._ndp_module` is the MoLSimilarity (reference indexing/utils_rails.py:25-233)."""
    if top_k_method in _NO_MOL:
        return _NO_MOL[top_k_method](item_embeddings, item_ids)
    if top_k_method in _BUILT:
        return _BUILT[top_k_method](model._ndp_module, item_embeddings, item_ids)
    if top_k_method in _KNOWN_UNBUILT:
        raise NotImplementedError(f"top_k_method {top_k_method} needs faiss-gpu and is out of scope for rails_amd")
    raise ValueError(f"Invalid top-k method {top_k_method}")
