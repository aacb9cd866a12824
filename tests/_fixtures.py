"""Helpers to load the golden fixtures written by oracle/gen_golden.py."""
import json
import os

import numpy as np
import torch

from oracle.mol_oracle import MoLConfig, hash_item_table

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PER_CONFIG = ["c1_ml1m", "c2_ml20m", "c3_books", "c4_16x16x64"]


class Fixture:
    def __init__(self, name: str):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        d = json.loads(str(self.z["cfg_json"]))
        d["uid_embedding_hash_sizes"] = tuple(d["uid_embedding_hash_sizes"])
        self.cfg = MoLConfig(**d)
        self.weights = self._weights()

    def _weights(self):
        w = {k[2:]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith("w/")}
        # uid tables are stored sliced to the rows used; expand back to (hash_size + 1, d)
        for i, hs in enumerate(self.cfg.uid_embedding_hash_sizes):
            key = f"_query_embeddings_fn._uid_embeddings_{i}.weight"
            rows = w.pop(key + ".rows")
            full = torch.zeros((hs + 1, self.cfg.dot_product_dimension), dtype=torch.float32)
            full[rows] = w[key]
            w[key] = full
        return w

    def t(self, key: str) -> torch.Tensor:
        return torch.from_numpy(np.asarray(self.z[key]))

    def has(self, key: str) -> bool:
        return key in self.z.files

    @property
    def kw(self):
        return {"user_ids": self.t("user_ids")} if self.has("user_ids") else {}

    @property
    def user_ids(self):
        return self.t("user_ids") if self.has("user_ids") else None


def full_size_inputs(fx: Fixture):
    """Rebuild the by-recipe inputs of the F7 fixtures (oracle/gen_golden.py:full_size_fixture)."""
    n = int(fx.z["N"])
    X = torch.from_numpy(hash_item_table(int(fx.z["table_seed"]), 0, n, fx.cfg.item_embedding_dim)).unsqueeze(0)
    g = torch.Generator().manual_seed(int(fx.z["item_ids_seed"]))
    ids = torch.cumsum(torch.randint(1, 4, (n,), generator=g), 0).to(torch.int64).unsqueeze(0)
    return X, ids


def assert_topk_matches(scores, ids, ref_scores, ref_ids, atol=1e-4, tie_tol=1e-5):
    """Tie-aware comparison: scores within `atol`; ids identical except inside groups of
    reference scores closer than `tie_tol` (torch.topk's order among ties is unspecified).  tie_tol = 1e-5 is SURVEY.md section 7's
    definition; profiles/r04_tie_branch_census.json: 31 of 77 244 compared positions differ from the reference's id at all, every one of
    them inside such a run."""
    scores, ids, ref_scores, ref_ids = (torch.as_tensor(x).cpu() for x in (scores, ids, ref_scores, ref_ids))
    assert scores.shape == ref_scores.shape and ids.shape == ref_ids.shape
    assert torch.allclose(scores, ref_scores, atol=atol, rtol=0), float((scores - ref_scores).abs().max())
    B, k = ids.shape
    for b in range(B):
        if torch.equal(ids[b], ref_ids[b]):
            continue
        # split the row into runs of near-equal reference scores and compare each run as a set;
        # the boundary run may exchange members with items just outside the top-k
        rs = ref_scores[b]
        start = 0
        for j in range(1, k + 1):
            if j == k or (rs[j - 1] - rs[j]) > tie_tol:
                a, r = set(ids[b, start:j].tolist()), set(ref_ids[b, start:j].tolist())
                if a != r:
                    assert j == k, f"row {b}: ids differ outside a tie group at [{start},{j})"
                    assert (rs[start] - rs[k - 1]) <= tie_tol
                start = j


def tie_branch_census(ids, ref_scores, ref_ids, tie_tol):
    """How much of an id comparison rests on the tie rule: rows / positions where the returned id differs from the reference's, and how
    many of those positions sit inside a run of reference scores closer than `tie_tol` (the rest would be real disagreements).
    "Bit-exact indices modulo ties" means: positions_differing is tiny and positions_outside_tie_runs is 0."""
    ids, ref_scores, ref_ids = (torch.as_tensor(x).cpu() for x in (ids, ref_scores, ref_ids))
    B, k = ids.shape
    diff = ids != ref_ids
    gaps = (ref_scores[:, :-1] - ref_scores[:, 1:]) > tie_tol               # True: a run boundary between j and j + 1
    run = torch.cat([torch.zeros((B, 1), dtype=torch.int64), gaps.to(torch.int64).cumsum(1)], 1)   # run index of every position
    outside = 0
    for b in diff.any(1).nonzero().flatten().tolist():
        for r in run[b][diff[b]].unique().tolist():
            sel = run[b] == r
            last_run = bool(sel[-1])
            if set(ids[b][sel].tolist()) != set(ref_ids[b][sel].tolist()) and not last_run:
                outside += int((diff[b] & sel).sum())
    return {"rows": B, "k": k, "rows_differing": int(diff.any(1).sum()), "positions_differing": int(diff.sum()),
            "positions_outside_tie_runs": outside, "tie_tol": tie_tol}


def variant_cases():
    """(name, cfg, weights, arrays) of tests/golden/variants.npz (oracle/gen_golden_variants.py: the reference on model
    variants and shapes beyond the BASELINE configs)."""
    z = np.load(os.path.join(GOLDEN, "variants.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        d = json.loads(str(z[f"{name}/cfg_json"]))
        d["uid_embedding_hash_sizes"] = tuple(d["uid_embedding_hash_sizes"])
        cfg = MoLConfig(**d)
        w = {k[len(name) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "/w/")}
        arrays = {k[len(name) + 1:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(name + "/") and "/w/" not in k and not k.endswith("cfg_json")}
        yield name, cfg, w, arrays
