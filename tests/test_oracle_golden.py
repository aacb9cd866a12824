"""CPU: the oracle (oracle/mol_oracle.py) against the golden vectors the reference produced."""
import numpy as np
import pytest
import torch

from oracle import mol_oracle as O
from tests._fixtures import PER_CONFIG, Fixture, assert_topk_matches, full_size_inputs


@pytest.fixture(scope="module", params=PER_CONFIG)
def fx(request):
    return Fixture(request.param)


def test_f1_every_stage(fx):
    n = int(fx.z["F1/n"])
    st = O.mol_stages(fx.cfg, fx.weights, fx.t("q"), fx.t("X")[:, :n], fx.user_ids)
    for key in ("Eq", "Ex", "gq", "gi", "cl", "gqi", "w", "pi", "logits"):
        ref = fx.t("F1/" + key)
        got = st[key].reshape(ref.shape)
        assert torch.equal(got, ref), f"{fx.name}:{key} max|d|={float((got - ref).abs().max())}"


def test_f2_brute_force_topk(fx):
    X, ids = fx.t("X"), fx.t("item_ids")
    N = X.shape[1]
    logits = O.mol_logits(fx.cfg, fx.weights, fx.t("q"), X, fx.user_ids, chunk=300)
    ref = fx.t("F2/all_logits")
    # chunking changes the GEMM blocking inside torch, not the math
    assert torch.allclose(logits, ref, atol=2e-6, rtol=0)
    for k in (10, 200, N):
        s, i, _ = O.brute_force_topk(fx.cfg, fx.weights, fx.t("q"), X, ids, k, fx.user_ids, chunk=N)
        assert_topk_matches(s, i, fx.t(f"F2/k{k}/scores"), fx.t(f"F2/k{k}/ids"), atol=2e-6)
        # the build's deterministic tie rule agrees with torch.topk wherever scores are distinct
        ds, di = O.select_topk_deterministic(ref, k)
        assert_topk_matches(ds, ids.reshape(-1)[di], fx.t(f"F2/k{k}/scores"), fx.t(f"F2/k{k}/ids"), atol=0)


@pytest.mark.parametrize("case", ["timing", "accuracy", "backfill", "nofilter"])
def test_f3_candidate_index(fx, case):
    X, ids = fx.t("X"), fx.t("item_ids")
    k = int(fx.z[f"F3/{case}/k"])
    trunc = int(fx.z[f"F3/{case}/truncate"])
    trunc = None if trunc < 0 else trunc
    inv = fx.t(f"F3/{case}/invalid_ids") if fx.has(f"F3/{case}/invalid_ids") else None
    kp = O.k_prime(k, inv, X.shape[1], trunc)
    s, i, _ = O.brute_force_topk(fx.cfg, fx.weights, fx.t("q"), X, ids, kp, fx.user_ids, chunk=X.shape[1])
    out_ids, out_scores = O.filter_seen_ids(i, s, inv, k)
    assert_topk_matches(out_scores, out_ids, fx.t(f"F3/{case}/scores"), fx.t(f"F3/{case}/ids"), atol=2e-6)
    if case == "backfill":
        # the fixture really exercises the back-fill branch: some returned ids are seen ids
        seen = (out_ids.unsqueeze(2) == inv.unsqueeze(1)).any(2)
        assert bool(seen.any())


@pytest.mark.parametrize("avg_k", [100, 500])
def test_f4_avg_topk(fx, avg_k):
    X, ids = fx.t("X"), fx.t("item_ids")
    coarse = O.avg_topk_coarse_scores(fx.cfg, fx.weights, fx.t("q"), X, fx.user_ids)
    assert torch.equal(coarse.float(), fx.t(f"F4/a{avg_k}/coarse_scores_bf16_as_f32"))
    s, i, _ = O.avg_topk(fx.cfg, fx.weights, fx.t("q"), X, ids, 50, avg_k, fx.user_ids,
                         coarse_idx=fx.t(f"F4/a{avg_k}/coarse_idx_forward"))
    assert_topk_matches(s, i, fx.t(f"F4/a{avg_k}/scores"), fx.t(f"F4/a{avg_k}/ids"), atol=2e-6)
    assert bool(fx.z[f"F4/a{avg_k}/raises_when_k_gt_avg"])
    with pytest.raises(ValueError):
        O.avg_topk(fx.cfg, fx.weights, fx.t("q"), X, ids, avg_k + 1, avg_k, fx.user_ids)


def test_f6_per_row_candidates(fx):
    X = fx.t("X").squeeze(0)
    cand = X[fx.t("F6/cand_idx")]
    got = O.mol_stages(fx.cfg, fx.weights, fx.t("q"), cand, fx.user_ids)["logits"]
    assert torch.equal(got, fx.t("F6/logits"))


def test_f5_harness_metrics():
    fx = Fixture("harness")
    X, ids, q = fx.t("X"), fx.t("item_ids"), fx.t("q")
    past, target = fx.t("past_ids"), fx.t("target_ids")
    N = X.shape[1]
    for mode, max_k, trunc in (("accuracy", 2500, None), ("timing", 120, 200)):
        k = min(max_k, N)
        kp = O.k_prime(k, past, N, trunc)
        s, i, _ = O.brute_force_topk(fx.cfg, fx.weights, q, X, ids, kp, None, chunk=N)
        top_ids, _ = O.filter_seen_ids(i, s, past, k)
        ref_ids = fx.t(f"F5/{mode}/eval_top_k_ids")
        m = O.eval_metrics(top_ids, target, max_k)
        # ranks depend on ids only; allow tie-order differences to move nothing (checked by equality)
        assert torch.equal(top_ids, ref_ids)
        for key in ("hr@1", "hr@5", "hr@10", "hr@50", "hr@100", "hr@200", "hr@500", "hr@1000"):
            assert torch.equal(m[key], fx.t(f"F5/{mode}/{key}")), key
        for key in ("ndcg@1", "ndcg@5", "ndcg@10", "ndcg@50", "ndcg@100", "ndcg@200", "mrr"):
            assert torch.allclose(m[key].float(), fx.t(f"F5/{mode}/{key}").float(), atol=1e-7), key
    # the fixture covers hits, misses and seen-and-filtered targets
    r = O.eval_ranks(fx.t("F5/timing/eval_top_k_ids"), target, 120)
    assert int((r == 121).sum()) > 0 and int((r <= 10).sum()) > 0


@pytest.mark.parametrize("name", ["full_c1_ml1m", "full_c2_ml20m", "full_c3_books"])
def test_f7_full_size(name):
    """the oracle against the reference's own output at the corpora's full sizes (C3: all 695 762 items of the headline workload, 8 queries)"""
    fx = Fixture(name)
    X, ids = full_size_inputs(fx)
    s, i, logits = O.brute_force_topk(fx.cfg, fx.weights, fx.t("q"), X, ids, 200, fx.user_ids, chunk=4096 if X.shape[1] < 100_000 else 65536)
    assert_topk_matches(s, i, fx.t("scores"), fx.t("ids"), atol=3e-6)
    assert np.allclose(logits.double().sum(1).numpy(), fx.z["logits_rowsum_f64"], rtol=0, atol=2e-2 * max(1.0, X.shape[1] / 27278))
    if "logits_first_row" in fx.z:
        assert torch.allclose(logits[0], fx.t("logits_first_row"), atol=3e-6, rtol=0)


def test_hash_item_table_is_shardable():
    a = O.hash_item_table(7, 0, 1000, 64)
    b = O.hash_item_table(7, 400, 100, 64)
    assert np.array_equal(a[400:500], b)
    assert abs(float(a.std()) - 0.02) < 5e-4 and float(np.abs(a).max()) <= 0.02 * 3.47


def _mips_inputs(z, tag):
    n, D = int(z[f"{tag}/N"]), z[f"{tag}/q"].shape[1]
    X = torch.from_numpy(O.hash_item_table(int(z[f"{tag}/table_seed"]), 0, n, D, sigma=1.0)).unsqueeze(0)
    g = torch.Generator().manual_seed(int(z[f"{tag}/ids_seed"]))
    ids = torch.cumsum(torch.randint(1, 4, (n,), generator=g), 0).to(torch.int64).unsqueeze(0)
    return torch.from_numpy(z[f"{tag}/q"]), X, ids


def test_f9_mips_and_dot_product():
    import os

    from tests._fixtures import GOLDEN

    z = np.load(os.path.join(GOLDEN, "mips.npz"))
    T = lambda k: torch.from_numpy(z[k])
    for tag in ("d50", "d64"):
        q, X, ids = _mips_inputs(z, tag)
        assert torch.equal(O.dot_product_similarity(q, X)[:2], T(f"{tag}/logits_head"))
        for k in (10, 200):
            s, i = O.mips_brute_force_topk(q, X, ids, k)
            assert torch.equal(s, T(f"{tag}/k{k}/scores")) and torch.equal(i, T(f"{tag}/k{k}/ids"))
    assert torch.equal(O.dot_product_similarity(T("rows/q1"), T("rows/X")), T("rows/out1"))
    assert torch.equal(O.dot_product_similarity(T("rows/q3"), T("rows/X")), T("rows/out3"))


def _union_case(z, cname):
    import json

    d = json.loads(str(z[f"{cname}/cfg_json"]))
    d["uid_embedding_hash_sizes"] = tuple(d["uid_embedding_hash_sizes"])
    cfg = O.MoLConfig(**d)
    w = {k[len(cname) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{cname}/w/")}
    for i, hs in enumerate(cfg.uid_embedding_hash_sizes):
        key = f"_query_embeddings_fn._uid_embeddings_{i}.weight"
        rows = w.pop(key + ".rows")
        full = torch.zeros((hs + 1, cfg.dot_product_dimension))
        full[rows] = w[key]
        w[key] = full
    T = lambda k: torch.from_numpy(z[f"{cname}/{k}"])
    uid = T("user_ids") if f"{cname}/user_ids" in z.files else None
    return cfg, w, T, uid


@pytest.mark.parametrize("cname", ["c1", "c3"])
def test_f10_naive_and_comb_rerank(cname):
    import os

    from tests._fixtures import GOLDEN

    z = np.load(os.path.join(GOLDEN, "union.npz"))
    cfg, w, T, uid = _union_case(z, cname)
    for mname, width in (("naive5", cfg.num_logits * 5), ("comb5_100", cfg.num_logits * 5 + 100)):
        idx = T(f"{mname}/sorted_all_indices")
        assert idx.shape[1] == width                          # the reference returns ALL candidates, not k
        s, i = O.union_rerank(cfg, w, T("q"), T("X"), T("item_ids"), idx, uid)
        assert_topk_matches(s, i, T(f"{mname}/scores"), T(f"{mname}/ids"), atol=2e-6)
        assert bool((T(f"{mname}/scores")[:, -1] == -32767.0).all())   # duplicates exist and sink to the end
    # the candidate generator: every index of the naive union is among the per-pair top-5 of the bf16 component scores
    cs = O.component_candidate_scores(cfg, w, T("q"), T("X"), uid).float()      # (B, P_Q, P_X, N)
    kth = torch.topk(cs, 5, dim=-1).values[..., -1:]                             # 5th best per (b, i, m)
    allowed = (cs >= kth).any(1).any(1)                                          # (B, N): item reachable by some pair
    idx = T("naive5/sorted_all_indices")
    assert bool(torch.gather(allowed, 1, idx).all())


def test_oracle_reproduces_the_reference_on_model_variants():
    """Plain-Linear query projection, GLU item projection, combination "none", H = 64, new shapes (variants.npz)."""
    from tests._fixtures import variant_cases

    n = 0
    for name, cfg, w, a in variant_cases():
        st = O.mol_stages(cfg, w, a["q"], a["X"])
        assert torch.equal(st["logits"], a["logits"]) and torch.equal(st["Eq"], a["Eq"]) and torch.equal(st["Ex"].reshape(a["Ex"].shape), a["Ex"]), name
        rows = O.mol_stages(cfg, w, a["q"], a["cand"])["logits"]
        assert float((rows - a["row_logits"]).abs().max()) <= 2e-6, name
        n += 1
    assert n == 5


def test_oracle_glu_matches_the_reference_layers():
    """oracle._glu (the restatement used inside the query / item projections) against the reference's GeGLU / SwiGLU modules run
    on their own (tests/golden/glu.npz)."""
    import os

    import numpy as np

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glu.npz"))
    for case in ("geglu_2d", "swiglu_2d", "geglu_3d", "swiglu_1row"):
        x, w, b, y = (torch.from_numpy(z[f"{case}.{k}"]) for k in ("x", "w", "b", "y"))
        got = O._glu(x.reshape(-1, x.shape[-1]), w, b, "geglu" if case.startswith("geglu") else "swiglu").reshape(y.shape)
        assert torch.equal(got, y), case
