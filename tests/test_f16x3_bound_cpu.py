"""CPU property tests of the a-priori bound of the proved exact top-k (rails_amd/f16x3_bound.py):
  * the product's bound == the oracle's independent restatement, term by term, on every BASELINE shape and every stress family;
  * |emulated fp32 kernel - float64| <= eps32 and |emulated f16x3 kernel (the measured MFMA datapath) - float64| <= eps16, for the logit
    AND for every intermediate stage (cl, t, q) against the stage bounds the derivation goes through;
  * the building blocks: the operand-split constants, phi's Lipschitz constant, the softmax-mixture Lipschitz bound.
The same families run on the GPU against the real kernels (tests/test_proved_gpu.py)."""
import dataclasses
import math

import numpy as np
import pytest
import torch

from oracle import f16x3_bound as OB
from oracle import mol_oracle as O
from rails_amd import f16x3_bound as FB

SHAPES = ["amzn-books", "ml-1m", "ml-20m", "synthetic-16x16x64", "synthetic-8x8x32"]
STRESS = ["gaussian", "outlier", "hot gate", "near overflow", "tiny components"]


def _case(workload: str, kind: str, seed: int):
    cfg = O.CONFIGS[workload]
    if cfg.uid_embedding_hash_sizes:
        cfg = dataclasses.replace(cfg, uid_embedding_hash_sizes=(63,))
    w = O.synthetic_weights(cfg, seed=seed, uid_rows=64 if cfg.uid_embedding_hash_sizes else None)
    w, item_scale = OB.stress_weights(w, kind, seed)
    p = "_gating_fn._qi_partial_module."
    args = (w[p + "1.weight"], w[p + "1.bias"], w[p + "3.weight"], w[p + "3.bias"], cfg.temperature, cfg.dot_product_dimension,
            cfg.query_dot_product_groups, cfg.item_dot_product_groups)
    return cfg, w, item_scale, args


@pytest.mark.parametrize("kind", STRESS)
@pytest.mark.parametrize("workload", SHAPES)
def test_product_bound_equals_the_restatement(workload, kind):
    _, _, _, args = _case(workload, kind, seed=3)
    a = FB.first_pass_bound(*args)
    b = OB.first_pass_bound(*(np.asarray(t) if torch.is_tensor(t) else t for t in args))
    if math.isinf(a["eps"]):
        assert math.isinf(b["eps"])
    for key in ("eps", "eps16", "eps32", "d_cl16", "d_cl32", "d_t16", "d_t32", "d_q16", "d_q32", "d_w16", "d_w32"):
        if key in a and math.isfinite(a[key]):
            assert a[key] == pytest.approx(b[key], rel=1e-9), key
    assert a["in_f16_range"] == b["in_f16_range"]


@pytest.mark.parametrize("kind", STRESS)
@pytest.mark.parametrize("workload", SHAPES)
def test_emulated_arithmetics_stay_inside_the_bound(workload, kind):
    cfg, w, item_scale, args = _case(workload, kind, seed=4)
    big = cfg.num_logits > 64
    B, X = (2, 6) if big else (4, 24)
    q = O.synthetic_queries(cfg, B, seed=7)
    items = torch.from_numpy(O.hash_item_table(9, 0, X, cfg.item_embedding_dim)) * item_scale
    uid = torch.arange(1, B + 1) if cfg.uid_embedding_hash_sizes else None
    eqp, ex, gqp, gi = OB.pair_operands(cfg, w, q, items, uid)
    w1p, b1p, w2, b2p = OB.prescale(*(np.asarray(t) for t in args[:4]))
    bound = OB.first_pass_bound(*(np.asarray(t) if torch.is_tensor(t) else t for t in args))
    assert float(np.abs(gqp * gi).max()) <= OB.GATE_GUARD          # the data-dependent guard holds for these inputs
    ref = OB.exact64(eqp, ex, gqp, gi, w1p, b1p, w2, b2p)
    e32 = OB.emulate_fp32(eqp, ex, gqp, gi, w1p, b1p, w2, b2p, cfg.query_dot_product_groups, cfg.item_dot_product_groups, seed=1)
    e16 = OB.emulate_f16x3(eqp, ex, gqp, gi, w1p, b1p, w2, b2p, cfg.query_dot_product_groups, cfg.item_dot_product_groups, seed=1)
    if not bound["in_f16_range"]:
        pytest.skip("operands leave the f16 range: the bound is infinite and the product does not speculate")
    rows = []
    for name, emu, tag in (("fp32", e32, "32"), ("f16x3", e16, "16")):
        d_cl = float(np.abs(emu["cl"] - ref["cl"]).max())
        d_t = np.abs(emu["t"] - ref["t"]).max(0)
        d_q = np.abs(emu["q"] - ref["q"]).max(0)
        d_s = float(np.abs(emu["s"] - ref["s"]).max())
        rows.append((name, d_cl / bound["d_cl" + tag], float((d_t / bound["per_h"]["d_t" + tag]).max()), float((d_q / bound["per_l"]["d_q" + tag]).max()), d_s / bound["eps" + tag]))
        assert np.isfinite(emu["s"]).all()
        assert d_cl <= bound["d_cl" + tag]
        assert (d_t <= bound["per_h"]["d_t" + tag]).all()
        assert (d_q <= bound["per_l"]["d_q" + tag]).all()
        assert d_s <= bound["eps" + tag]
    assert float(np.abs(e16["s"] - e32["s"]).max()) <= bound["eps"]
    print(workload, kind, "observed / bound (cl, t, q, s):", [(n, f"{a:.2e}", f"{b:.2e}", f"{c:.2e}", f"{d:.2e}") for n, a, b, c, d in rows], "eps", round(bound["eps"], 4))


@pytest.mark.parametrize("kind", STRESS)
@pytest.mark.parametrize("workload", SHAPES)
def test_per_pair_bound_holds_stage_by_stage(workload, kind):
    """first_pass_bound(cl_max = c) is stated for the pairs whose cross logits satisfy |cl_l| <= c: for single pairs (the smallest, the
    median and the largest c of the sample, and a few more) every stage error of both emulated arithmetics stays inside the stage bounds
    evaluated at THAT pair's own c -- the statement the per-pair upper bound of the first pass (upper_bound_poly) rests on."""
    cfg, w, item_scale, args = _case(workload, kind, seed=6)
    big = cfg.num_logits > 64
    B, X = (2, 6) if big else (4, 24)
    q = O.synthetic_queries(cfg, B, seed=8)
    items = torch.from_numpy(O.hash_item_table(11, 0, X, cfg.item_embedding_dim)) * item_scale
    uid = torch.arange(1, B + 1) if cfg.uid_embedding_hash_sizes else None
    eqp, ex, gqp, gi = OB.pair_operands(cfg, w, q, items, uid)
    w1p, b1p, w2, b2p = OB.prescale(*(np.asarray(t) for t in args[:4]))
    np_args = tuple(np.asarray(t) if torch.is_tensor(t) else t for t in args)
    if not OB.first_pass_bound(*np_args)["in_f16_range"]:
        pytest.skip("operands leave the f16 range: the bound is infinite and the product does not speculate")
    ref = OB.exact64(eqp, ex, gqp, gi, w1p, b1p, w2, b2p)
    e32 = OB.emulate_fp32(eqp, ex, gqp, gi, w1p, b1p, w2, b2p, cfg.query_dot_product_groups, cfg.item_dot_product_groups, seed=2)
    e16 = OB.emulate_f16x3(eqp, ex, gqp, gi, w1p, b1p, w2, b2p, cfg.query_dot_product_groups, cfg.item_dot_product_groups, seed=2)
    c_pair = np.abs(ref["cl"]).max(1)
    order = np.argsort(c_pair)
    pick = sorted({int(order[0]), int(order[1]), int(order[len(order) // 2]), int(order[-2]), int(order[-1])})
    worst = 0.0
    for i in pick:
        b = OB.first_pass_bound(*np_args, cl_max=float(c_pair[i]))
        top = OB.first_pass_bound(*np_args)
        assert b["eps"] <= top["eps"] * (1 + 1e-12)
        for emu, tag in ((e32, "32"), (e16, "16")):
            assert abs(emu["cl"][i] - ref["cl"][i]).max() <= b["d_cl" + tag]
            assert (np.abs(emu["t"][i] - ref["t"][i]) <= b["per_h"]["d_t" + tag]).all()
            assert (np.abs(emu["q"][i] - ref["q"][i]) <= b["per_l"]["d_q" + tag]).all()
            assert abs(emu["s"][i] - ref["s"][i]) <= b["eps" + tag]
            worst = max(worst, float((np.abs(emu["t"][i] - ref["t"][i]) / b["per_h"]["d_t" + tag]).max()))
        assert abs(e16["s"][i] - e32["s"][i]) <= b["eps"]
    print(workload, kind, "c of the pairs", [round(float(c_pair[i]), 3) for i in pick], "worst observed / bound of t:", f"{worst:.2e}")


@pytest.mark.parametrize("workload,kind", [("ml-1m", "gaussian"), ("ml-1m", "outlier"), ("amzn-books", "hot gate"), ("synthetic-16x16x64", "gaussian")])
def test_upper_bound_poly_covers_the_restated_bound(workload, kind):
    """The coefficients the product hands to the UPPER first pass: non-negative float32s whose quadratic, as the device evaluates it, covers the
    oracle's own eps(c + d_cl16) over the whole range of c (oracle.upper_poly_shortfall restates the requirement); tight (within a few % of
    eps at the a-priori end, and far below it where the pairs of a corpus sit); and on emulated pairs first pass + P(c16) >= fp32 logit."""
    cfg, w, item_scale, args = _case(workload, kind, seed=3)
    np_args = tuple(np.asarray(t) if torch.is_tensor(t) else t for t in args)
    res = FB.upper_bound_poly(*args)
    top = OB.first_pass_bound(*np_args)
    if not math.isfinite(top["eps"]):
        assert res["poly"] is None
        return
    ub2, ub1, ub0 = res["poly"]
    assert min(ub2, ub1, ub0) >= 0 and all(float(np.float32(v)) == v for v in res["poly"])
    assert OB.upper_poly_shortfall(res["poly"], *np_args, grid=res["grid"]) <= 0.0
    assert OB.upper_poly_shortfall(tuple(v * 0.97 for v in res["poly"]), *np_args, grid=res["grid"]) > 0.0       # (the check can fail)
    c_top = res["c_top"]
    p_top = (ub2 * c_top + ub1) * c_top + ub0
    assert top["eps"] <= p_top <= 1.06 * top["eps"] + 1e-4
    c_typ = 0.375 * c_top
    assert (ub2 * c_typ + ub1) * c_typ + ub0 <= 0.25 * top["eps"]
    # emulated pairs
    big = cfg.num_logits > 64
    B, X = (2, 6) if big else (4, 24)
    q = O.synthetic_queries(cfg, B, seed=9)
    items = torch.from_numpy(O.hash_item_table(13, 0, X, cfg.item_embedding_dim)) * item_scale
    uid = torch.arange(1, B + 1) if cfg.uid_embedding_hash_sizes else None
    eqp, ex, gqp, gi = OB.pair_operands(cfg, w, q, items, uid)
    w1p, b1p, w2, b2p = OB.prescale(*(np.asarray(t) for t in args[:4]))
    e32 = OB.emulate_fp32(eqp, ex, gqp, gi, w1p, b1p, w2, b2p, cfg.query_dot_product_groups, cfg.item_dot_product_groups, seed=3)
    e16 = OB.emulate_f16x3(eqp, ex, gqp, gi, w1p, b1p, w2, b2p, cfg.query_dot_product_groups, cfg.item_dot_product_groups, seed=3)
    c16 = np.abs(e16["cl"]).max(1).astype(np.float32)
    f = np.float32
    p32 = (f(ub2) * c16 + f(ub1)).astype(np.float32) * c16 + f(ub0)
    upper = (e16["s"].astype(np.float32) + p32.astype(np.float32)).astype(np.float32)
    assert (upper.astype(np.float64) >= e32["s"]).all()
    assert float((upper - e32["s"]).max()) <= p_top


def test_device_self_check_accepts_the_model_and_rejects_departures_from_it():
    """rails_amd/arith_check.py (run once per device before the proved mode binds) with emulated probes: arithmetic that follows the model
    passes; an f16 MFMA that rounds its result to f16, one that flushes f16 subnormal operands, an fp32 MFMA accumulating in bf16-like
    precision and a 3-ulp exp each fail it."""
    from rails_amd import arith_check as AC

    def p16(a, b, c): return (c.double() + a.double() @ b.double()).float()
    def p32(a, b, c): return (c.double() + a.double() @ b.double()).float()

    def ps(x):
        x64 = x.double()
        return torch.stack([torch.exp2(x64), 1 / x64, x64 / (1 + torch.exp2(x64))]).float()

    cpu = torch.device("cpu")
    good = AC.measure(cpu, p16, p32, ps)
    assert good["ok"] and good["h2_f16_mfma_kc_kp"] < 0.5 and good["h2_f16_subnormals_kept"]
    assert not AC.measure(cpu, lambda a, b, c: (c + (a.float() @ b.float()).half().float()), p32, ps)["ok"]
    flush = lambda t: torch.where(t.abs().float() < 2.0 ** -14, torch.zeros_like(t), t)      # noqa: E731
    r = AC.measure(cpu, lambda a, b, c: p16(flush(a), flush(b), c), p32, ps)
    assert not r["ok"] and not r["h2_f16_subnormals_kept"]
    assert not AC.measure(cpu, p16, lambda a, b, c: p32(a, b, c).bfloat16().float(), ps)["ok"]

    def ps_bad(x):
        out = ps(x)
        out[0] = out[0] * (1 + 3 * 2.0 ** -23)
        return out

    assert not AC.measure(cpu, p16, p32, ps_bad)["ok"]


def test_operand_split_constants():
    """|x - hi - lo| <= R |x| + A and |lo| <= LAM |x| + A' for both split flavours, over normal, tiny (f16-subnormal) and large values"""
    g = np.random.default_rng(0)
    x = np.concatenate([g.standard_normal(200_000) * s for s in (1e-7, 1e-5, 1e-3, 1.0, 50.0, 6.0e4)]).astype(np.float32)
    x = x[np.abs(x) < 65000]
    for kernel in (True, False):
        hi, lo = OB.split_f16(x, kernel)
        r, a, lam, a2 = OB._split_constants(kernel)
        x64 = x.astype(np.float64)
        assert (np.abs(x64 - hi - lo) <= r * np.abs(x64) + a).all()
        assert (np.abs(lo) <= lam * np.abs(x64) + a2).all()
        assert (np.abs(hi) <= np.abs(x64)).all()


def test_phi_lipschitz_and_magnitude():
    t = np.linspace(-200.0, 200.0, 2_000_001)
    phi = OB._phi64(t)
    assert np.abs(np.diff(phi) / np.diff(t)).max() <= OB.LIP
    assert (np.abs(phi) <= np.abs(t) + 1e-300).all()


def test_mixture_lipschitz_bound():
    """|f(w + delta, c) - f(w, c)| <= max |delta| (max c - min c) / 2 for f = sum softmax(w) c: random and adversarial (two-cluster) cases"""
    g = np.random.default_rng(1)
    worst = 0.0
    for trial in range(4000):
        L = int(g.integers(2, 65))
        w = g.standard_normal(L) * g.choice([0.1, 1.0, 10.0])
        c = g.standard_normal(L) * 20.0
        a = 10.0 ** g.uniform(-6, 0.5)
        delta = g.choice([-a, a], L) if trial % 2 else g.uniform(-a, a, L)
        if trial % 4 == 3:       # extremal: delta follows the sign of c - median, weights split evenly between two atoms
            c = np.where(np.arange(L) % 2 == 0, -20.0, 20.0)
            w = np.zeros(L)
            delta = np.where(c > 0, a, -a)
        sm = lambda v: np.exp(v - v.max()) / np.exp(v - v.max()).sum()
        lhs = abs((sm(w + delta) * c).sum() - (sm(w) * c).sum())
        rhs = a * (c.max() - c.min()) / 2
        worst = max(worst, lhs / rhs)
        assert lhs <= rhs * (1 + 1e-9)
    assert worst > 0.5      # the bound is approached: it is not vacuous


def test_the_bound_runs_on_few_threads_and_restores_the_callers_setting():
    """first_pass_bound / upper_bound_poly limit OpenMP to four threads while they run (a bind in one-rank-per-GPU processes launched without
    OMP_NUM_THREADS took 10-15 s on a 256-thread host) and put the caller's thread count back -- also when the body raises."""
    import torch
    from rails_amd import f16x3_bound as FB

    seen = []

    @FB._few_cpu_threads
    def body(fail):
        seen.append(torch.get_num_threads())
        if fail:
            raise RuntimeError("x")
        return 7

    before = torch.get_num_threads()
    try:
        torch.set_num_threads(6)
        assert body(False) == 7 and seen[-1] == 4 and torch.get_num_threads() == 6
        with pytest.raises(RuntimeError):
            body(True)
        assert torch.get_num_threads() == 6
        torch.set_num_threads(2)
        assert body(False) == 7 and seen[-1] == 2 and torch.get_num_threads() == 2
    finally:
        torch.set_num_threads(before)
