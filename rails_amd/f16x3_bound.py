"""A-PRIORI bound on |first pass - fp32 logit| of the verified exact top-k ("proved mode", topk_modules.MoLBruteForceTopK).

The first pass of precision "f16x3-exact" scores every (query, item) pair with the split-f16 kernels (csrc/mol_score_f16_unit.h,
mol_score_wsplit_f16.h); the candidates it picks are re-scored by the fp32 kernels (csrc/mol_score_fp32_unit.h, mol_score_small.hip,
mol_score_wsplit.h), whose bits are what the module returns.  The result IS the dense fp32 top-k whenever

        e_k  >  m + eps,        |s16(b, x) - s32(b, x)| <= eps  for EVERY pair (b, x)

(e_k: k-th fp32 score among the candidates, m: smallest first-pass score among them; rails_rescore_select).  This file computes
such an eps from the pair-gate weights alone.  Both kernels evaluate the same real function of the same fp32 inputs
(Eq / tau, Ex, gq', gi, W1', b1', W2, b2': the packs are written from the same fp32 values by the same prologue / index build;
reference: rails/similarities/mol/similarity_fn.py:389-413, gate :148-201, combiner :31-46)

    cl_l  = sum_d Eq'[p, d] Ex[m, d]                       l = (p, m)
    t_h   = b1'_h + sum_l W1'[h, l] cl_l                   W1' = -log2e W1, b1' = -log2e b1  (mol_layout.h)
    hid_h = phi(t_h),  phi(t) = t / (1 + 2^t)              = -log2e silu(pre)
    q_l   = b2'_l + sum_h W2[l, h] hid_h
    t2_l  = gq'_l gi_l + q_l ;  u_l = phi(t2_l)            = -log2e * w_l
    s     = sum_l softmax(w)_l cl_l                        (the eval-time renormalisation divides by sum pi = 1)

so eps = eps16 + eps32 with |s16 - s| <= eps16 and |s32 - s| <= eps32, s the exact real value.

Arithmetic model (u = 2^-24; every hypothesis is exercised by a test):
  H1  v_mfma_f32_32x32x2_f32 / 16x16x4_f32 are chains of round-to-nearest fmaf: a length-n chain started at c0 returns
      c0 + sum a_j b_j + e with |e| <= gamma(n) |c0| + sum_j gamma(n - j + 1) |a_j b_j| (j = 1 is accumulated first; gamma(n) =
      n u / (1 - n u)): a term is rounded once by its own fmaf and once by every later one.  The order of the chains is the
      layout's (mol_layout.h: K-step e of GEMM2 consumes the logits logit_of(e, 0), logit_of(e, 1), K-step f of GEMM3 the hidden
      units hidden_of(f, 0), hidden_of(f, 1); every fp32 shell visits them in this order -- they return the same bits); the order
      of the two products INSIDE one instruction is not assumed: the larger one is charged the larger factor.
  H2  v_mfma_f32_32x32x16_f16 returns C + sum_16 a_i b_i + e with |e| <= KC u (|C| + sum |a_i b_i|) + KP u sum |a_i b_i|.  Products of f16 are
      exact; what the instruction loses was measured on the part (tools/r05_probe2.py, tests/test_proved_gpu.py::
      test_f16_mfma_accumulation_model): every addend is cut (toward zero) below 2^-26 of the largest addend's binade -- 17 addends,
      < 0.25 u each relative to |C| + sum |p| -- a product is also cut below 2^-24 of the largest product of ITS lane half's eight
      (< u of that product each, seven per half), and the sum is rounded to nearest (u).  That is 5 u and 7 u; KC = 6 and KP = 8 leave a margin, and the test fails when an
      operand family built to hit these cases exceeds 0.8 of the bound.  f16 subnormals are kept (same test).  An accumulator takes the
      three products of K-step s (lo*hi, hi*lo, hi*hi: 16 logits / hidden units each) as MFMAs 3s+1, 3s+2, 3s+3 of M = 3 K/16: the KC
      part of every instruction is relative to the running accumulator, so the hi*hi mass of K-step s is charged KC u by at most
      M - 3s - 1 instructions (one more when two partial accumulators are added at the end) and the two small products by at most
      M + 1; the KP part is charged once, to the products of the instruction itself.
  H3  v_exp_f32 and v_rcp_f32 are accurate to 1 ulp (relative 2 u); plain fp32 VALU arithmetic and the final division are
      correctly rounded (relative u; the TU is built without fast-math, Makefile).
  H4  flushed subnormal results cost an absolute 2^-126 per operation: carried as OMEGA per stage, ~1e-30, never visible.

Operand splits (mol_score_f16_unit.h split_pair; mol_index.hip / mol_query.hip split_f16): hi = RTZ_f16(x), r = x - hi exact in fp32,
lo = RTZ_f16(r) in the kernel (cl, hid) and RNE_f16(r) in the packs (Ex, Eq, W1', W2).  With f16 subnormals kept:
    |x - hi - lo| <= R |x| + A,   |lo| <= LAM |x| + A',      kernel: R = 2^-20, A = 2^-24     packs: R = 2^-21, A = 2^-25
A product block adds hi*hi + hi*lo + lo*hi; what is lost against a*b is a db + b da - da db + lo_a lo_b.

Propagation: phi has |phi'| <= 1.1 and |phi(t)| <= |t|; the mixture f(w, c) = sum softmax(w)_l c_l satisfies
|f(w + delta, c) - f(w, c)| <= max|delta| (max c - min c) / 2 (d/ds f(w + s delta, c) = Cov_pi(delta, c) and E|delta - E delta| <= max|delta|)
and |f(w, c') - f(w, c)| <= max|c' - c|.  Row-L1 norms of the weights carry magnitudes through the gate MLP.

Guards, without which the bound is infinite (the caller then does not speculate): dot_product_l2_norm (|cl| <= 1/tau), the glu_silu
combiner, a pair gate with hidden layer, finite weights inside the f16 range, and -- checked ON THE DEVICE per call
(rails_rescore_verdict's guard argument) -- max |gq'| max |gi| <= gate_guard, the one data-dependent magnitude of the chain
(the rounding of t2 is relative to |t2|).

oracle/f16x3_bound.py restates this bound and property-tests it against float64 evaluations of both arithmetics.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

U = 2.0 ** -24
KC = 6.0                 # H2: error of one f16 MFMA <= KC u (|C| + sum |p|) + KP u sum |p|
KP = 8.0
OMEGA = 2.0 ** -100
LOG2E_F32 = 1.4426950408889634  # the kernels' kLog2e literal; as a float it is 1.44269502162933349609375
LIP = 1.1                # sup |phi'| = 1.09984
GATE_GUARD = 256.0       # bound on |gq'| |gi| = log2e |gq gi| enforced on the device (|gq gi| <= 177)
F16_LIMIT = 60000.0

# split constants: (R, A, LAM, A') -- see the module docstring
_PACK = (2.0 ** -21, 2.0 ** -25, 2.0 ** -10 * (1 + 2.0 ** -11), 2.0 ** -24)
_KERN = (2.0 ** -20, 2.0 ** -24, 2.0 ** -10, 2.0 ** -24)


def _few_cpu_threads(fn):
    """The bound is a few hundred float64 operations on weight-sized tensors: with one OpenMP thread per logical CPU (the default in a process
    launched without OMP_NUM_THREADS, e.g. one rank per GPU under mp.spawn on a 256-thread host) every one of them is a fork / join of the whole
    pool and several ranks spin against each other -- a module's bind took 10-15 s instead of 0.3 (tests/test_sharded_gpu.py, round 6).  Run it
    on at most four threads and put the caller's setting back."""
    import functools

    @functools.wraps(fn)
    def run(*args, **kwargs):
        n = torch.get_num_threads()
        if n <= 4:
            return fn(*args, **kwargs)
        torch.set_num_threads(4)
        try:
            return fn(*args, **kwargs)
        finally:
            torch.set_num_threads(n)

    return run


def gamma(n: float, unit: float = U) -> float:
    return n * unit / (1.0 - n * unit)


def _block(sa, sb):
    """relative and absolute loss of one product block a*b with splits sa, sb:
    |a b - block| <= rho |a||b| + beta_b |a| + beta_a |b| + abs2,   accumulated magnitude <= mag |a||b| + ma |a| + mb |b|."""
    ra, aa, la, aa2 = sa
    rb, ab, lb, ab2 = sb
    rho = ra + rb + ra * rb + la * lb
    beta_b = ab * (1 + ra) + la * ab2       # multiplies |a|
    beta_a = aa * (1 + rb) + lb * aa2       # multiplies |b|
    abs2 = aa * ab + aa2 * ab2
    mag = 1 + la + lb
    return rho, beta_a, beta_b, abs2, mag, ab2, aa2


def _acc_row(reg: int, hi: int) -> int:
    return (reg & 3) + 8 * (reg >> 2) + 4 * hi


def logit_order(p_q: int, p_x: int):
    """column of W1' at chain position 2e + hi of GEMM2 (mol_layout.h logit_of)"""
    rpq = p_q // 2
    return [_acc_row(e % rpq, hi) * p_x + e // rpq for e in range(p_q * p_x // 2) for hi in (0, 1)]


def hidden_order(hidden: int):
    """column of W2 at chain position 2f + hi of GEMM3 (mol_layout.h hidden_of)"""
    return [32 * (f // 16) + _acc_row(f % 16, hi) for f in range(hidden // 2) for hi in (0, 1)]


def _chain32(terms: torch.Tensor, c0: torch.Tensor) -> torch.Tensor:
    """H1 for rows of |a_j b_j| bounds given in chain order (pairs (2e, 2e+1) share one instruction): (rows,) error bounds"""
    n = terms.shape[1]
    pair = terms.view(terms.shape[0], n // 2, 2)
    big, small = pair.max(2).values, pair.min(2).values
    j = torch.arange(n // 2, dtype=torch.float64)
    g_big = (n - 2 * j) * U / (1 - (n - 2 * j) * U)            # first of the pair: n - 2e roundings
    g_small = (n - 2 * j - 1) * U / (1 - (n - 2 * j - 1) * U)
    return gamma(n) * c0 + big @ g_big + small @ g_small


def _chain16(terms: torch.Tensor, c0: torch.Tensor, small_mass: torch.Tensor, kc: float, kp: float) -> torch.Tensor:
    """H2 for rows of |a b| bounds in chain order, 16 per K-step, three MFMAs per K-step: (rows,) accumulation-error bounds.
    small_mass: bound on the summed magnitude of the lo*hi and hi*lo products of a row."""
    n = terms.shape[1]
    nk = n // 16
    m = 3 * nk
    unit = kc * U
    step = terms.view(terms.shape[0], nk, 16).sum(2)
    s = torch.arange(nk, dtype=torch.float64)
    g = (m - 3 * s) * unit / (1 - (m - 3 * s) * unit)
    total = step.sum(1) + small_mass
    return gamma(m + 1, unit) * (c0 + small_mass) + step @ g + kp * U * total * (1 + gamma(m + 1, unit))


@_few_cpu_threads
def first_pass_bound(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, temperature: float, dot_dim: int,
                     p_q: int, p_x: int, kc: float = KC, kp: float = KP, gate_guard: float = GATE_GUARD,
                     cl_max: Optional[float] = None) -> Dict[str, float]:
    """eps with |first pass (f16x3) - fp32 kernel| <= eps for every pair, from the pair-gate weights (reference parameter names
    _gating_fn._qi_partial_module.{1,3}.{weight,bias}); infinite when a guard fails.  Also returns the two halves and the
    intermediate magnitudes (for the report and for the oracle's restatement to be compared term by term).
    cl_max: the bound holds for the pairs whose cross logits satisfy |cl_l| <= cl_max for every l (exact values; default and cap: the
    a-priori 1/tau of unit-norm sub-embeddings).  Only the magnitudes DOWNSTREAM of GEMM1 use it -- GEMM1's own error is relative to
    sum_d |Eq'||Ex|, which the norms bound and cl does not.  The bound is non-decreasing in cl_max (every coefficient is >= 0)."""
    f64 = torch.float64
    k32 = torch.tensor(-LOG2E_F32, dtype=torch.float32)
    w1p = (k32 * w1.detach().float().cpu()).to(f64).abs()          # |W1'| exactly as the pack kernels round it
    b1p = (k32 * b1.detach().float().cpu()).to(f64).abs()
    b2p = (k32 * b2.detach().float().cpu()).to(f64).abs()
    w2a = w2.detach().float().cpu().to(f64).abs()
    H, L = w1p.shape
    d = int(dot_dim)
    inv_tau = 1.0 / float(torch.tensor(temperature, dtype=torch.float32))
    out: Dict[str, float] = {"kc": kc, "kp": kp, "gate_guard": gate_guard, "eps": math.inf}
    if L != p_q * p_x or tuple(w2a.shape) != (L, H) or L % 32 or H % 32 or d % 16 or p_q % 2:
        return out
    if not (bool(torch.isfinite(w1p).all()) and bool(torch.isfinite(w2a).all()) and bool(torch.isfinite(b1p).all()) and bool(torch.isfinite(b2p).all())):
        return out
    slack = 1.0 + (d + 8) * U                     # fp32 l2 normalisation: ||Ex_m||_2 <= slack, ||Eq'_p||_2 <= slack / tau
    c0 = inv_tau * slack * slack                  # >= sum_d |Eq'||Ex| >= |cl|
    n1a, n1b = math.sqrt(d) * inv_tau * slack, math.sqrt(d) * slack
    cm = c0 if cl_max is None else min(c0, max(0.0, float(cl_max)))      # |cl_l| of the pairs the bound is stated for
    a1 = w1p.sum(1)                               # (H,) row L1 norms of W1'
    a2 = w2a.sum(1)                               # (L,)
    th = gamma(7)                                 # phi in floating point: exp2 (2u) + add (u) + rcp (2u) + mul (u)
    w1o = w1p[:, logit_order(p_q, p_x)]           # columns in chain order
    w2o_cols = hidden_order(H)

    def tail(dq: torch.Tensor, q_star: torch.Tensor, dcl: float, x1: float, n_sum: int) -> Dict[str, float]:
        """from the error dq_l of the pair-gate output q to the error of the logit"""
        t2 = gate_guard + q_star + dq                                    # |t2| in either arithmetic
        dt2 = dq + U * t2 * (1 + U)                                      # fma(gq', gi, q): one rounding
        du = LIP * dt2 + th * (t2 + dt2) + OMEGA
        umax = float(((t2 + dt2) * (1 + th)).max())
        nu = gamma(2) + gamma(n_sum) + math.log(2.0) * U * 2.0 * umax * (1 + U) + OMEGA   # exp2, the two sums, fl(mn - u)
        dw = math.log(2.0) * float(du.max()) + nu
        ds = dcl + dw * x1 + (math.expm1(2.0 * nu) * (1 + gamma(4)) + gamma(4)) * x1
        return {"ds": ds, "dw": dw, "du": float(du.max()), "nu": nu, "t2_max": float(t2.max())}

    # ---- the f16x3 first pass --------------------------------------------------------------------------------------------------
    rho, beta_a, beta_b, abs2, mag, a2b, a2a = _block(_PACK, _PACK)               # GEMM1: Eq' x Ex, both split by the packs
    g1 = c0 * mag + n1a * a2b + n1b * a2a                                         # (the distribution of the mass over k is unknown: flat factor)
    dcl16 = rho * c0 + beta_b * n1a + beta_a * n1b + d * abs2 + (gamma(3 * d / 16 + 1, kc * U) + kp * U * (1 + gamma(3 * d / 16 + 1, kc * U))) * g1
    x1 = cm + dcl16
    rho, beta_a, beta_b, abs2, mag, a2b, a2a = _block(_PACK, _KERN)               # GEMM2: W1' (pack) x cl (kernel split)
    s2 = a1 * x1
    small2 = s2 * (mag - 1) + a1 * a2b + L * x1 * a2a
    dt16 = a1 * dcl16 + rho * s2 + beta_b * a1 + beta_a * L * x1 + L * abs2 + _chain16(w1o * x1, b1p, small2, kc, kp)
    t_star = b1p + a1 * cm
    t16 = t_star + dt16
    dh16 = LIP * dt16 + th * t16 + OMEGA
    y16 = t16 * (1 + th)
    s3 = w2a @ y16                                                                # GEMM3: W2 (pack) x hid (kernel split)
    ysum = float(y16.sum())
    small3 = s3 * (mag - 1) + a2 * a2b + ysum * a2a
    dq16 = w2a @ dh16 + rho * s3 + beta_b * a2 + beta_a * ysum + H * abs2 + _chain16((w2a * y16)[:, w2o_cols], b2p, small3, kc, kp)
    q_star = b2p + w2a @ t_star
    in_range = max(x1, float(y16.max()), float(w1p.max()), float(w2a.max())) < F16_LIMIT
    half16 = tail(dq16, q_star, dcl16, x1, L // 2 + 8)

    # ---- the fp32 kernels ------------------------------------------------------------------------------------------------------
    dcl32 = gamma(d) * c0
    x1f = cm + dcl32
    dt32 = a1 * dcl32 + _chain32(w1o * x1f, b1p)
    t32 = t_star + dt32
    dh32 = LIP * dt32 + th * t32 + OMEGA
    y32 = t32 * (1 + th)
    dq32 = w2a @ dh32 + _chain32((w2a * y32)[:, w2o_cols], b2p)
    half32 = tail(dq32, q_star, dcl32, x1f, L // 2 + 8)

    eps = half16["ds"] + half32["ds"]
    out.update({
        "eps": eps if in_range else math.inf, "eps16": half16["ds"], "eps32": half32["ds"],
        "A1": float(a1.max()), "A2": float(a2.max()), "T_star": float(t_star.max()), "Q_star": float(q_star.max()),
        "d_cl16": dcl16, "d_cl32": dcl32, "d_t16": float(dt16.max()), "d_t32": float(dt32.max()),
        "d_q16": float(dq16.max()), "d_q32": float(dq32.max()), "d_w16": half16["dw"], "d_w32": half32["dw"],
        "t2_max": half16["t2_max"], "in_f16_range": bool(in_range),
    })
    return out


@_few_cpu_threads
def upper_bound_poly(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, temperature: float, dot_dim: int, p_q: int, p_x: int,
                     grid: int = 96, **kw) -> Dict[str, object]:
    """Coefficients (ub2, ub1, ub0), all >= 0 and float32-representable, of a PER-PAIR bound: for every pair whose first pass computed cross
    logits cl16 with c = max_l |cl16_l|,

        first pass logit + fl32((ub2 c + ub1) c + ub0), added in fp32   >=   the fp32 kernels' logit of the pair

    (the UPPER variant of the first pass writes the left-hand side: csrc/mol_score_wsplit.h, rails_mol_score_dense_upper).  Construction:
    eps(x) = first_pass_bound(cl_max = x)["eps"] is non-decreasing in x and holds for the pairs with |cl_l| <= x exactly; the pair's exact
    |cl_l| <= c + d_cl16 (GEMM1's own error, a-priori).  On a grid 0 = g_0 < ... < g_J covering every reachable c, E_j = eps(g_j + d_cl16)
    bounds the pairs with c <= g_j; a quadratic P with coefficients >= 0 (non-decreasing) that satisfies P(g_{j-1}) >= E_j for j = 1..J
    therefore satisfies P(c) >= E_j >= |first pass - fp32| for every c in (g_{j-1}, g_j] (and P(0) >= E_1 >= E_0).  P is a least-squares fit
    through (g_{j-1}, E_j) over the coefficient subsets that come out non-negative, shifted up by its largest shortfall.  The device
    evaluates P in fp32 (two fmas of non-negative terms: relative 2 u) and adds it to a logit of magnitude < 64 (half an ulp: 2^-18):
    the coefficients are inflated by 2^-18 relative, ub0 by 2^-17 absolute, and each is rounded UP to a float32.
    -> {"poly": (ub2, ub1, ub0) or None when the bound is infinite, "c_top", "eps_top" = eps at the a-priori |cl| <= 1/tau, "max_slack"}"""
    top = first_pass_bound(w1, b1, w2, b2, temperature, dot_dim, p_q, p_x, **kw)
    out: Dict[str, object] = {"poly": None, "eps_top": top["eps"]}
    if not math.isfinite(top["eps"]):
        return out
    inv_tau = 1.0 / float(torch.tensor(temperature, dtype=torch.float32))
    slack = 1.0 + (int(dot_dim) + 8) * U
    dcl = float(top["d_cl16"])
    c_top = (inv_tau * slack * slack + dcl) * (1.0 + 2.0 ** -20)           # no computed |cl16| exceeds this
    g = [c_top * j / grid for j in range(grid + 1)]
    e = [float(first_pass_bound(w1, b1, w2, b2, temperature, dot_dim, p_q, p_x, cl_max=x + dcl, **kw)["eps"]) for x in g]
    if not all(math.isfinite(v) for v in e):
        return out
    xs = torch.tensor(g[:-1], dtype=torch.float64)
    ys = torch.tensor(e[1:], dtype=torch.float64)
    cols = torch.stack([xs * xs, xs, torch.ones_like(xs)], 1)
    best = None
    for mask in range(1, 8):
        use = [i for i in range(3) if mask >> i & 1]
        sol = torch.linalg.lstsq(cols[:, use], ys.unsqueeze(1)).solution.squeeze(1)
        if bool((sol < 0).any()):
            continue
        coef = torch.zeros(3, dtype=torch.float64)
        coef[use] = sol
        short = float((ys - cols @ coef).clamp_min(0).max())
        coef[2] += short
        cost = float((cols @ coef - ys).mean())            # mean slack of the envelope over the grid
        if best is None or cost < best[0]:
            best = (cost, coef)
    if best is None:
        return out
    coef = best[1].clone()
    coef *= 1.0 + 2.0 ** -18
    coef[2] += 2.0 ** -17
    c32 = coef.to(torch.float32)
    up = torch.nextafter(c32, torch.full_like(c32, float("inf")))
    c32 = torch.where(c32.to(torch.float64) < coef, up, c32)
    assert bool((c32.to(torch.float64) >= coef).all()) and bool((c32 >= 0).all())
    poly = tuple(float(v) for v in c32)
    out.update({"poly": poly, "c_top": c_top, "d_cl16": dcl, "max_slack": float((cols @ best[1] - ys).max()), "mean_slack": best[0],
                "grid": grid, "eps_of_c": {f"{g[j]:.3f}": e[j] for j in range(0, grid + 1, max(1, grid // 8))}})
    return out
