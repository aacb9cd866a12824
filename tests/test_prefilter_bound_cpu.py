"""CPU: the error bound behind the int8 pre-filter of the fused coarse top-K' (oracle.int8_prefilter_bound restates the kernels'
quantisation and their epsilon).  For every (query, item) pair |q . x - s s_q I| <= eps, hence an item whose bf16 score reaches a
threshold has I >= floor((thr_lo - eps) / (s s_q)) - 2, the integer bound the select scan starts its accumulators from."""
import numpy as np
import pytest
import torch

from oracle import mol_oracle as O


def _bf16(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("case", ["gaussian", "outlier", "half_steps", "tiny_query", "sparse", "d128"])
def test_int8_bound_holds_for_every_pair(case):
    g = torch.Generator().manual_seed(["gaussian", "outlier", "half_steps", "tiny_query", "sparse", "d128"].index(case) + 11)
    d = 128 if case == "d128" else 32
    n, B = 20_000, 16
    x = torch.randn((n, d), generator=g) / (8 * d) ** 0.5
    q = torch.randn((B, d), generator=g) * 0.5
    if case == "outlier":
        x[17] *= 1000.0
    if case == "half_steps":      # values that sit exactly between two quantisation steps: the worst case of |e_k| = 1/2
        s = float(x.abs().max()) / 127.0
        x = (torch.randint(-126, 126, (n, d), generator=g).float() + 0.5) * s
        x[0, 0] = 127.0 * s
    if case == "tiny_query":
        q = q * 1e-6
    if case == "sparse":
        x = x * (torch.rand((n, d), generator=g) < 0.1)
        q = q * (torch.rand((B, d), generator=g) < 0.2)
    xb, qb = _bf16(x), _bf16(q)
    I, eps, s, s_q = O.int8_prefilter_bound(qb, xb)
    exact = qb.double() @ xb.double().T                                   # the real dot product of the bf16 operands
    err = (exact - (s.double() * s_q.double())[:, None] * I.double()).abs()
    assert bool((err <= eps.double()[:, None]).all()), float((err / eps.double()[:, None]).max())
    # the integer test: every pair whose bf16-rounded score reaches the K'-th score has I >= bound
    score = exact.float().bfloat16().float()
    thr = torch.sort(score, dim=1, descending=True).values[:, 99]         # K' = 100
    thr_lo = torch.nextafter(thr.bfloat16(), torch.tensor(-float("inf")).bfloat16()).float() if hasattr(torch, "nextafter") else thr - thr.abs() * 2 ** -7
    bound = torch.floor((thr_lo - eps) / (s * s_q)) - 2.0
    kept = score >= thr[:, None]
    assert bool((I[kept].float() >= bound[:, None].expand_as(score)[kept]).all())
    # and the bound is not vacuous on ordinary data: few pairs pass it
    if case in ("gaussian", "d128"):
        assert float((I.float() >= bound[:, None]).float().mean()) < 0.05
