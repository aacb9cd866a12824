"""rails_candidates_select / rails_candidates_finish (round 6: the fused tail of the proved exact top-k) against a torch restatement of
their contract (include/rails_amd.h): the candidate set is {x : bin(s_x) >= b_t} for the lowest threshold bin that leaves at most `cap`
candidates, the finish returns the candidates' top-k by (exact score desc, position asc), the verdict state of rails_rescore_verdict's
layout, and leaves the workspace zeroed.  Reference call sites replaced: rails/indexing/mol_top_k.py:99-130 (torch.topk over all logits)
and indexing/candidate_index.py:149-175 (the seen-id filter, fused into the finish)."""
import numpy as np
import pytest
import torch

from rails_amd import engine as E

pytestmark = pytest.mark.gpu
BINS = 4096


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _bins(s: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    """the kernels' bin function in the same fp32 operations"""
    lo32, hi32 = torch.tensor(lo, dtype=torch.float32), torch.tensor(hi, dtype=torch.float32)
    scale = torch.tensor(float(BINS), dtype=torch.float32) / (hi32 - lo32)
    x = (s.float().cpu() - lo32) * scale
    x = torch.nan_to_num(x, nan=0.0, posinf=float(BINS - 1), neginf=0.0).clamp(0.0, float(BINS - 1))
    return x.to(torch.int64)


def _expected(s: torch.Tensor, lo: float, hi: float, cap: int):
    """per row: (threshold bin, sorted positions of the expected candidates)"""
    out = []
    b = _bins(s, lo, hi)
    for r in range(s.shape[0]):
        h = torch.bincount(b[r], minlength=BINS)
        from_top = torch.flip(torch.cumsum(torch.flip(h, [0]), 0), [0])       # from_top[i] = #{bin >= i}
        fits = (from_top <= cap).nonzero()
        bt = int(fits[0]) if fits.numel() else BINS
        out.append((bt, (b[r] >= bt).nonzero().reshape(-1)))
    return out


def _ws_clean(ws, rows) -> bool:
    """what a select + finish pair must leave zeroed (include/rails_amd.h): counts, flags, the arrival counter, both histograms -- the rows'
    verdict words behind the counter are overwritten by every call and may hold the last one's"""
    rp = (rows + 3) // 4 * 4
    return int(ws[: 2 * rp + 1].abs().sum()) == 0 and int(ws[2 * rp + 8 + 4 * rows :].abs().sum()) == 0


def _select(s, cap, lo, hi):
    B = s.shape[0]
    ws = E.candidates_workspace(B, s.device)
    pos = torch.zeros((B, cap), dtype=torch.int64, device=s.device)
    a = torch.zeros((B, cap), dtype=torch.float32, device=s.device)
    E.candidates_select(s, cap, lo, hi, ws, pos, a)
    torch.cuda.synchronize()
    return ws, pos, a


@pytest.mark.parametrize("B,N,cap", [(4, 1000, 64), (32, 27278, 512), (5, 65536, 1024), (8, 70001, 1024), (3, 200003, 2048), (32, 695762, 1024), (2, 695762, 10272),
                                      (7, 300, 512)])
def test_threshold_selection_matches_the_contract(dev, B, N, cap):
    g = torch.Generator().manual_seed(B * 1000 + cap)
    s = (torch.randn(B, N, generator=g) * 3.0).to(dev)
    s[0, : min(N, 40)] = 2.5                       # a run of ties
    lo, hi = -20.4, 20.4
    ws, pos, a = _select(s, cap, lo, hi)
    counts = ws[:B].cpu()
    exp = _expected(s, lo, hi, cap)
    sc = s.cpu()
    for r in range(B):
        bt, want = exp[r]
        c = int(counts[r])
        assert c == want.numel() <= cap, (r, c, want.numel(), bt)
        got = pos[r, :c].cpu()
        order = torch.argsort(got)
        assert torch.equal(got[order], want)
        assert torch.equal(a[r, :c].cpu()[order], sc[r, want])
        if N <= cap:
            assert bt == 0 and c == N
    # the finish with exact = the same scores: the candidates' top-k equals torch's (ties by position), the verdict clears, the workspace is zero again
    k = min(50, int(counts.min()))
    exact = torch.zeros((B, cap), dtype=torch.float32, device=dev)
    for r in range(B):
        c = int(counts[r])
        exact[r, :c] = s[r, pos[r, :c]]
    state = torch.zeros(8, dtype=torch.float32, device=dev)
    host = torch.zeros(8, dtype=torch.float32).pin_memory()
    ids = torch.arange(N, dtype=torch.int64, device=dev) * 3 + 1
    out_s, out_i, _, _ = E.candidates_finish(exact, a, pos, cap, ws, ids, N, k, 0.0, 1.0, False, None, 0, 0.0, state, host)
    torch.cuda.synchronize()
    assert _ws_clean(ws, B), "the workspace is not left zeroed"
    for r in range(B):
        key = torch.stack([-sc[r].double(), torch.arange(N, dtype=torch.float64)], 1).numpy()
        ref = torch.from_numpy(np.lexsort((key[:, 1], key[:, 0]))[:k].copy())
        assert torch.equal(out_i[r].cpu(), ref * 3 + 1), r
        assert torch.equal(out_s[r].cpu(), sc[r, ref])
    st = state.cpu()
    assert torch.equal(st, host), (st, host)
    whole = all(int(counts[r]) == N for r in range(B))
    gap = min(float(sc[r, exp[r][1]].topk(k).values[-1] - (sc[r, exp[r][1]].min() if int(counts[r]) < N else float("-inf"))) for r in range(B))
    assert st.view(torch.int32)[1] == (0 if (gap > 0 or whole) else 1)
    assert float(st[3]) == 0.0 and float(st[5]) == 1.0 and float(st[6]) == float(st.view(torch.int32)[1])
    assert float(st[4]) == pytest.approx(gap) or (whole and float(st[4]) == float("inf"))


def test_nan_crowding_and_guard_fail_the_verdict(dev):
    B, N, cap, k = 4, 100_000, 512, 20
    g = torch.Generator().manual_seed(5)
    base = (torch.randn(B, N, generator=g) * 2.0).to(dev)
    lo, hi = -20.4, 20.4
    ids = torch.arange(N, dtype=torch.int64, device=dev)

    def run(s, eps=0.0, guard=None, limit=0.0, exact_delta=None):
        ws, pos, a = _select(s, cap, lo, hi)
        counts = ws[:B].cpu()
        exact = torch.zeros((B, cap), dtype=torch.float32, device=dev)
        for r in range(B):
            c = int(counts[r])
            exact[r, :c] = s[r, pos[r, :c]]
        if exact_delta is not None:
            exact += exact_delta
        state = torch.zeros(8, dtype=torch.float32, device=dev)
        E.candidates_finish(exact, a, pos, cap, ws, ids, N, k, eps, 1.0, False, guard, 0 if guard is None else guard.shape[1], limit, state, None)
        torch.cuda.synchronize()
        assert _ws_clean(ws, B)
        return state.cpu(), counts

    st, counts = run(base)
    assert st.view(torch.int32)[1] == 0 and counts.min() >= k
    # eps larger than the margin between the k-th candidate and the threshold: every row fails
    st, _ = run(base, eps=50.0)
    assert st.view(torch.int32)[1] == 1 and float(st[6]) == 1.0
    # a NaN anywhere in a row
    s = base.clone(); s[2, 77_777] = float("nan")
    st, _ = run(s)
    assert st.view(torch.int32)[1] == 1 and float(st[3]) == float("inf")
    # crowded: more than cap scores in the top bin -> no candidates -> redo
    s = base.clone(); s[1, :2000] = 15.0
    st, counts = run(s)
    assert counts[1] == 0 and st.view(torch.int32)[1] == 1
    # a guard value beyond its limit, and a NaN guard
    guard = torch.full((B, 64), 0.5, device=dev)
    st, _ = run(base, guard=guard, limit=1.0)
    assert st.view(torch.int32)[1] == 0 and float(st[7]) == 0.5
    guard[3, 5] = -2.0
    st, _ = run(base, guard=guard, limit=1.0)
    assert st.view(torch.int32)[1] == 1 and float(st[7]) == 2.0
    # an observed |exact - approx| is recorded (state[0], state[3]) and widens eps (safety 1)
    st, _ = run(base, exact_delta=0.25)
    assert float(st[0]) == pytest.approx(0.25, abs=1e-6) and float(st[2]) == pytest.approx(0.25, abs=1e-6)


def test_finish_filter_and_message_forms(dev):
    """the seen-id filter inside the finish == rails_filter_seen_ids over its top-k; the item-sharded message form carries top-k | ids | m | err"""
    B, N, cap, kp, k = 6, 50_000, 512, 200, 120
    g = torch.Generator().manual_seed(9)
    s = (torch.randn(B, N, generator=g) * 2.0).to(dev)
    ids = torch.arange(N, dtype=torch.int64, device=dev) * 2 + 5
    ws, pos, a = _select(s, cap, -20.4, 20.4)
    counts = ws[:B].cpu()
    exact = torch.zeros((B, cap), dtype=torch.float32, device=dev)
    for r in range(B):
        exact[r, : int(counts[r])] = s[r, pos[r, : int(counts[r])]] + 0.125
    top = torch.topk(s, kp, dim=1).indices
    inv = ids[torch.cat([top[:, :30], torch.randint(0, N, (B, 50), generator=g).to(dev)], 1)]
    state = torch.zeros(8, dtype=torch.float32, device=dev)
    ws2 = ws.clone()
    out_s, out_i, f_i, f_s = E.candidates_finish(exact, a, pos, cap, ws, ids, N, kp, 0.0, 1.0, False, None, 0, 0.0, state, None, seen=(inv, k))
    r_i, r_s = E.filter_seen_ids(out_i, out_s, inv, k)
    assert torch.equal(f_i, r_i) and torch.equal(f_s, r_s)
    msg = torch.zeros((B, 2 * kp + 2), dtype=torch.int64, device=dev)
    E.candidates_finish(exact, a, pos, cap, ws2, ids, N, kp, 0.0, 1.0, False, None, 0, 0.0, None, None, msg=msg)
    torch.cuda.synchronize()
    assert _ws_clean(ws2, B)
    assert torch.equal(msg[:, :kp].to(torch.int32).view(torch.float32), out_s) and torch.equal(msg[:, kp : 2 * kp], out_i)
    m = msg[:, 2 * kp].to(torch.int32).view(torch.float32).cpu()
    err = msg[:, 2 * kp + 1].to(torch.int32).view(torch.float32).cpu()
    for r in range(B):
        assert float(m[r]) == float(a[r, : int(counts[r])].min()) and float(err[r]) == pytest.approx(0.125, abs=1e-6)
