"""Self-check of the arithmetic model behind the proved exact top-k, on the device the module runs on.

The a-priori bound of rails_amd/f16x3_bound.py rests on three hypotheses about the part (H1: v_mfma_f32_32x32x2_f32 is a chain of
round-to-nearest fmas; H2: v_mfma_f32_32x32x16_f16 errs by at most KC u (|C| + sum |p|) + KP u sum |p| and keeps f16 subnormals; H3: v_exp_f32 /
v_rcp_f32 are good to one ulp).  They were measured on an MI355X (tests/test_proved_gpu.py runs the full families); this module re-measures a
compact set of them -- random operands and the operand families built for the measured datapath -- once per device and process, through the same
probe entry points (rails_mfma_probe_f16 / _f32, rails_scalar_probe_f32), before MoLBruteForceTopK binds the proved mode.  A device that
violates one of them gets the dense fp32 kernels (and a warning), not a proof that does not hold there.
No counterpart in the reference (rails/indexing/mol_top_k.py:99-130 scores every item in one precision)."""
from __future__ import annotations

import warnings
from typing import Callable, Dict, Optional

import torch

from . import f16x3_bound as FB

U = FB.U
_cache: Dict[str, Dict[str, object]] = {}


def _f16_cases(g: torch.Generator):
    n = 16
    rnd = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    a, b = rnd(n, 32, 16), rnd(n, 16, 32)
    out = [("gaussian", a, b, rnd(n, 32, 32)), ("large C", 2.0 ** -5 * a, 2.0 ** -5 * b, 1024 * (1 + torch.rand(n, 32, 32, generator=g)))]
    for spread in (8, 13):
        ea = torch.randint(-spread, 1, (n, 32, 16), generator=g).float()
        eb = torch.randint(-spread, 1, (n, 16, 32), generator=g).float()
        out.append((f"mixed exponents 2^-{2 * spread}", (1 + torch.rand(n, 32, 16, generator=g)) * 2 ** ea,
                    (1 + torch.rand(n, 16, 32, generator=g)) * 2 ** eb * torch.sign(rnd(n, 16, 32)), 2.0 ** -spread * rnd(n, 32, 32)))
    # built for the datapath: per lane half one product ~ 1 and seven just under 2^-24 of it; a C ~ 1 with sixteen products just under 2^-26 / 2^-24 of it
    big = torch.full((n, 32, 16), (2 - 2.0 ** -10) * 2.0 ** -13)
    bb = torch.full((n, 16, 32), (2 - 2.0 ** -10) * 2.0 ** -13)
    big[:, :, 0] = 1.0
    big[:, :, 8] = 1.0
    bb[:, 0, :] = 1.0 + torch.rand(n, 32, generator=g).round(decimals=2)
    bb[:, 8, :] = 1.0 + torch.rand(n, 32, generator=g).round(decimals=2)
    out.append(("two big products + 14 just under 2^-24", big, bb, torch.zeros(n, 32, 32)))
    sm = torch.full((n, 32, 16), (2 - 2.0 ** -10) * 2.0 ** -14)
    sb = torch.full((n, 16, 32), (2 - 2.0 ** -10) * 2.0 ** -14)
    out.append(("C + 16 products just under 2^-26", sm, sb, 1.0 + torch.rand(n, 32, 32, generator=g)))
    out.append(("C + 16 products just under 2^-24", 2 * sm, 2 * sb, 1.0 + torch.rand(n, 32, 32, generator=g)))
    sub = torch.randint(1, 1024, (n, 32, 16), generator=g).float() * 2.0 ** -24
    out.append(("f16 subnormal operands", sub, 16 * b, torch.zeros(n, 32, 32)))
    return [(name, x.half(), y.half(), z.float()) for name, x, y, z in out]


@FB._few_cpu_threads
def measure(device: torch.device, probe_f16: Optional[Callable] = None, probe_f32: Optional[Callable] = None,
            probe_scalar: Optional[Callable] = None) -> Dict[str, object]:
    """Worst observed error / model bound per hypothesis (<= 1 means the hypothesis held on every case).  The probes default to the library's."""
    from . import engine as E

    probe_f16 = probe_f16 or E.mfma_probe_f16
    probe_f32 = probe_f32 or E.mfma_probe_f32
    probe_scalar = probe_scalar or E.scalar_probe
    g = torch.Generator().manual_seed(20260930)
    h2, subnormals_kept = 0.0, True
    for name, a, b, c in _f16_cases(g):
        d = probe_f16(a.to(device), b.to(device), c.to(device)).cpu().double()
        a64, b64, c64 = a.double(), b.double(), c.double()
        exact = c64 + a64 @ b64
        psum = a64.abs() @ b64.abs()
        bound = U * (FB.KC * (c64.abs() + psum) + FB.KP * psum)
        h2 = max(h2, float(((d - exact).abs() / bound.clamp_min(1e-300)).max()))
        if "subnormal" in name:
            subnormals_kept = bool(float(d.abs().max()) > 0 and float(((d - exact).abs() / (c64.abs() + psum).clamp_min(1e-300)).max()) < 2.0 ** -20)
    h1 = 0.0
    for scale_c in (1.0, 1e3, 0.0):
        a, b, c = torch.randn(16, 32, 2, generator=g), torch.randn(16, 2, 32, generator=g), scale_c * torch.randn(16, 32, 32, generator=g)
        d = probe_f32(a.to(device), b.to(device), c.to(device)).cpu().double()
        a64, b64, c64 = a.double(), b.double(), c.double()
        p0, p1 = a64[:, :, 0:1] * b64[:, 0:1, :], a64[:, :, 1:2] * b64[:, 1:2, :]
        bound = 2 * U * (c64.abs() + torch.maximum(p0.abs(), p1.abs())) + U * torch.minimum(p0.abs(), p1.abs())
        h1 = max(h1, float(((d - (c64 + p0 + p1)).abs() / bound.clamp_min(1e-300)).max()))
    x = torch.cat([torch.randn(1 << 12, generator=g) * s for s in (0.1, 1.0, 8.0, 40.0)] + [torch.linspace(-120, 120, 1 << 11)]).clamp(-124.0, 124.0)
    out = probe_scalar(x.to(device)).cpu().double()
    x64 = x.double()
    e_exp = float(((out[0] - torch.exp2(x64)).abs() / torch.exp2(x64)).max())
    nz = x64.abs() > 1e-30
    e_rcp = float(((out[1][nz] - 1 / x64[nz]).abs() * x64[nz].abs()).max())
    phi = x64 / (1 + torch.exp2(x64))
    sel = phi.abs() > 1e-30
    e_phi = float(((out[2][sel] - phi[sel]).abs() / phi[sel].abs()).max())
    h3 = max(e_exp / (2 * U * 1.001), e_rcp / (2 * U * 1.001), e_phi / FB.gamma(7))
    ok = bool(h1 <= 1.0 + 1e-6 and h2 <= 1.0 and subnormals_kept and h3 <= 1.0)
    return {"ok": ok, "h1_fp32_mfma_fma_chain": h1, "h2_f16_mfma_kc_kp": h2, "h2_f16_subnormals_kept": subnormals_kept, "h3_exp_rcp_phi": h3}


def device_ok(device: torch.device) -> bool:
    """True iff the arithmetic model held on this device (measured once per device and process; a failure warns and is remembered)."""
    key = str(torch.device(device))
    if key not in _cache:
        try:
            _cache[key] = measure(torch.device(device))
        except Exception as exc:   # noqa: BLE001 -- a probe that cannot run proves nothing: the caller takes the dense kernels
            _cache[key] = {"ok": False, "error": f"{type(exc).__name__}: {exc}"[:200]}
        if not _cache[key]["ok"]:
            warnings.warn(f"rails_amd: the arithmetic model of the proved exact top-k does not hold on {key} ({_cache[key]}); MoLBruteForceTopK runs the dense fp32 kernels")
    return bool(_cache[key]["ok"])


def report(device: torch.device) -> Dict[str, object]:
    device_ok(device)
    return dict(_cache[str(torch.device(device))])
