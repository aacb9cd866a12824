"""Round 5 GPU probes: (1) the datapath of v_mfma_f32_32x32x16_f16 (how many bits below the largest addend survive, truncation vs rounding),
(2) rails_topk at the proved mode's candidate counts for 1 / 8 / 32 rows, (3) census of the candidates a proved call needs."""
import sys, os, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rails_amd import engine as E
dev = torch.device("cuda:0")
U = 2.0 ** -24

def probe(a, b, c):
    return E.mfma_probe_f16(a.to(dev), b.to(dev), c.to(dev)).cpu().double()

print("== (1) MFMA f16 datapath ==")
# one big addend (C = 1 + m 2^-23) + 16 equal tiny products 2^-s: which s still contribute?
for big_is_c in (True, False):
    for s in range(20, 34):
        n = 8
        a = torch.zeros(n, 32, 16); b = torch.zeros(n, 16, 32); c = torch.zeros(n, 32, 32)
        # products: a = 2^-(s//2), b = 2^-(s - s//2)  (normal f16 down to 2^-14 each)
        a[:] = 2.0 ** -(s // 2); b[:] = 2.0 ** -(s - s // 2)
        if big_is_c:
            c[:] = 1.0
        else:
            a[:, :, 0] = 1.0; b[:, 0, :] = 1.0      # product 0 is the big one (1.0); the other 15 are tiny
        d = probe(a.half(), b.half(), c)
        exact = c.double() + a.half().double() @ b.half().double()
        print(f"big={'C' if big_is_c else 'p0'} tiny=2^-{s}: D-1 = {float(d[0,0,0]-1):.3e} exact-1 = {float(exact[0,0,0]-1):.3e}  err/ulp(1) = {float((d[0,0,0]-exact[0,0,0])/2**-23):+.3f}")
# adversarial search: random sign / exponent patterns, maximise |e| / (u mag)
g = torch.Generator().manual_seed(1)
worst = (0.0, None)
for trial in range(60):
    n = 64
    spread = int(torch.randint(1, 14, (1,), generator=g))
    ea = torch.randint(-spread, 1, (n, 32, 16), generator=g).float(); eb = torch.randint(-spread, 1, (n, 16, 32), generator=g).float()
    a = (1 + torch.rand(n, 32, 16, generator=g)) * 2 ** ea; b = (1 + torch.rand(n, 16, 32, generator=g)) * 2 ** eb
    if trial % 3 == 0:
        b = b * torch.sign(torch.randn(n, 16, 32, generator=g))
    cs = [0.0, 1.0, 2.0 ** -spread, 2.0 ** spread][trial % 4]
    c = cs * (1 + torch.rand(n, 32, 32, generator=g)) * (torch.sign(torch.randn(n, 32, 32, generator=g)) if trial % 2 else 1.0)
    d = probe(a.half(), b.half(), c.float())
    a64, b64, c64 = a.half().double(), b.half().double(), c.float().double()
    exact = c64 + a64 @ b64; mag = c64.abs() + a64.abs() @ b64.abs()
    r = float(((d - exact).abs() / (U * mag)).max())
    big = torch.maximum(c64.abs(), (a64.abs().unsqueeze(3) * b64.abs().unsqueeze(1)).amax(2))
    r2 = float(((d - exact).abs() / (U * big)).max())
    if r > worst[0]: worst = (r, (trial, spread, cs, r2))
print("adversarial random search: worst |e|/(u mag) =", worst)
# all-positive equal-magnitude terms near 1.5 (max carries)
a = torch.full((8, 32, 16), 1.4990234375); b = torch.full((8, 16, 32), 1.4990234375); c = torch.full((8, 32, 32), 1.5 * 2 ** 4)
d = probe(a.half(), b.half(), c); exact = c.double() + a.half().double() @ b.half().double()
print("equal terms: err/u|D| =", float(((d - exact).abs() / (U * exact.abs())).max()))

print("== (2) topk timings ==")
gx = torch.Generator(device=dev).manual_seed(0)
for rows in (1, 8, 32):
    x = torch.randn((rows, 695762), device=dev, generator=gx) * 3
    for k in (288, 512, 544, 768, 1024, 1536, 2048):
        ws = torch.empty(E._lib.load().rails_topk_workspace_bytes(rows, 695762, k), dtype=torch.uint8, device=dev)
        for _ in range(3): E.topk(x, k, workspace=ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): E.topk(x, k, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        print(f"rows={rows:3d} k={k:5d}: {e0.elapsed_time(e1) / 10 * 1e3:9.1f} us")

print("== (3) candidate census ==")
import rails_amd
from oracle import mol_oracle as O
def build(cfg, w, precision):
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups, cfg.item_dot_product_groups,
        cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim, cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim,
        cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False, query_nonlinearity=cfg.query_nonlinearity,
        uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(w, strict=True); mol = mol.to(dev).eval(); mol.precision = precision
    return mol
out = {}
for name, N in (("amzn-books", 695762), ("ml-20m", 27278), ("ml-1m", 3883), ("synthetic-16x16x64", 2_000_000)):
    cfg = O.CONFIGS[name]; w = O.synthetic_weights(cfg, seed=0)
    B = 128 if N < 1_000_000 else 32
    X = E.hash_item_table(1, 0, N, cfg.item_embedding_dim, dev).unsqueeze(0)
    ids = torch.arange(1, N + 1, device=dev).unsqueeze(0)
    q = O.synthetic_queries(cfg, B, seed=2).to(dev)
    kw = {"user_ids": torch.arange(1, B + 1, device=dev)} if cfg.uid_embedding_hash_sizes else {}
    with torch.inference_mode():
        tk16 = rails_amd.MoLBruteForceTopK(build(cfg, w, "f16x3"), X, ids, exact_mode="dense")
        s16 = tk16.all_logits(q, **kw)
        tk32 = rails_amd.MoLBruteForceTopK(build(cfg, w, None), X, ids, exact_mode="dense")
        s32 = tk32.all_logits(q, **kw)
    err = float((s16 - s32).abs().max())
    rec = {"N": N, "B": B, "max_err": err}
    for kp in (200, 2561):
        if kp >= N: continue
        ek = torch.topk(s32, kp, dim=1).values[:, -1:]
        for eps in (0.4, 0.6, 0.8, 1.0, 1.25, 1.5, 2.0, 2.5, 3.0, 4.0, 6.0):
            cnt = (s16 >= ek - eps).sum(1)
            rec[f"k{kp}_eps{eps}"] = (float(cnt.float().mean()), int(cnt.max()))
    out[name] = rec
    print(name, json.dumps(rec))
    del s16, s32, tk16, tk32, X
    torch.cuda.empty_cache()
json.dump(out, open("gpurun_out/s2_census.json", "w"), indent=1)
