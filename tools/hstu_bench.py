#!/usr/bin/env python3
"""Latency of the HSTU query encoder (rails_amd.HSTU.encode, eval path) at the three rails-final geometries, B = 32, with the
CPU oracle (torch-CPU restatement of the reference) timed beside it on a bounded number of repeats."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rails_amd.hstu import HSTU
from oracle import hstu_oracle as HO

GEOM = {   # configs/*/hstu-mol-...-rails-final.gin: blocks, heads, dqk = dv, D, max_sequence_length (+ 1 output position... eval feeds +11)
    "ml-1m": dict(D=50, blocks=8, heads=2, dh=25, N=211, items=3883),
    "ml-20m": dict(D=256, blocks=16, heads=8, dh=32, N=211, items=27278),
    "amzn-books": dict(D=64, blocks=16, heads=8, dh=8, N=61, items=695762),
}
dev = torch.device("cuda", 0)
B = 32
rows = []
for name, gm in GEOM.items():
    torch.manual_seed(0)
    m = HSTU(gm["N"] - 1, 1, gm["D"], gm["blocks"], gm["heads"], gm["dh"], gm["dh"], gm["items"]).eval()
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = HO.HSTUConfig(max_sequence_len=gm["N"], embedding_dim=gm["D"], num_blocks=gm["blocks"], num_heads=gm["heads"], attention_dim=gm["dh"],
                        linear_dim=gm["dh"], num_items=gm["items"])
    g = torch.Generator().manual_seed(1)
    N = gm["N"]
    lengths = torch.randint(N // 2, N + 1, (B,), generator=g)
    ids = torch.randint(1, gm["items"] + 1, (B, N), generator=g) * (torch.arange(N).unsqueeze(0) < lengths.unsqueeze(1))
    ts = 1_000_000_000 + torch.cumsum((10.0 ** (torch.rand((B, N), generator=g) * 6)).long(), 1)
    t0 = time.perf_counter(); ref = HO.encode(cfg, w, lengths, ids, ts); cpu_ms = (time.perf_counter() - t0) * 1e3
    m = m.to(dev)
    l_d, i_d, t_d = lengths.to(dev), ids.to(dev), ts.to(dev)
    with torch.inference_mode():
        emb = m.get_item_embeddings(i_d)
        for _ in range(3):
            out = m.encode(l_d, i_d, emb, {"timestamps": t_d})
        torch.cuda.synchronize()
        rounds = []
        for _ in range(5):                      # median of 5 rounds of 20 calls: one disturbed round must not be the record
            t0 = time.perf_counter()
            for _ in range(20):
                out = m.encode(l_d, i_d, emb, {"timestamps": t_d})
            torch.cuda.synchronize()
            rounds.append((time.perf_counter() - t0) / 20 * 1e3)
        ms = sorted(rounds)[2]
    rows.append({"geometry": f"{name}: D={gm['D']}, {gm['blocks']} blocks, {gm['heads']} heads x {gm['dh']}, N={N}, B={B}", "encode_ms": ms, "encode_ms_rounds": rounds,
                 "sequences_per_s": B / ms * 1e3, "cpu_oracle_ms": cpu_ms, "max_abs_diff_vs_oracle": float((out.cpu() - ref).abs().max())})
print(json.dumps({"rows": rows}, indent=1))
