#!/usr/bin/env python3
"""Randomised check beyond the test suite: the fused scans (rails_mol_coarse_topk, rails_mol_component_topk: sample -> threshold -> select scan with
workgroup-aggregated appends -> key selection) against the materialising path (scores of every item + exact top-k) over random corpus sizes,
batch sizes, K' / k_g, averaged or summed queries and tables with heavy ties.  Wherever every row's candidate count lies inside
[K', capacity] (the out-of-range word is 0) scores and positions must be bit-equal; where it does not, the word must say so.
  python tools/fuzz_fused_scans.py [--cases 60] [--seed 0]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from rails_amd import engine as E
from oracle import mol_oracle as O   # configurations and input generators only
from tests.test_gpu_parity import build_module

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(a.seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = flagged = 0
t0 = time.time()
with torch.inference_mode():
    mods = {}
    for case in range(a.cases):
        cfg_name = ("amzn-books", "ml-20m", "ml-1m")[ri(0, 2)]
        cfg = O.CONFIGS[cfg_name]
        if cfg_name not in mods:
            mods[cfg_name] = build_module(cfg, O.synthetic_weights(cfg, seed=3), dev)
        mol = mods[cfg_name]
        n = ri(262_144, 1_200_000)
        ties = ri(0, 3) == 0
        X = torch.from_numpy(O.hash_item_table(100 + case, 0, n, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
        if ties:
            X = X[:, torch.arange(n, device=dev) % ri(500, 5000)].contiguous()       # few distinct items: long runs of equal scores
        ids = torch.arange(1, n + 1, dtype=torch.int64, device=dev).unsqueeze(0)
        B = (1, 3, 8, 32, 33, 64, 128)[ri(0, 6)]
        q = O.synthetic_queries(cfg, B, seed=1000 + case).to(dev)
        kw = {"user_ids": torch.arange(B, dtype=torch.int64, device=dev) * 5 + 2} if cfg.uid_embedding_hash_sizes else {}
        kp = (50, 200, 500, 1000, 2000, 4000)[ri(0, 5)]
        at = rails_amd.MoLAvgTopK(mol, X, ids, avg_top_k=kp)
        eng = at._bind()
        _, eq, _ = eng.query_pack(q, kw.get("user_ids"), want_plain=True)
        average = bool(ri(0, 1))
        table = at._table()
        fused = eng.coarse_topk(eq, table, average, kp, with_flag=True)
        tag = f"case {case}: {cfg_name} n={n} B={B} K'={kp} avg={average} ties={ties}"
        if fused is None:
            print(tag, "-> unsupported sizes")
        else:
            fs, fp, counts, flag = fused
            cap = eng.coarse_topk_capacity(kp, n, B)
            in_range = bool(((counts >= kp) & (counts <= cap)).all())
            if int(flag.item()) != (0 if in_range else 1):
                bad += 1
                print(tag, "-> WORD", int(flag.item()), "counts", int(counts.min()), int(counts.max()), "cap", cap)
            if in_range:
                ok = True
                for b0 in range(0, B, 32):
                    rs, rp = E.topk(eng.coarse_scores(eq[b0:b0 + 32], table, average), kp)
                    ok = ok and torch.equal(fs[b0:b0 + 32], rs) and torch.equal(fp[b0:b0 + 32], rp)
                if not ok:
                    bad += 1
                    print(tag, "-> MISMATCH")
            else:
                flagged += 1
        # component scans (B * P_Q <= 256 query rows per call)
        kg = (1, 5, 10, 25, 50, 100)[ri(0, 5)]
        Bc = min(B, 256 // cfg.query_dot_product_groups, 128 // cfg.query_dot_product_groups if cfg.dot_product_dimension >= 128 else 999)
        nv = rails_amd.MoLNaiveTopK(mol, X, ids, k_per_group=kg)
        ctab = nv._component_table()
        cflag = torch.zeros(1, dtype=torch.int32, device=dev)
        eqc = eq[:Bc].contiguous()
        cf = eng.component_topk(eqc, ctab, kg, cflag)
        tag = f"case {case}: {cfg_name} n={n} B={Bc} k_g={kg} ties={ties} (component)"
        if cf is None:
            print(tag, "-> unsupported sizes")
        elif int(cflag.item()) == 0:
            sc_c, pos_c, _ = cf
            rows_per_q = cfg.query_dot_product_groups * cfg.item_dot_product_groups
            ok = True
            step = max(1, (1 << 28) // (rows_per_q * n))          # ~1 GB of materialised scores at a time
            for b0 in range(0, Bc, step):
                ms = eng.component_scores(eqc[b0:b0 + step], ctab)
                rs, rp = E.topk(ms, kg)
                r0, r1 = b0 * rows_per_q, min(Bc, b0 + step) * rows_per_q
                ok = ok and torch.equal(sc_c[r0:r1], rs) and torch.equal(pos_c[r0:r1], rp)
                del ms
            if not ok:
                bad += 1
                print(tag, "-> MISMATCH")
        else:
            flagged += 1
        del at, nv, X, table, ctab
        torch.cuda.empty_cache()
print(f"{a.cases} cases, {bad} failures, {flagged} calls out of range (flagged as such), {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
