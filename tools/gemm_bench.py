#!/usr/bin/env python3
"""rails_gemm_f32 alone at the HSTU layers' sizes, both weight layouts:  python tools/gemm_bench.py [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rails_amd import _lib  # noqa: E402
from rails_amd.engine import _ptr, _stream  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
cases = [tuple(int(v) for v in sys.argv[1:4])] if len(sys.argv) >= 4 else [(6752, 1024, 256), (6752, 256, 256), (6752, 512, 128)]
for M, N, K in cases:
    x = torch.randn((M, K), device=dev)
    out = torch.empty((M, N), device=dev)
    for nk in (1, 0):
        w = torch.randn((N, K) if nk else (K, N), device=dev)
        for _ in range(5):
            lib.rails_gemm_f32(_ptr(x), K, _ptr(w), nk, None, None, 0, M, N, K, 1, None, 0, _ptr(out), N, _stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.rails_gemm_f32(_ptr(x), K, _ptr(w), nk, None, None, 0, M, N, K, 1, None, 0, _ptr(out), N, _stream())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"M={M} N={N} K={K} weight {'(N,K)' if nk else '(K,N)'}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s  ({2.0 * M * N * K / us / 1e6 / 157.3:.2f} of the fp32 MFMA peak)")
