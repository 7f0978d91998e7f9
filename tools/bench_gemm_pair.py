#!/usr/bin/env python3
"""Developer tool: parity-mode transformer GEMM shapes, 128 x 256 two-barrier kernel (gemm_8ph = 0) against the eight-phase kernel (2).
    python tools/bench_gemm_pair.py [--images 512] [--arch b32|l14]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dbmm_amd  # noqa: E402,F401
from dbmm_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=512)
ap.add_argument("--arch", default="b32")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
L, W = {"b32": (50, 768), "l14": (577, 1024), "b16": (197, 768), "text": (77, 512)}[a.arch]
M = a.images * L
dev = "cuda"
for name, N, K, res, act in (("qkv", 3 * W, W, False, 0), ("out_proj", W, W, True, 0), ("c_fc", 4 * W, W, False, 2), ("c_proj", W, 4 * W, True, 0)):
    x = torch.randn(M, K, device=dev); w = (torch.randn(N, K, device=dev) * K ** -0.5).half().float(); b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    ph, we, _ = ops.split_planes_f16(w, allow_single=True)
    am = x.abs().max().reshape(1)
    line = f"{name:9s} M={M:7d} N={N:5d} K={K:5d}"
    for opt in (0, 2):
        ops.set_option("gemm_8ph", opt)
        for _ in range(2):
            ops.gemm(x, w, b, r, act=act, w_planes_f16=ph, w_exp=we, a_absmax=am)
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm(x, w, b, r, act=act, w_planes_f16=ph, w_exp=we, a_absmax=am); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        line += f"   gemm_8ph={opt}: {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:6.1f} TF"
    print(line)
