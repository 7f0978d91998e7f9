#!/usr/bin/env python3
"""Developer tool: the conv3 + downsample dual-source GEMM of the three strided stages at a given batch, 128 x 128 kernel against the eight-phase one.
    python tools/bench_dual.py [--batch 1024] [--iters 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dbmm_amd  # noqa: E402,F401
from dbmm_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = "cuda"
for name, hw, K, K2, N in (("l2.0", 28, 128, 256, 512), ("l3.0", 14, 256, 512, 1024), ("l4.0", 7, 512, 1024, 2048)):
    M = a.batch * hw * hw
    y = torch.relu(torch.randn(M, K, device=dev)); x = torch.relu(torch.randn(M, K2, device=dev))
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half().float(); w2 = (torch.randn(N, K2, device=dev) * K2 ** -0.5).half().float()
    ph, we, _ = ops.split_planes_f16(w, allow_single=True); ph2, we2, _ = ops.split_planes_f16(w2, allow_single=True)
    sc = 0.5 + torch.rand(N, device=dev); ratio = (0.5 + torch.rand(N, device=dev)) * 2.0 ** (we - we2)
    b = torch.randn(N, device=dev)
    ya, xa = y.abs().max().reshape(1), x.abs().max().reshape(1)
    line = f"{name}  M={M:8d} N={N:5d} K={K:4d} K2={K2:4d}"
    for opt in (0, 2):
        ops.set_option("dual_8ph", opt)
        cam = torch.zeros(1, device=dev)
        for _ in range(2):
            ops.gemm_dual(y, ya, ph, we, sc, x, xa, ph2, ratio, b, ops.ACT_RELU, cam)
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm_dual(y, ya, ph, we, sc, x, xa, ph2, ratio, b, ops.ACT_RELU, cam); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        line += f"   dual_8ph={opt}: {ms:.3f} ms {2.0 * M * N * (K + K2) / ms / 1e9:7.1f} TF ({ops._last_igemm_tag()[:24]})"
    print(line)
