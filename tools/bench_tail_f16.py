#!/usr/bin/env python3
"""Developer tool: fp16 GEMM shapes whose 256 x 256 tiles leave a short last round: one launch (tail_split = 0), K cut (2), row split (3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbmm_amd  # noqa: F401
from dbmm_amd import ops


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]


shapes = [(25600, 768, 768, True, 0), (25600, 768, 3072, True, 0), (51200, 768, 768, True, 0), (51200, 768, 3072, True, 0),
          (200704, 256, 1024, False, 1), (200704, 1024, 256, True, 1), (200704, 512, 1024, False, 1), (50176, 2048, 512, True, 1),
          (147712, 1024, 1024, True, 0), (147712, 1024, 4096, True, 0), (78848, 512, 2048, True, 0)]
for M, N, K, res, act in shapes:
    a = torch.randn((M, K), device="cuda").half(); w = (torch.randn((N, K), device="cuda") * K ** -0.5).half()
    b = torch.randn((N,), device="cuda"); r = torch.randn((M, N), device="cuda").half() if res else None
    tiles = ((M + 255) // 256) * (N // 256)
    row = f"M={M:7d} N={N:5d} K={K:5d} res={int(res)} tiles={tiles:5d} (rem {tiles % 256:3d}):"
    for v in (0, 2, 3):
        ops.set_option("tail_split", v)
        ms = t(lambda: ops.gemm_f16(a, w, b, residual=r, act=act))
        row += f"   split={v} {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:6.1f} TF"
    print(row)
