#!/usr/bin/env python3
"""Developer tool: the fp16 GEMM shapes of ViT-L/14@336px (256 images) and ViT-B/32 (1024 images), per kernel variant
(DBMM_F16_8PH=0 / 1 is read per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbmm_amd  # noqa: F401
from dbmm_amd import ops


def t(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]


shapes = [(147712, 3072, 1024, False, 0), (147712, 1024, 1024, True, 0), (147712, 4096, 1024, False, 2), (147712, 1024, 4096, True, 0),
          (51200, 2304, 768, False, 0), (51200, 768, 768, True, 0), (51200, 3072, 768, False, 2), (51200, 768, 3072, True, 0),
          (16384, 4096, 4096, False, 0)]
for M, N, K, res, act in shapes:
    a = torch.randn((M, K), device="cuda").half(); w = (torch.randn((N, K), device="cuda") * K ** -0.5).half()
    b = torch.randn((N,), device="cuda"); r = torch.randn((M, N), device="cuda").half() if res else None
    row = f"M={M:7d} N={N:5d} K={K:5d} res={int(res)} act={act}:"
    for v in ("0", "1"):
        ops.set_option("f16_8ph", int(v))
        ms = t(lambda: ops.gemm_f16(a, w, b, residual=r, act=act))
        row += f"   8ph={v} {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF"
    print(row)
