#!/usr/bin/env python3
"""Developer tool: the eight-phase fp16 GEMM on the ViT-B/32 shapes at 512 images (25,600 token rows), option tail_split = 0 / 1,
and the 128 x 128 kernel (f16_8ph = 0) beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbmm_amd  # noqa: F401
from dbmm_amd import ops


def t(fn, n=12):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]


M = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
for N, K, res, act in ((2304, 768, False, 0), (768, 768, True, 0), (3072, 768, False, 2), (768, 3072, True, 0)):
    a = torch.randn((M, K), device="cuda").half(); w = (torch.randn((N, K), device="cuda") * K ** -0.5).half()
    b = torch.randn((N,), device="cuda"); r = torch.randn((M, N), device="cuda").half() if res else None
    row = f"M={M:7d} N={N:5d} K={K:5d} res={int(res)} act={act}: tiles {((M + 255) // 256) * (N // 256):5d}"
    for name, o8, ts in (("8ph", 1, 0), ("8ph+tail", 1, 1), ("128x128", 0, 0)):
        ops.set_option("f16_8ph", o8); ops.set_option("tail_split", ts)
        ms = t(lambda: ops.gemm_f16(a, w, b, residual=r, act=act))
        row += f"   {name} {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:6.0f} TF"
    print(row)
