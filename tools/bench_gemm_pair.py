#!/usr/bin/env python3
"""Developer tool: the parity-mode (fp32-accurate, fp16-pair) GEMM shapes of ViT-L/14@336px (128 images) and ViT-B/32
(1024 images), two-barrier kernel vs the deep-pipelined one (DBMM_GEMM_8PH=0 / 1 is read per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbmm_amd  # noqa: F401
from dbmm_amd import ops


def t(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]


shapes = [(73856, 3072, 1024, False, 0), (73856, 1024, 1024, True, 0), (73856, 4096, 1024, False, 2), (73856, 1024, 4096, True, 0),
          (51200, 2304, 768, False, 0), (51200, 768, 768, True, 0), (51200, 3072, 768, False, 2), (51200, 768, 3072, True, 0)]
for M, N, K, res, act in shapes:
    a = torch.randn((M, K), device="cuda"); w = (torch.randn((N, K), device="cuda") * K ** -0.5).half().float()
    b = torch.randn((N,), device="cuda"); r = torch.randn((M, N), device="cuda") if res else None
    ph, we, n = ops.split_planes_f16(w, allow_single=True)
    aam = a.abs().max().reshape(1)
    row = f"M={M:7d} N={N:5d} K={K:5d} res={int(res)} act={act}:"
    for v in ("0", "1"):
        ops.set_option("gemm_8ph", 2 * int(v))
        am = torch.zeros(1, device="cuda")
        ms = t(lambda: ops.gemm(a, w, b, r, act=act, w_planes_f16=ph, w_exp=we, a_absmax=aam, c_absmax=am))
        row += f"   8ph={v} {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF-eq"
    print(row)
