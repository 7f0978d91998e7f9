#!/usr/bin/env python3
"""Developer tool: fp16 conv3 + residual -> next conv1 as one launch (chain_f16_kernel) against the two launches, layer-1 / 2 shapes at a batch.
    python tools/bench_chain_f16.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbmm_amd  # noqa
from dbmm_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]


for name, H, K, N, P in (("layer 1", 56, 64, 256, 64), ("layer 2", 28, 128, 512, 128)):
    M = B * H * H
    y2 = torch.relu(torch.randn((M, K), device="cuda")).half(); res = torch.relu(torch.randn((M, N), device="cuda")).half()
    w3 = (torch.randn((N, K), device="cuda") * K ** -0.5).half(); w1 = (torch.randn((P, N), device="cuda") * N ** -0.5).half()
    s3, b3, s1, b1 = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda"), torch.ones(P, device="cuda"), torch.zeros(P, device="cuda")
    ms_c = t(lambda: ops.chain_f16(y2, (w3, s3, b3), res, (w1, s1, b1)))
    ms_a = t(lambda: ops.conv1x1_f16(y2, w3, s3, b3, residual=res))
    x = ops.conv1x1_f16(y2, w3, s3, b3, residual=res)
    ms_b = t(lambda: ops.conv1x1_f16(x, w1, s1, b1))
    by_c = 2 * M * (K + 2 * N + P)
    print(f"{name}: M={M} K={K} N={N} P={P}   chain {ms_c * 1e3:7.1f} us ({by_c / ms_c / 1e9:5.2f} TB/s)   conv3 + residual {ms_a * 1e3:7.1f} us + conv1' {ms_b * 1e3:7.1f} us = {(ms_a + ms_b) * 1e3:7.1f} us")
