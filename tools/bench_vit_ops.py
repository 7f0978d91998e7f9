#!/usr/bin/env python3
"""Developer tool: LayerNorm and attention-core kernels at ViT-B/32 / ViT-L/14 / text shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbmm_amd
from dbmm_amd import ops

def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]

for rows, E in [(25600, 768), (25600, 512), (147712, 1024)]:
    x = torch.randn(rows, E, device="cuda"); g = torch.ones(E, device="cuda"); b = torch.zeros(E, device="cuda")
    am = torch.zeros(1, device="cuda")
    ms = t(lambda: ops.layernorm(x, g, b)); ms2 = t(lambda: ops.layernorm(x, g, b, y_absmax=am))
    print(f"layernorm {rows}x{E}: {ms * 1e3:7.1f} us ({rows * E * 8 / ms / 1e6:6.0f} GB/s)   with absmax {ms2 * 1e3:7.1f} us")
for B, L, heads, causal in [(512, 50, 12, False), (64, 577, 16, False), (512, 77, 8, True)]:
    E = heads * 64
    qkv = torch.randn(B * L, 3 * E, device="cuda")
    ms = t(lambda: ops.mha_core(qkv, B, L, E, heads, causal))
    fl = 4.0 * L * L * 64 * heads * B * (0.5 if causal else 1.0)
    print(f"mha_core B{B} L{L} heads{heads} causal={causal}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s (useful)")
