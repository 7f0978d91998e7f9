#!/usr/bin/env python3
"""Developer tool: where one wave of the eight-wave chain kernel spends its cycles (needs the -DCH8_STAMP build of the library:
DBMM_LIB=tools/_bin/libdbmm_stamp.so)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dbmm_amd  # noqa: E402,F401
from dbmm_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
H, K, N, P = 14, 256, 1024, 256
g = torch.Generator(device=dev); g.manual_seed(1)
rn = lambda *sh: torch.randn(sh, device=dev, generator=g)


def entry(w):
    ph, we, _ = ops.split_planes_f16(w, allow_single=True)
    return dict(w=w, ph=ph, we=we, sc=0.5 + torch.rand((w.shape[0],), device=dev, generator=g), b=rn(w.shape[0]) * 0.1)


c3, c1 = entry((rn(N, K) * K ** -0.5).half().float()), entry((rn(P, N) * N ** -0.5).half().float())
y2, res = torch.relu(rn(B, H, H, K)), torch.relu(rn(B, H, H, N) * 2.0)
ya = y2.abs().max().reshape(1)
for _ in range(3):
    ops.bottleneck_chain(y2, ya, c3, res, c1, torch.zeros(1, device=dev), torch.zeros(1, device=dev))
torch.cuda.synchronize()
out = (ctypes.c_longlong * 24)()
fn = _lib.lib().dbmm_debug_chain8_stamps
fn.argtypes = [ctypes.c_void_p]; fn.restype = None
fn(out)
names = ["tick 1 body", "barrier", "tick 2 body", "barrier", "tick 3 body", "barrier"]
for g, (label, seq) in enumerate((("wave 0 (group A)", "EP | C1 | C3'"), ("wave 4 (group B)", "C3 | EP | C1"))):
    v = list(out[8 * g:8 * g + 6])
    tot = sum(v)
    print(f"B={B}: {label}, ticks = {seq}: cycles per slab {tot / 16:.0f}")
    for n, x in zip(names, v):
        print(f"  {n:14s} {x / 16:8.0f}  {100.0 * x / max(tot, 1):5.1f} %")
print("EP of wave 0, cycles per slab:", {n: round(out[16 + i] / 16) for i, n in enumerate(["wait for residual / BN loads", "DMA issue", "reads + arithmetic + stores + slab", "ds_max", "residual loads issue"])})
