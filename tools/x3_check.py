#!/usr/bin/env python3
"""Developer tool: accuracy (vs an fp64 reference) and speed of the split-precision igemm path
against the fp32-MFMA path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import dbmm_amd
from dbmm_amd import ops, synth

def t(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]

def rel(a, b): return ((a.double() - b).abs().max() / b.abs().max()).item()

# accuracy: GEMM
for (M, N, K) in [(512, 256, 1024), (1000, 128, 64), (300, 64, 576)]:
    a = synth.normal(1, "a", (M, K)); w = synth.normal(2, "w", (N, K), K ** -0.5); b = synth.normal(3, "b", (N,))
    ref = a.double() @ w.double().t() + b.double()
    ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
    o32 = ops.gemm(ad, wd, bd); pl = ops.split_planes(wd); o3 = ops.gemm(ad, wd, bd, w_planes=pl)
    print(f"gemm {M}x{N}x{K}: fp32-mfma err {rel(o32.cpu(), ref):.2e}  x3 err {rel(o3.cpu(), ref):.2e}")
# accuracy: conv
for (B, H, Cin, Cout) in [(2, 16, 32, 128), (3, 14, 64, 128), (2, 9, 16, 64), (4, 7, 512, 256)]:
    x = synth.normal(1, "x", (B, Cin, H, H)); w = synth.normal(2, "w", (Cout, Cin, 3, 3), (Cin * 9) ** -0.5)
    b = synth.normal(3, "b", (Cout,), 0.1)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda(); wp, wl = ops.pack_conv_weight(w.cuda()); pl = ops.split_planes(wp)
    o32 = ops.conv_bn_act(xd, wp, b.cuda(), None, 3, 3, 1, 1, ops.ACT_RELU, wl)
    o3 = ops.conv_bn_act(xd, wp, b.cuda(), None, 3, 3, 1, 1, ops.ACT_RELU, wl, w_planes=pl)
    ph, we, _ = ops.split_planes_f16(wp); xam = xd.abs().max().reshape(1); yam = torch.zeros(1, device="cuda")
    o2 = ops.conv_bn_act(xd, wp, b.cuda(), None, 3, 3, 1, 1, ops.ACT_RELU, wl, w_planes_f16=ph, w_exp=we, x_absmax=xam, y_absmax=yam)
    print(f"conv B{B} H{H} {Cin}->{Cout}: fp32-mfma err {rel(o32.cpu(), ref):.2e}  x3 err {rel(o3.cpu(), ref):.2e}  "
          f"x2 err {rel(o2.cpu(), ref):.2e} [{ops._last_igemm_tag()}] absmax {yam.item():.6f} vs {o2.abs().max().item():.6f}")
# speed on RN50 shapes at B=512
B = 512
for name, H, Cin, Cout, k in [("l2.0.c2", 56, 128, 128, 3), ("l2.1.c2", 28, 128, 128, 3), ("l3.1.c2", 14, 256, 256, 3),
                              ("l4.1.c2", 7, 512, 512, 3), ("l3.1.c1", 14, 1024, 256, 1), ("l3.1.c3", 14, 256, 1024, 1),
                              ("l1.1.c2", 56, 64, 64, 3), ("l2.0.c1", 56, 256, 128, 1)]:
    x = torch.randn(B, H, H, Cin, device="cuda"); w, wl = ops.pack_conv_weight(torch.randn(Cout, Cin, k, k, device="cuda") * (Cin * k * k) ** -0.5)
    bb = torch.randn(Cout, device="cuda"); pl = ops.split_planes(w); pad = 1 if k == 3 else 0
    t32 = t(lambda: ops.conv_bn_act(x, w, bb, None, k, k, 1, pad, ops.ACT_RELU, wl))
    t3 = t(lambda: ops.conv_bn_act(x, w, bb, None, k, k, 1, pad, ops.ACT_RELU, wl, w_planes=pl))
    ph, we, _ = ops.split_planes_f16(w); xam = x.abs().max().reshape(1); yam = torch.zeros(1, device="cuda")
    t2 = t(lambda: ops.conv_bn_act(x, w, bb, None, k, k, 1, pad, ops.ACT_RELU, wl, w_planes_f16=ph, w_exp=we, x_absmax=xam, y_absmax=yam))
    fl = 2.0 * B * H * H * Cout * Cin * k * k
    print(f"{name}: fp32-mfma {t32:.3f} ms ({fl / t32 / 1e9:.0f} TF)   x3 {t3:.3f} ms ({fl / t3 / 1e9:.0f} TF-eq)   "
          f"x2 {t2:.3f} ms ({fl / t2 / 1e9:.0f} TF-eq)  [{ops._last_igemm_tag()}]")
