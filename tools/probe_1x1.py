#!/usr/bin/env python3
"""Developer tool: the two layer-3 1x1 conv shapes of RN50 at B = 1024 (conv1: K 1024 -> 256; conv3 + residual: K 256 -> 1024) on the
128 x 128 fp16-pair kernel, a few launches each -- a target for `rocprofv3 --pmc ...` passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbmm_amd  # noqa: F401
from dbmm_amd import ops

ops.set_option("conv1x1_8ph", 0)
B, H = 1024, 14
dev = "cuda"
for name, Cin, Cout, res in (("c1", 1024, 256, False), ("c3", 256, 1024, True)):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 1, 1, device=dev) * Cin ** -0.5).half().float()
    wp, wl = ops.pack_conv_weight(w)
    ph, we, _ = ops.split_planes_f16(wp, allow_single=True)
    r = torch.randn(B, H, H, Cout, device=dev) if res else None
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=x.abs().max().reshape(1), y_absmax=torch.zeros(1, device=dev),
              out_scale=0.5 + torch.rand(Cout, device=dev))
    b = torch.randn(Cout, device=dev)
    for _ in range(6):
        ops.conv_bn_act(x, wp, b, r, 1, 1, 1, 0, ops.ACT_RELU, wl, **kw)
    torch.cuda.synchronize()
    del x, r
print("done")
