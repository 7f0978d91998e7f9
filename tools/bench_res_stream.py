#!/usr/bin/env python3
"""Developer tool (GPU box): conv3 + residual + ReLU launches of layer 3 (K 256 -> N 1024, 1024 images) and layer 2's last block on
conv1x1_res_stream_kernel against the 128 x 128-tile kernel: per-launch time and algorithmic TB/s."""
import sys
import torch
sys.path.insert(0, ".")
from dbmm_amd import ops, _lib  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV); g.manual_seed(0)
for (B, H, K, N) in [(1024, 14, 256, 1024), (2048, 14, 256, 1024), (512, 14, 256, 1024), (256, 28, 128, 512)]:
    x = torch.relu(torch.randn((B, H, H, K), device=DEV, generator=g)); res = torch.relu(torch.randn((B, H, H, N), device=DEV, generator=g))
    w = (torch.randn((N, K, 1, 1), device=DEV, generator=g) * K ** -0.5).half().float()
    sc = 0.5 + torch.rand((N,), device=DEV, generator=g); b = torch.randn((N,), device=DEV, generator=g) * 0.1
    wp, wl = ops.pack_conv_weight(w, chunk_major=32)
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    xam = x.abs().max().reshape(1)
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=xam, out_scale=sc)
    M = B * H * H
    nbytes = 4 * (M * K + 2 * M * N)
    for rep in range(2):
        for mode in (0, 1):
            ops.set_option("conv1x1_res_stream", mode)
            am = torch.zeros(1, device=DEV)
            for _ in range(3):
                ops.conv_bn_act(x, wp, b, res, 1, 1, 1, 0, ops.ACT_RELU, wl, y_absmax=am, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv_bn_act(x, wp, b, res, 1, 1, 1, 0, ops.ACT_RELU, wl, y_absmax=am, **kw)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f"B={B} {H}x{H} K={K} N={N} res_stream={mode} {ops._last_igemm_tag():60s} {ms:.4f} ms  {nbytes / ms / 1e9:.2f} TB/s", flush=True)
