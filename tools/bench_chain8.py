#!/usr/bin/env python3
"""Developer tool: conv3 + residual -> next conv1 of the RN50 layer-3 geometry (K = P = 256, N = 1024, 14 x 14 maps) as ONE launch of the
eight-wave chain kernel (option chain8 = 1) against the two separate launches (chain8 = 0), same operands, HIP events, median."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dbmm_amd  # noqa: E402,F401
from dbmm_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=8)
    a = ap.parse_args()
    dev = "cuda"
    B, H, K, N, P = a.batch, 14, 256, 1024, 256
    g = torch.Generator(device=dev); g.manual_seed(1)
    rn = lambda *sh: torch.randn(sh, device=dev, generator=g)

    def entry(w):
        ph, we, _ = ops.split_planes_f16(w, allow_single=True)
        return dict(w=w, ph=ph, we=we, sc=0.5 + torch.rand((w.shape[0],), device=dev, generator=g), b=rn(w.shape[0]) * 0.1)
    c3, c1 = entry((rn(N, K) * K ** -0.5).half().float()), entry((rn(P, N) * N ** -0.5).half().float())
    y2, res = torch.relu(rn(B, H, H, K)), torch.relu(rn(B, H, H, N) * 2.0)
    ya = y2.abs().max().reshape(1)
    M = B * H * H

    def chained():
        return ops.bottleneck_chain(y2, ya, c3, res, c1, torch.zeros(1, device=dev), torch.zeros(1, device=dev))

    def separate():
        xam = torch.zeros(1, device=dev)
        x = ops.conv_bn_act(y2, c3["w"], c3["b"], res, 1, 1, 1, 0, ops.ACT_RELU, w_planes_f16=c3["ph"], w_exp=c3["we"], x_absmax=ya,
                            y_absmax=xam, out_scale=c3["sc"])
        y = ops.conv_bn_act(x, c1["w"], c1["b"], None, 1, 1, 1, 0, ops.ACT_RELU, w_planes_f16=c1["ph"], w_exp=c1["we"], x_absmax=xam,
                            y_absmax=torch.zeros(1, device=dev), out_scale=c1["sc"])
        return x, y

    def timeit(fn):
        for _ in range(2):
            fn()
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]
    r = chained()
    if r is None:
        raise SystemExit("chain8 kernel not available")
    xs, ys = separate()
    print(f"max |x' chained - separate| / max = {((r[0] - xs).abs().max() / xs.abs().max()).item():.2e}, y1': "
          f"{((r[1] - ys).abs().max() / ys.abs().max()).item():.2e}")
    tc, ts = timeit(chained), timeit(separate)
    by_c = 4.0 * M * (K + 2 * N + P)
    by_s = 4.0 * M * (K + 2 * N) + 4.0 * M * (N + P)
    fl = 2.0 * M * N * (K + P)
    print(f"B={B}: chained {tc:.3f} ms  {by_c / tc / 1e6:.0f} GB/s algorithmic  {fl / tc / 1e9:.0f} TF-eq | separate {ts:.3f} ms "
          f"{by_s / ts / 1e6:.0f} GB/s | {ts / tc:.2f}x")


if __name__ == "__main__":
    main()
