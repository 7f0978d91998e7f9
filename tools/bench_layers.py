#!/usr/bin/env python3
"""Developer tool: per-layer timing of the RN50 conv/GEMM shapes at a given batch (HIP events,
median of n).  Prints achieved TFLOP/s (fp32 MFMA peak 157.3) and algorithmic GB/s.
    python tools/bench_layers.py [--batch 512] [--iters 10] [--filter conv2]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dbmm_amd  # noqa: E402,F401
from dbmm_amd import ops  # noqa: E402


def rn50_layers(B):
    """(name, H, Cin, Cout, k, residual) of every igemm launch of ModifiedResNet-50 @224."""
    L = [("stem2", 112, 32, 32, 3, False), ("stem3", 112, 32, 64, 3, False)]
    inpl, H = 64, 56
    for li, (n, planes) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512)), start=1):
        for bi in range(n):
            stride = 2 if (li > 1 and bi == 0) else 1
            L.append((f"l{li}.{bi}.c1", H, inpl, planes, 1, False))
            L.append((f"l{li}.{bi}.c2", H, planes, planes, 3, False))
            Ho = H // stride
            if bi == 0:
                L.append((f"l{li}.{bi}.ds", Ho, inpl, planes * 4, 1, False))
            L.append((f"l{li}.{bi}.c3", Ho, planes, planes * 4, 1, True))
            inpl, H = planes * 4, Ho
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--filter", default="")
    ap.add_argument("--unique", action="store_true", help="time each distinct shape once")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (dbmm_set_option), repeatable")
    ap.add_argument("--k", type=int, default=0, help="only convs with this kernel size (1 | 3)")
    a = ap.parse_args()
    for o in a.opt:
        n, v = o.split("=")
        ops.set_option(n, int(v))
    dev = "cuda"
    seen, rows = {}, []
    tot_ms = tot_fl = 0.0
    for name, H, Cin, Cout, k, res in rn50_layers(a.batch):
        if (a.filter and a.filter not in name) or (a.k and k != a.k):
            continue
        key = (H, Cin, Cout, k, res)
        if key in seen:
            ms = seen[key]
        else:
            x = torch.randn(a.batch, H, H, Cin, device=dev)
            w, wl = ops.pack_conv_weight(torch.randn(Cout, Cin, k, k, device=dev) * (Cin * k * k) ** -0.5,
                                         chunk_major={"1": True, "32": 32}.get(os.environ.get("BENCH_WL"), False))
            mode = os.environ.get("BENCH_SPLIT", "f16x1")        # f16x1 (exact fp16 weights, one plane) | f16 | bf16 | off
            kw = {}
            if mode == "bf16":
                kw = dict(w_planes=ops.split_planes(w))
            elif mode in ("f16", "f16x1"):
                if mode == "f16x1":
                    w = w.half().float()
                ph, we, _ = ops.split_planes_f16(w, allow_single=mode == "f16x1")
                kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=x.abs().max().reshape(1),
                          y_absmax=torch.zeros(1, device=dev))
                if mode == "f16x1":
                    kw["out_scale"] = 0.5 + torch.rand(Cout, device=dev)
            b = torch.randn(Cout, device=dev)
            r = torch.randn(a.batch, H, H, Cout, device=dev) if res else None
            pad = 1 if k == 3 else 0
            for _ in range(2):
                ops.conv_bn_act(x, w, b, r, k, k, 1, pad, ops.ACT_RELU, wl, **kw)
            ts = []
            for _ in range(a.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.conv_bn_act(x, w, b, r, k, k, 1, pad, ops.ACT_RELU, wl, **kw); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort(); ms = ts[len(ts) // 2]
            seen[key] = ms
            del x, w, r, kw
        M = a.batch * H * H
        fl = 2.0 * M * Cout * Cin * k * k
        by = 4.0 * M * (Cin + Cout * (2 if res else 1))
        tot_ms += ms; tot_fl += fl
        rows.append((name, M, Cout, Cin * k * k, ms, fl / ms / 1e9, by / ms / 1e6))
    for name, M, N, K, ms, tf, gbs in rows:
        print(f"{name:10s} M={M:8d} N={N:5d} K={K:5d} {ms:8.3f} ms {tf:7.1f} TF {gbs:7.0f} GB/s")
    print(f"TOTAL {tot_ms:.2f} ms  {tot_fl / tot_ms / 1e9:.1f} TF avg  ({a.batch / tot_ms * 1e3:.0f} img/s conv-only)")


if __name__ == "__main__":
    main()
