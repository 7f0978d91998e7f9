"""Developer tool (GPU box): the fp16-mode conv kernels on the RN50 layer shapes at a batch: time, TFLOP/s, TB/s (algorithmic bytes).
    python tools/bench_conv_f16.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbmm_amd  # noqa
from dbmm_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator(device="cuda"); g.manual_seed(0)
print("--- 1x1: H, Cin, Cout, residual")
for H, Cin, Cout, res in [(56, 64, 64, 0), (56, 256, 64, 0), (56, 64, 256, 1), (56, 256, 128, 0), (28, 128, 512, 1), (28, 512, 128, 0), (28, 256, 512, 0),
                          (28, 512, 256, 0), (14, 256, 1024, 1), (14, 1024, 256, 0), (14, 512, 1024, 0), (14, 1024, 512, 0), (7, 512, 2048, 1),
                          (7, 2048, 512, 0), (7, 1024, 2048, 0)]:
    M = B * H * H
    x = torch.randn((M, Cin), device="cuda", generator=g).half(); w = (torch.randn((Cout, Cin), device="cuda", generator=g) * Cin ** -0.5).half()
    sc = torch.ones(Cout, device="cuda"); b = torch.zeros(Cout, device="cuda")
    r = torch.randn((M, Cout), device="cuda", generator=g).half() if res else None
    by = 2 * (M * Cin + M * Cout * (2 if res else 1) + Cout * Cin); fl = 2.0 * M * Cin * Cout
    row = f"{H:3d} {Cin:5d} {Cout:5d} {res}:"
    for mode in (0, 2):
        ops.set_option("conv1x1_stream", mode)
        ms = t(lambda: ops.conv1x1_f16(x, w, sc, b, residual=r))
        row += f"   mode {mode}: {ms * 1e3:7.1f} us {fl / ms / 1e9:7.1f} TF {by / ms / 1e9:6.2f} TB/s"
    print(row)
ops.set_option("conv1x1_stream", 1)
print("--- 3x3: H, Cin, Cout, pool")
for H, Cin, Cout, pool in [(112, 32, 32, 1), (112, 32, 64, 2), (56, 64, 64, 1), (56, 128, 128, 2), (28, 128, 128, 1), (28, 256, 256, 2), (14, 256, 256, 1),
                           (14, 512, 512, 2), (7, 512, 512, 1)]:
    x = torch.randn((B, H, H, Cin), device="cuda", generator=g).half()
    w = (torch.randn((Cout, 9 * Cin), device="cuda", generator=g) * (9 * Cin) ** -0.5).half()
    sc = torch.ones(Cout, device="cuda"); b = torch.zeros(Cout, device="cuda")
    ms = t(lambda: ops.conv3x3_f16(x, w, sc, b, pool=pool))
    M = B * H * H
    fl = 2.0 * M * Cout * 9 * Cin; by = 2 * (M * Cin + M * Cout // (pool * pool) + Cout * 9 * Cin)
    print(f"{H:3d} {Cin:5d} {Cout:5d} pool {pool}: {ms * 1e3:7.1f} us {fl / ms / 1e9:7.1f} TF {by / ms / 1e9:6.2f} TB/s")
