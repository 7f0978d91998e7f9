#!/usr/bin/env python3
"""Developer tool: device preprocessing throughput vs the host (Pillow) path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dbmm_amd
from PIL import Image
from dbmm_amd import preprocess as PP
from dbmm_amd.clip.clip import _transform          # the host (Pillow) preprocess of clip.load, clip/clip.py:79-86

host_pp = _transform(224)

for name, (H, W), B in [("CelebA 218x178", (218, 178), 512), ("Waterbirds-like 375x500", (375, 500), 512),
                        ("12 MP 3000x4000", (3000, 4000), 16)]:
    rng = np.random.RandomState(0)
    host = [rng.randint(0, 256, (H, W, 3)).astype(np.uint8) for _ in range(min(B, 32))]
    dev = [torch.from_numpy(host[i % len(host)]).cuda() for i in range(B)]
    PP.preprocess_batch(dev, 224); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        PP.preprocess_batch(dev, 224)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for im in host:
        host_pp(Image.fromarray(im, "RGB"))
    t_host = (time.perf_counter() - t0) / len(host)
    print(f"{name}: device {B / t_dev:9.0f} img/s ({t_dev / B * 1e6:7.1f} us/img incl. launch)   "
          f"host Pillow 1 core {1 / t_host:7.0f} img/s ({t_host * 1e3:6.2f} ms/img)")
