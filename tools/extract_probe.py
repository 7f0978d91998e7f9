#!/usr/bin/env python3
"""Developer tool: the from-host extraction pipeline (extract.Extractor) for a few batches -- run under
`rocprofv3 --kernel-trace --memory-copy-trace --stats` to see whether the H2D / D2H copies overlap the encoder."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dbmm_amd  # noqa: E402,F401
from dbmm_amd import extract, synth  # noqa: E402
from dbmm_amd.clip.model import build_model  # noqa: E402

n, Bl = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 1024
dev = "cuda"
model = build_model(synth.clip_state_dict(2, "RN50")).to(dev)
Wz = synth.text_matrix(12, 1024, 2, "zs").to(dev)
raw = (synth.uniform(77, "u8img", (64, 218, 178, 3)) * 255.999).to(torch.uint8)
host = raw.repeat(Bl // 64, 1, 1, 1).contiguous().pin_memory()
y, c, g = synth.labels(6, Bl)
names = [f"{i:06d}.jpg" for i in range(Bl)]
d = tempfile.mkdtemp()


def batches(k):
    for _ in range(k):
        yield host, (y, g, c, torch.zeros(Bl, dtype=torch.int64)), names


ex = extract.Extractor(model, Wz, "celeba", max_batch=Bl)
ex.run(batches(2), os.path.join(d, "w.emb"), 2 * Bl)
torch.cuda.synchronize()
marks = []
orig = ex._drain


def timed_drain(slot, writer, acc):
    t0 = time.perf_counter(); slot.done.synchronize(); t1 = time.perf_counter()
    orig(slot, writer, acc)
    marks.append((t1 - t0, time.perf_counter() - t1))


ex._drain = timed_drain
orig_compute = ex._compute
enq = []


def timed_compute(slot, b):
    t = time.perf_counter(); orig_compute(slot, b); enq.append(time.perf_counter() - t)


ex._compute = timed_compute
for k in (n // 3, n, n):
    marks.clear(); enq.clear()
    t0 = time.perf_counter()
    ex.run(batches(k), os.path.join(d, "c.emb"), k * Bl)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print({kk: round(v * 1e3 / k, 2) for kk, v in ex.stats.items() if kk.startswith("t_")}); [ex.stats.__setitem__(kk, 0.0) for kk in ex.stats if kk.startswith("t_")]
    print(f"{k} batches: {dt * 1e3:.1f} ms total = {dt / k * 1e3:.2f} ms per batch; enqueue of the device work (ms): "
          f"{[round(e * 1e3, 1) for e in enq]}; per drain: wait for the GPU / host write (ms):",
          [(round(a * 1e3, 1), round(b * 1e3, 1)) for a, b in marks])
# the same device work on a resident batch
from dbmm_amd import adapter, preprocess as PP
dev_raw = host.to(dev)
for _ in range(2):
    adapter.zeroshot_tail(model.encode_image(PP.preprocess_uniform(dev_raw, 224)).float(), Wz, 0.02)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(n):
    adapter.zeroshot_tail(model.encode_image(PP.preprocess_uniform(dev_raw, 224)).float(), Wz, 0.02)
torch.cuda.synchronize()
print(f"device-resident: {(time.perf_counter() - t1) / n * 1e3:.2f} ms per batch")
