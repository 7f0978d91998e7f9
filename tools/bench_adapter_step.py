"""Developer tool (GPU box): the adapter-only train step (BASELINE configs[0] shape) -- us/step for the fused C step, at a batch.
    python tools/bench_adapter_step.py [B] [D] [steps]          (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, time, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbmm_amd  # noqa
from dbmm_amd import adapter, optim, synth
from types import SimpleNamespace

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
d = tempfile.mkdtemp()
paths = []
for nm, C in (("c", 2), ("s", 2), ("g", 4)):
    m = synth.text_matrix(1, D, C, nm); p = os.path.join(d, nm + ".json")
    json.dump({f"{nm}{i}": m[:, i].tolist() for i in range(C)}, open(p, "w")); paths.append(p)
ad = adapter.Adapter(D, 128); ad.load_state_dict(synth.adapter_state_dict(3, D, 128))
clf = adapter.CustomCLIP(ad, *paths, temperature=0.01).cuda().train()
opt = optim.set_optimizer(SimpleNamespace(learning_rate=0.1, momentum=0.9, weight_decay=5e-5), clf)
x = synth.normal(5, f"x{B}", (B, D), 0.5).cuda()
y, c, g = (t.cuda() for t in synth.labels(6, B))
for _ in range(20):
    clf.train_step(x, y, opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss, logits, rows = clf.train_step(x, y, opt)
torch.cuda.synchronize()
print(f"B={B} D={D}: {(time.perf_counter() - t0) / steps * 1e6:.1f} us/step, loss {loss.item():.5f}")
