#!/usr/bin/env python3
"""Developer diagnostic: the exact flow of tests/test_gpu_adapter.py for one (D, B), printing every error."""
import copy, io, contextlib, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import dbmm_amd
from dbmm_amd import adapter, optim, synth
from types import SimpleNamespace

def summary(t, nsample=256):
    f = t.detach().double().flatten().cpu(); step = max(1, f.numel() // nsample)
    return f[::step][:nsample].float().numpy()

D, B, H = int(sys.argv[1]), int(sys.argv[2]), 128
g = np.load(os.path.join(ROOT, "tests", "golden", "adapter.npz" if D == 1024 else f"adapter_D{D}.npz"))
d = tempfile.mkdtemp(); paths = []
for nm, (m, cols) in dict(clip_class=(synth.text_matrix(1, D, 2, "class"), ["c0", "c1"]), clip_spurious=(synth.text_matrix(1, D, 2, "spurious"), ["s0", "s1"]),
                          clip_group=(synth.text_matrix(1, D, 4, "group"), ["g0", "g1", "g2", "g3"])).items():
    p = os.path.join(d, nm + ".json"); json.dump({n: m[:, i].numpy().tolist() for i, n in enumerate(cols)}, open(p, "w")); paths.append(p)
x = synth.normal(5, f"x{B}", (B, D), 0.5).cuda(); y, c, grp = (t.cuda() for t in synth.labels(6, B))
ns = SimpleNamespace(learning_rate=0.1, learning_rate_reg=0.05, momentum=0.9, weight_decay=5e-5)
crit = torch.nn.CrossEntropyLoss()
def cmp(tag, name, t):
    t = t.detach().float().cpu()
    if f"{tag}/{name}" in g.files:
        ref = g[f"{tag}/{name}"]; e = np.abs(t.numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
    else:
        ref = g[f"{tag}/{name}_sample"]; e = np.abs(summary(t) - ref).max() / max(np.abs(ref).max(), 1e-30)
    print(f"  {tag}/{name}: {e:.2e}")
for use_group in (False, True):
    tag = f"custom_B{B}_{'group' if use_group else 'class'}"
    ad = adapter.Adapter(D, H); ad.load_state_dict(synth.adapter_state_dict(3, D, H))
    clf = adapter.CustomCLIP(ad, *paths, temperature=0.01).cuda(); opt = optim.set_optimizer(ns, clf)
    labels = grp if use_group else y
    clf.train()
    for step in range(3):
        logits = clf(x.detach(), use_group); loss = crit(logits, labels); opt.zero_grad(); loss.backward(); opt.step()
    for k, v in clf.state_dict().items():
        if v.dtype.is_floating_point: cmp(tag + "/after3", k, v)
    clf.eval()
    with torch.no_grad(): ev, evs = clf(x), clf.forward_spurious(x)
    stage1 = clf
for ni in (True, False):
    for use_group in (False, True):
        tag = f"multi_B{B}_{'ni' if ni else 'rn'}_{'group' if use_group else 'class'}"
        old = copy.deepcopy(stage1)
        new_ad = adapter.Adapter(D, H); new_ad.load_state_dict(synth.adapter_state_dict(4, D, H))
        with contextlib.redirect_stdout(io.StringIO()):
            ma = adapter.MultipleAdapter(old, new_ad, init_near_identity=ni, ebd_weight=0.5).cuda()
        opt = optim.set_optimizer_reg(ns, ma); labels = grp if use_group else y
        ma.train()
        logits = ma(x.detach(), use_group); loss = crit(logits, labels); opt.zero_grad(); loss.backward()
        print(tag, "logits abs err", (logits.detach().cpu() - torch.from_numpy(g[tag + "/step0/logits"])).abs().max().item())
        for n, p in ma.named_parameters():
            if p.grad is not None: cmp(tag + "/step0/grad", n, p.grad)
