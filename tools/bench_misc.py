#!/usr/bin/env python3
"""Developer tool: timings outside the RN50 bench line -- ViT-B/32 encode_image, encode_text,
and the adapter-only step (BASELINE configs[0] shape: bs=256, D=1024) in steps/s."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dbmm_amd  # noqa: E402,F401
from dbmm_amd import adapter, optim, synth  # noqa: E402
from dbmm_amd.clip.model import build_model  # noqa: E402


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    which = sys.argv[1:] or ["adapter", "vit", "text"]
    if "adapter" in which:
        D, H = 1024, 128
        d = tempfile.mkdtemp()
        paths = []
        for nm, C in (("c", 2), ("s", 2), ("g", 4)):
            m = synth.text_matrix(1, D, C, nm); p = os.path.join(d, nm + ".json")
            json.dump({f"{nm}{i}": m[:, i].tolist() for i in range(C)}, open(p, "w")); paths.append(p)
        from types import SimpleNamespace
        for B in (256, 1024):
            ad = adapter.Adapter(D, H); ad.load_state_dict(synth.adapter_state_dict(3, D, H))
            clf = adapter.CustomCLIP(ad, *paths).cuda().train()
            opt = optim.set_optimizer(SimpleNamespace(learning_rate=0.1, momentum=0.9, weight_decay=5e-5), clf)
            x = synth.normal(5, "x", (B, D), 0.5).cuda(); y = synth.labels(6, B)[0].cuda()
            crit = torch.nn.CrossEntropyLoss()

            def fused():
                loss, _, _ = clf.loss(x, y); opt.zero_grad(); loss.backward(); opt.step()

            def dropin():
                loss = crit(clf(x.detach()), y); opt.zero_grad(); loss.backward(); opt.step()
            def onecall():
                clf.train_step(x, y, opt)
            tf, td, to = timeit(fused, 50, 5), timeit(dropin, 50, 5), timeit(onecall, 200, 10)
            print(f"adapter step B={B}: one-call train_step {to * 1e6:.0f} us ({B / to:.0f} samples/s), "
                  f"autograd+fused CE {tf * 1e6:.0f} us, drop-in {td * 1e6:.0f} us")
            if B == 256:
                from dbmm_amd import trainer
                N = 64 * 256
                table = trainer.EmbeddingTable(synth.normal(7, "tab", (N, D), 0.5).numpy(), *[t.numpy() for t in synth.labels(8, N)[:2]])
                trainer.train_epoch(table, clf, opt, B)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3):
                    trainer.train_epoch(table, clf, opt, B)
                torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 3
                print(f"device-resident epoch: {N} samples, bs {B}: {te * 1e3:.1f} ms/epoch = {te / (N // B) * 1e6:.0f} us/step "
                      f"({N / te:.0f} samples/s) incl. batch gather + group counters")
    if "vit" in which:
        model = build_model(synth.clip_state_dict(2, "ViT-B/32")).cuda()
        for B in (64, 512):
            img = torch.randn(B, 3, 224, 224, device="cuda")
            t = timeit(lambda: model.encode_image(img), 5, 2)
            print(f"ViT-B/32 encode_image B={B}: {t * 1e3:.2f} ms  {B / t:.0f} img/s  ({B / t * 8.82e9 / 1e12:.1f} TF)")
        if "text" in which:
            tok = torch.zeros(8, 77, dtype=torch.int32); tok[:, 0] = 49406; tok[:, 1:6] = 320; tok[:, 6] = 49407
            tok = tok.cuda()
            t = timeit(lambda: model.encode_text(tok), 5, 2)
            print(f"encode_text n=8: {t * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
