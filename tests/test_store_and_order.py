"""CPU tests for the 'next' rows of SURVEY section 8f: binary embedding store + JSON shim, and the
DataLoader-order replication used by the device-resident training loop."""
import json
import os

import numpy as np
import torch

from dbmm_amd import store, synth, trainer


def _reference_json_entry(dataset, y, group, cof, split, img_emb, pred):
    """the statements of clip_inference.py:235-257 on torch scalars (restated, not imported: the
    script's own imports -- clip, sklearn, run.sweeping -- do not resolve offline)"""
    if dataset == "waterbirds":
        key_list = ['y', 'place', 'group', 'split', 'image_embedding', 'y_pred']
        d = dict.fromkeys(key_list)
        d['y'] = str(y.cpu().numpy()); d['group'] = str(group.cpu().numpy()); d['place'] = str(cof.cpu().numpy())
    else:
        key_list = ['blond', 'male', 'group', 'split', 'image_embedding', 'y_pred']
        d = dict.fromkeys(key_list)
        d['blond'] = str(y.cpu().numpy()); d['group'] = str(group.cpu().numpy()); d['male'] = str(cof.cpu().numpy())
    d['split'] = str(split.cpu().numpy())
    d['image_embedding'] = img_emb.clone().detach().cpu().numpy().tolist()
    d['y_pred'] = str(pred.clone().detach().cpu().numpy())
    return d


def test_store_roundtrip_and_json_schema(tmp_path):
    n, D = 37, 64
    emb = synth.normal(1, "e", (n, D))
    y, c, g = synth.labels(2, n)
    split = torch.arange(n) % 3
    pred = (y + 1) % 2
    for dataset in ("celeba", "waterbirds"):
        names = [f"{'049.bird/' if dataset == 'waterbirds' else ''}{i:06d}.jpg" for i in range(n)]
        p = store.save(str(tmp_path / f"{dataset}.emb"), emb.numpy(), y.numpy(), c.numpy(), g.numpy(), split.numpy(),
                       pred.numpy(), names, dataset)
        s = store.load(p)
        assert len(s) == n and s.dim == D and s.dataset == dataset
        assert np.array_equal(s.embedding, emb.numpy()) and np.array_equal(s.group, g.numpy())
        assert s.filenames == names and np.array_equal(s.select(2), np.nonzero(split.numpy() == 2)[0])
        jp = str(tmp_path / f"{dataset}.json")
        store.export_json(s, jp)
        got = json.load(open(jp))
        want = {nm: _reference_json_entry(dataset, y[i], g[i], c[i], split[i], emb[i], pred[i]) for i, nm in enumerate(names)}
        assert got == want and list(got[names[0]].keys()) == list(want[names[0]].keys())
        # ... and the text is byte-identical to json.dump of the reference-built dict
        assert open(jp).read() == json.dumps(want)
        back = store.load(store.import_json(jp, str(tmp_path / f"{dataset}_2.emb")))
        assert back.dataset == dataset and np.array_equal(back.embedding, emb.numpy())
        for f in ("y", "confounder", "group", "split", "y_pred"):
            assert np.array_equal(getattr(back, f), getattr(s, f))


def test_store_rejects_garbage(tmp_path):
    p = tmp_path / "x.emb"
    p.write_bytes(b"not a store")
    try:
        store.load(str(p))
        raise AssertionError("expected ValueError")
    except ValueError:
        pass


def test_dataloader_shuffle_order_matches_torch():
    from torch.utils.data import DataLoader
    for n, bs in ((37, 8), (256, 256), (1000, 64)):
        torch.manual_seed(42)
        ref = [int(i) for b in DataLoader(list(range(n)), shuffle=True, batch_size=bs) for i in b]
        torch.manual_seed(42)
        assert trainer.dataloader_shuffle_order(n).tolist() == ref
    # two consecutive epochs consume the global RNG like two DataLoader iterations do
    torch.manual_seed(7)
    dl = DataLoader(list(range(50)), shuffle=True, batch_size=50)
    e1 = [int(i) for b in dl for i in b]; e2 = [int(i) for b in dl for i in b]
    torch.manual_seed(7)
    assert trainer.dataloader_shuffle_order(50).tolist() == e1 and trainer.dataloader_shuffle_order(50).tolist() == e2


def test_incremental_writer_equals_save(tmp_path):
    """store.Writer (rows appended batch by batch by extract.Extractor) produces the file store.save writes, and the same JSON text"""
    n, D = 53, 32
    emb = synth.normal(3, "e", (n, D))
    y, c, g = synth.labels(4, n)
    split, pred = torch.arange(n) % 3, (y + 1) % 2
    names = [f"img_{i:05d}.jpg" for i in range(n)]
    a = store.save(str(tmp_path / "a.emb"), emb.numpy(), y.numpy(), c.numpy(), g.numpy(), split.numpy(), pred.numpy(), names, "celeba")
    w = store.Writer(str(tmp_path / "b.emb"), n, D, "celeba")
    for lo in range(0, n, 16):
        hi = min(n, lo + 16)
        w.append(emb[lo:hi].numpy(), y[lo:hi], c[lo:hi], g[lo:hi], split[lo:hi], pred[lo:hi], names[lo:hi])
    w.close()
    A, B = store.load(a), store.load(str(tmp_path / "b.emb"))
    assert np.array_equal(A.embedding, B.embedding) and A.filenames == B.filenames and len(B) == n
    for f in ("y", "confounder", "group", "split", "y_pred"):
        assert np.array_equal(getattr(A, f), getattr(B, f))
    store.export_json(A, str(tmp_path / "a.json")); store.export_json(B, str(tmp_path / "b.json"))
    assert open(tmp_path / "a.json").read() == open(tmp_path / "b.json").read()
    w = store.Writer(str(tmp_path / "c.emb"), 4, D)
    w.append(emb[:2].numpy(), y[:2], c[:2], g[:2], split[:2], pred[:2], names[:2])
    try:
        w.close()
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
    assert not os.path.exists(tmp_path / "c.emb")              # a short store is never renamed into place


def test_prompt_templates_are_the_references():
    """templates.py holds the reference's eight prompts per dataset (fixture: tests/golden/tokens_prompts.json, written from the
    reference's template modules by oracle/make_golden.py)"""
    from dbmm_amd import templates
    from conftest import GOLDEN
    ref = json.load(open(os.path.join(GOLDEN, "tokens_prompts.json")))
    ours = [p for ds in ("celeba", "waterbirds") for which in ("class", "spurious", "group") for p in templates.prompts(ds, which)]
    assert ours == ref[:16]
