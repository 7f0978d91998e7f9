"""Stage-1 extraction pipeline (extract.Extractor; reference loop clip_inference.py:188-271): rows written by the double-buffered
pipeline -- pinned staging, side-stream H2D, device preprocessing, encode_image, fused zero-shot tail, minority flags, ONE D2H per
batch -- equal the straightforward batch-by-batch path bit for bit, and the exported JSON equals the dict the reference's own
statements build."""
import json

import numpy as np
import pytest
import torch

from dbmm_amd import adapter, extract, preprocess, store, synth
from dbmm_amd.clip.model import build_model
from test_store_and_order import _reference_json_entry

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batches(dataset, sizes, H, W, seed, as_float=None, model=None, pinned=False):
    out, k = [], 0
    for bi, b in enumerate(sizes):
        raw = (synth.uniform(seed + bi, "u8", (b, H, W, 3)) * 255.999).to(torch.uint8)
        y, c, g = synth.labels(seed + 100 + bi, b)
        split = torch.arange(k, k + b) % 3
        prefix = "/data/waterbirds/x/049.bird/" if dataset == "waterbirds" else "/data/celeba/img_align_celeba/"
        names = [f"{prefix}{i:06d}.jpg" for i in range(k, k + b)]
        img = raw
        if as_float:
            img = preprocess.preprocess_uniform(raw.to(DEV), model.visual.input_resolution).cpu()
        if pinned:
            img = img.pin_memory()
        out.append((img, (y, g, c, split), names))
        k += b
    return out, k


@pytest.mark.parametrize("dataset,as_float,pinned", [("celeba", False, False), ("waterbirds", False, True), ("celeba", True, False)])
def test_extractor_rows_equal_the_batch_path(dataset, as_float, pinned, tmp_path):
    model = build_model(synth.clip_state_dict(3, "tiny-RN")).cuda()
    R, D = model.visual.input_resolution, model.visual.output_dim
    W = synth.text_matrix(5, D, 2, "zs").cuda()
    sizes = [5, 8, 3, 8, 1, 6]
    batches, n = _batches(dataset, sizes, 90, 74, 11, as_float, model, pinned)
    ex = extract.Extractor(model, W, dataset, max_batch=8)
    path = str(tmp_path / "clip.emb")
    res = ex.run(iter(batches), path, n)
    assert ex.stats["batches"] == len(sizes) and ex.stats["d2h_copies"] == len(sizes) and ex.stats["images"] == n
    s = store.load(path)
    assert len(s) == n and s.dim == D and s.dataset == dataset
    lo = 0
    for img, (y, g, c, split), names in batches:
        b = len(names)
        x = img.to(DEV)
        if x.dtype == torch.uint8:
            x = preprocess.preprocess_uniform(x.contiguous(), R)
        f = model.encode_image(x)
        logits, pred = adapter.zeroshot_tail(f, W, 0.02)
        mi, mip = adapter.minority_flags(dataset, y.cuda(), c.cuda(), pred)
        assert np.array_equal(s.embedding[lo:lo + b], f.cpu().numpy())                     # bit for bit
        assert np.array_equal(s.y_pred[lo:lo + b], pred.cpu().numpy()) and np.array_equal(res["pred"][lo:lo + b], pred.cpu().numpy())
        assert np.array_equal(res["is_minor"][lo:lo + b], mi.cpu().numpy()) and np.array_equal(res["is_minor_pred"][lo:lo + b], mip.cpu().numpy())
        assert np.array_equal(s.y[lo:lo + b], y.numpy()) and np.array_equal(s.confounder[lo:lo + b], c.numpy())
        assert np.array_equal(s.group[lo:lo + b], g.numpy()) and np.array_equal(s.split[lo:lo + b], split.numpy())
        # the reference's softmax / max (clip_inference.py:213-216) picks the same class
        assert torch.equal(torch.max(logits.softmax(dim=-1).cpu(), dim=1)[1], pred.cpu())
        lo += b
    assert s.filenames == [extract.row_key(dataset, nm) for _, _, names in batches for nm in names]
    assert s.filenames[0] == ("049.bird/000000.jpg" if dataset == "waterbirds" else "000000.jpg")
    # JSON shim: the dict the reference's statements build (keys are the basename for CelebA, the last two path parts for Waterbirds)
    jp = str(tmp_path / "clip.json")
    store.export_json(s, jp)
    got = json.load(open(jp))
    emb_t = torch.from_numpy(np.asarray(s.embedding))
    want, i = {}, 0
    for img, (y, g, c, split), names in batches:
        for j, nm in enumerate(names):
            want[extract.row_key(dataset, nm)] = _reference_json_entry(dataset, y[j], g[j], c[j], split[j], emb_t[i], torch.from_numpy(np.asarray(s.y_pred))[i])
            i += 1
    assert got == want


def test_extractor_rejects_oversized_batches_and_short_runs(tmp_path):
    model = build_model(synth.clip_state_dict(3, "tiny-RN")).cuda()
    W = synth.text_matrix(5, model.visual.output_dim, 2, "zs").cuda()
    batches, n = _batches("celeba", [4, 4], 70, 70, 3)
    with pytest.raises(ValueError):
        extract.Extractor(model, W, "celeba", max_batch=2).run(iter(batches), str(tmp_path / "a.emb"), n)
    with pytest.raises(ValueError):
        extract.Extractor(model, W, "celeba", max_batch=4).run(iter(batches), str(tmp_path / "b.emb"), n + 1)   # fewer rows than promised
    with pytest.raises(NotImplementedError):
        extract.Extractor(model, W, "imagenet")
