"""encode_image / encode_text parity on the MI355X: the HIP path against (a) the golden
vectors produced by the reference's own modules and (b) the oracle on the same seeded inputs.
Tolerances: BASELINE.json asks for cosine logits within 1e-3 (fp32) at T = 0.01/0.02."""
import numpy as np
import pytest
import torch

import clip_oracle as CO
from conftest import relerr, summary
from dbmm_amd import adapter, synth
from dbmm_amd.clip.model import build_model

pytestmark = pytest.mark.gpu
ARCHS = ["tiny-RN", "tiny-RN-w32", "tiny-ViT", "RN50", "ViT-B/32", "ViT-L/14@336px"]   # last: full depth, B = 1 (BASELINE configs[4])


def gname(arch):
    return "clip_" + arch.replace("/", "-").replace("@", "-") + ".npz"


@pytest.fixture(scope="module")
def models():
    cache = {}

    def get(arch, seed):
        if (arch, seed) not in cache:
            cache.clear()
            cache[(arch, seed)] = build_model(synth.clip_state_dict(seed, arch)).cuda()
        return cache[(arch, seed)]
    return get


@pytest.mark.parametrize("arch", ARCHS)
def test_encode_image_vs_golden(arch, golden, models):
    g = golden(gname(arch))
    seed, B, res = int(g["seed"]), int(g["batch"]), int(g["res"])
    model = models(arch, seed)
    img = synth.images(seed + 100, B, res).cuda()
    if "ViT" in arch:
        out = model.encode_image(img)
    else:
        out, stages = model.visual(img, return_stages=True)
        for k, t in stages.items():                     # NHWC, same order as the fixture
            sums, sample = summary(t)
            ref = g[f"{k}_sample"]
            assert np.abs(sample - ref).max() <= 1e-4 * np.abs(ref).max(), (arch, k)
            assert abs(sums[1] - g[f"{k}_sums"][1]) <= 1e-4 * g[f"{k}_sums"][1], (arch, k)
    assert out.shape == g["embedding"].shape and out.dtype == torch.float32
    assert relerr(out.cpu(), g["embedding"]) < 5e-5
    # zero-shot tail: temperature-scaled cosine logits within 1e-3, predictions bit-exact
    W = synth.text_matrix(seed + 1, out.shape[1], 2, "zs").cuda()
    logits, pred = adapter.zeroshot_tail(out, W)
    assert np.abs(logits.cpu().numpy() - g["zs_logits"]).max() < 1e-3
    assert (pred.cpu().numpy() == g["zs_pred"]).all() and pred.dtype == torch.int64


@pytest.mark.parametrize("arch", ARCHS)
def test_encode_text_vs_golden(arch, golden, models):
    g = golden(gname(arch))
    model = models(arch, int(g["seed"]))
    out = model.encode_text(torch.from_numpy(g["tokens"]).cuda())
    assert relerr(out.cpu(), g["text_embedding"]) < 5e-5


@pytest.mark.parametrize("arch,B", [("tiny-RN", 5), ("tiny-ViT", 3), ("RN50", 3), ("ViT-L14-336-2L", 2)])
def test_encode_image_vs_oracle_other_inputs(arch, B, models):
    """fresh seed / odd batch (ragged M tiles) against the oracle run here on the CPU."""
    seed = 11
    sd = synth.clip_state_dict(seed, arch)
    model = models(arch, seed)
    res = model.visual.input_resolution
    img = synth.images(77, B, res)
    with torch.no_grad():
        ref = CO.encode_image(sd, img)
    out = model.encode_image(img.cuda())
    assert relerr(out.cpu(), ref) < 5e-5
    f = out / out.norm(dim=-1, keepdim=True)
    fr = ref / ref.norm(dim=-1, keepdim=True)
    W = synth.text_matrix(5, ref.shape[1], 4, "w")
    Wn = W / W.norm(dim=0, keepdim=True)
    assert ((f.cpu() @ Wn) / 0.01 - (fr @ Wn) / 0.01).abs().max() < 1e-3


@pytest.mark.parametrize("arch,B", [("RN101", 2), ("RN50x4", 1)])
def test_other_rn_architectures_of_the_reference_cli(arch, B, models):
    """clip_inference.py:280 also offers RN101 and RN50x4: a 23-block stage, and widths (80, 320, ...) that are not multiples
    of 32, which take the general kernels instead of the fp16-pair / halo / chain ones -- against the oracle on the CPU."""
    seed = 5
    sd = synth.clip_state_dict(seed, arch)
    model = models(arch, seed)
    res = model.visual.input_resolution
    img = synth.images(31, B, res)
    with torch.no_grad():
        ref = CO.encode_image(sd, img)
    out = model.encode_image(img.cuda())
    assert tuple(out.shape) == tuple(ref.shape) and relerr(out.cpu(), ref) < 5e-5


def test_clip_forward_and_surface(models):
    model = models("tiny-RN", 3)
    assert model.dtype == torch.float32 and model.visual.input_resolution == 64
    img = synth.images(1, 2, 64).cuda()
    tok = torch.zeros(3, 77, dtype=torch.int32); tok[:, 0] = 510; tok[:, 1] = 7; tok[:, 2] = 511
    li, lt = model(img, tok.cuda())
    assert li.shape == (2, 3) and lt.shape == (3, 2)
    sd = synth.clip_state_dict(3, "tiny-RN")
    with torch.no_grad():
        fi, ft = CO.encode_image(sd, img.cpu()), CO.encode_text(sd, tok)
        fi = fi / fi.norm(dim=1, keepdim=True); ft = ft / ft.norm(dim=1, keepdim=True)
        ref = sd["logit_scale"].exp() * fi @ ft.t()
    assert (li.cpu() - ref).abs().max() < 1e-3


def test_large_batch_property_rn50(models):
    """at a batch the oracle would take minutes for: rows are independent, so any row of a big
    batch must equal the same image encoded alone.  Not bit-exact: tile shapes and the stream-K
    split points depend on the batch, which changes the (deterministic) fp32 summation order."""
    model = models("RN50", 2)
    img = synth.images(9, 48, 224).cuda()
    big = model.encode_image(img)
    small = model.encode_image(img[17:19].contiguous())
    assert torch.isfinite(big).all()
    assert relerr(big[17:19].cpu(), small.cpu()) < 1e-5
    again = model.encode_image(img)
    assert torch.equal(big, again)               # same shape -> same schedule -> bit-identical


def test_rn_tower_wide_dynamic_range_and_unrounded_weights():
    """hostile statistics for the fp16-pair path: BatchNorm scales spread over four decades,
    activations with huge outlier channels, conv weights that are NOT fp16-representable in the
    checkpoint (the reference rounds them on load, clip/model.py:433 -- so must we).  Batch 40 at
    96 px gives every kernel family work (halo, pooled, dual-source, 1x1)."""
    from dbmm_amd.clip.model import build_model, _as_loaded
    arch, seed = "tiny-RN-w32", 21
    sd = synth.clip_state_dict(seed, arch)
    for i, k in enumerate(sorted(k for k in sd if k.startswith("visual.") and k.endswith(".weight") and ".bn" in k
                                 or k.startswith("visual.bn") and k.endswith(".weight")
                                 or "downsample.1.weight" in k)):
        sd[k] = sd[k] * (50.0 if i % 3 == 0 else (0.02 if i % 3 == 1 else 1.0))
    for k in list(sd):
        if k.startswith("visual.") and ("conv" in k or "downsample.0" in k) and k.endswith(".weight"):
            sd[k] = sd[k] * 1.0003                      # no longer exact in fp16
    model = build_model({k: v.clone() for k, v in sd.items()}).cuda()
    loaded = {k: _as_loaded(k, v) for k, v in sd.items()}          # what the reference's build_model would hold
    img = synth.images(5, 40, 96)
    img[:, 1] *= 30.0                                    # one colour plane dominates the stem
    with torch.no_grad():
        ref = CO.encode_image(loaded, img)
    out = model.encode_image(img.cuda())
    assert torch.isfinite(out).all()
    assert relerr(out.cpu(), ref) < 5e-5
