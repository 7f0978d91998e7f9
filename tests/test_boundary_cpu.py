"""Host-side members of the drop-in boundary against fixtures generated from the reference's own
functions (oracle/make_golden.py): tokenizer, index construction, stratified split, LR schedule
helpers, clip.load file formats.  CPU only -- these are integer / float64 host computations, no
kernel is involved."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from dbmm_amd import adapter, optim, synth
from dbmm_amd.clip import clip as dclip
from dbmm_amd.clip import simple_tokenizer
from dbmm_amd.clip.model import build_model


def _bpe_path():
    """the OpenAI merges table is data this repository does not ship: $DBMM_BPE_PATH, ~/.cache/clip,
    or (build container only) the reference checkout's copy"""
    try:
        return simple_tokenizer.default_bpe()
    except FileNotFoundError:
        p = "/root/reference/clip/bpe_simple_vocab_16e6.txt.gz"
        return p if os.path.isfile(p) else None


needs_bpe = pytest.mark.skipif(_bpe_path() is None, reason="bpe_simple_vocab_16e6.txt.gz (OpenAI CLIP merges table, not "
                               "shipped) not found: set DBMM_BPE_PATH")


@pytest.fixture()
def tokenizer(monkeypatch):
    monkeypatch.setenv("DBMM_BPE_PATH", _bpe_path())
    monkeypatch.setattr(dclip, "_tokenizer", None)
    yield
    dclip._tokenizer = None


@needs_bpe
def test_tokenize_bit_exact_vs_reference_tokenizer(golden, tokenizer):
    """clip.tokenize (clip/clip.py:197-237) == the reference SimpleTokenizer's id rows, 18 prompts: the
    16 class / spurious / group prompts of both datasets, one with apostrophes, digits, double
    spaces and punctuation runs, and the empty string."""
    prompts = json.load(open(os.path.join(GOLDEN, "tokens_prompts.json")))
    want = golden("tokens.npz")["tokens"]
    got = dclip.tokenize(prompts)
    assert got.dtype == torch.int32 and tuple(got.shape) == want.shape == (len(prompts), 77)
    assert np.array_equal(got.numpy(), want)
    assert np.array_equal(dclip.tokenize(prompts[0]).numpy(), want[:1])        # a bare string is one row


@needs_bpe
def test_tokenize_overflow_and_truncate(tokenizer):
    long = " ".join(["photo"] * 100)
    with pytest.raises(RuntimeError, match="too long for context length"):       # clip/clip.py:234
        dclip.tokenize(long)
    row = dclip.tokenize(long, truncate=True)[0]
    assert row[0] == 49406 and row[-1] == 49407 and (row != 0).all()             # clip/clip.py:231-233
    short = dclip.tokenize("a", context_length=8)
    assert tuple(short.shape) == (1, 8) and short[0, 3:].eq(0).all()


def test_group_index_and_balance_val_vs_reference(golden):
    g = golden("indices.npz")
    y, c, grp = adapter.group_index(g["raw_y"], g["raw_c"])                      # inputs hold -1 / +1 (CelebA csv)
    assert (g["raw_y"] == -1).any() and y.min() == 0 and c.min() == 0
    assert grp.dtype == np.int64 and np.array_equal(grp, g["group"])
    garr = g["balance_garr"][200:900]
    for bsr in (16, 100000):
        np.random.seed(42)
        idx, bs = adapter.balance_val_indices(garr, 4, bsr)
        assert np.array_equal(idx, g[f"balance_idx_{bsr}"]) and bs == int(g[f"balance_bs_{bsr}"])
    # minority flags: the truth tables of clip_inference.py:219-233
    t, s, p = (torch.from_numpy(g[k]) for k in ("minor_t", "minor_s", "minor_p"))
    for ds, pre in (("waterbirds", "wb"), ("celeba", "ca")):
        a, b = adapter.minority_flags(ds, t, s, p)
        assert a.dtype == torch.int64 and np.array_equal(a.numpy(), g[pre + "_is_minor"])
        assert np.array_equal(b.numpy(), g[pre + "_is_minor_pred"])


def test_stratified_split_vs_reference(golden):
    """data/celeba_embeddings_reg.py:95-107 (sklearn train_test_split, random_state=42)."""
    g = golden("split.npz")
    for name in ("a", "b"):
        garr = g[f"{name}_garr"]
        for ts in (0.5, 0.25):
            reg, val = adapter.stratified_split_indices(garr, ts)
            assert np.array_equal(reg, g[f"{name}_reg_{ts}"]) and np.array_equal(val, g[f"{name}_val_{ts}"])
    ds = SimpleNamespace(group_array=g["a_garr"])
    reg_set, val_set = adapter.stratified_split_dataset(ds)
    assert isinstance(reg_set, torch.utils.data.Subset) and reg_set.dataset is ds
    assert np.array_equal(np.asarray(val_set.indices), g["a_val_0.5"])
    assert len(set(reg_set.indices) & set(val_set.indices)) == 0


def test_lr_helpers_vs_reference():
    """demo/util.py:70-115 on the fixture grid, including the cosine branch of adjust_learning_rate_reg,
    which reads a misspelt attribute in the reference and raises AttributeError there and here."""
    j = json.load(open(os.path.join(GOLDEN, "lr_schedule.json")))
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.5)
    fns = dict(adjust_learning_rate=optim.adjust_learning_rate, adjust_learning_rate_reg=optim.adjust_learning_rate_reg,
               warmup_learning_rate=optim.warmup_learning_rate, warmup_learning_rate_reg=optim.warmup_learning_rate_reg)
    seen_err = 0
    for name, cosine, epoch, batch_id, total, want in j["rows"]:
        a = SimpleNamespace(**dict(j["args"], cosine=cosine))
        opt.param_groups[0]["lr"] = 0.5
        call = (lambda: fns[name](a, opt, epoch)) if name.startswith("adjust") else \
               (lambda: fns[name](a, epoch, batch_id, total, opt))
        if want == "AttributeError":
            with pytest.raises(AttributeError):
                call()
            seen_err += 1
        else:
            call()
            assert optim.get_lr(opt) == pytest.approx(want, rel=1e-15, abs=0), (name, cosine, epoch, batch_id)
    assert seen_err > 0


class _Tree(torch.nn.Module):
    """bare module tree holding a state dict under its dotted names (stand-in for the scripted CLIP
    module inside an OpenAI TorchScript archive: clip.load only takes its state_dict, clip/clip.py:126-136)"""
    def __init__(self, sd):
        super().__init__()
        for k, v in sd.items():
            mod, parts = self, k.split(".")
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    mod.add_module(p, torch.nn.Module())
                mod = getattr(mod, p)
            if v.is_floating_point():
                mod.register_parameter(parts[-1], torch.nn.Parameter(v.clone(), requires_grad=False))
            else:
                mod.register_buffer(parts[-1], v.clone())


@pytest.mark.parametrize("arch", ["tiny-RN", "tiny-ViT"])
def test_clip_load_state_dict_file_and_jit_archive(arch, tmp_path):
    sd = synth.clip_state_dict(3, arch)
    want = build_model({k: v.clone() for k, v in sd.items()}).state_dict()
    p_sd, p_jit = str(tmp_path / "m.pt"), str(tmp_path / "m_jit.pt")
    torch.save(sd, p_sd)
    torch.jit.script(_Tree(sd)).save(p_jit)
    with pytest.raises(Exception):
        torch.load(p_jit, map_location="cpu", weights_only=True)                 # really the archive branch
    for path in (p_sd, p_jit):
        with pytest.warns(UserWarning, match="needs an MI355X"):
            model, preprocess = dclip.load(path, device="cpu")
        got = model.state_dict()
        assert got.keys() == want.keys()
        for k in want:
            assert torch.equal(got[k], want[k]), (path, k)
        assert model.visual.input_resolution == preprocess.n_px and not model.training
    with pytest.raises(RuntimeError, match="jit=True"):
        dclip.load(p_sd, device="cpu", jit=True)
    with pytest.raises(RuntimeError, match="not found"):
        dclip.load("RN50", device="cpu", download_root=str(tmp_path))           # a name never downloads
    with pytest.raises(RuntimeError, match="available models"):
        dclip.load("no-such-model", device="cpu")
    assert "RN50" in dclip.available_models() and "ViT-L/14@336px" in dclip.available_models()


def test_plans_are_dropped_when_parameters_change():
    """load_state_dict() and in-place edits must invalidate the cached execution plans (stale fp16 planes
    next to fresh fp32 weights would be silently wrong)."""
    model = build_model(synth.clip_state_dict(3, "tiny-RN"))
    v = model.visual
    v._plan, v._plan_key = {"stub": True}, v._param_key()
    with torch.no_grad():
        v.conv2.weight.mul_(1.0)                              # in-place edit bumps the version counter
    assert v._plan_key != v._param_key()
    v._plan_key = v._param_key()
    model.load_state_dict(model.state_dict())
    assert v._plan is None
    t = model.transformer
    t._planes, t._planes_key = ["stub"], t._weight_key()
    model.load_state_dict(model.state_dict())                 # copy_ into every parameter
    assert t._planes_key != t._weight_key()
