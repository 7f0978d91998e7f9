"""Pins oracle/ against the golden vectors produced by the reference's own modules
(oracle/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import adapter_oracle as AO
import clip_oracle as CO
from conftest import GOLDEN, relerr, summary
from dbmm_amd import synth

ARCHS = ["tiny-RN", "tiny-RN-w32", "tiny-ViT", "RN50", "ViT-B/32", "ViT-L/14@336px"]   # last: full depth, B = 1 (BASELINE configs[4])


def gname(arch):
    return "clip_" + arch.replace("/", "-").replace("@", "-") + ".npz"


@pytest.mark.parametrize("arch", ARCHS)
def test_encode_image_and_text(arch, golden):
    g = golden(gname(arch))
    seed, B, res = int(g["seed"]), int(g["batch"]), int(g["res"])
    sd = synth.clip_state_dict(seed, arch)
    img = synth.images(seed + 100, B, res)
    torch.set_num_threads(8)
    with torch.no_grad():
        if "ViT" in arch:
            out = CO.vit_encode_image(sd, img)
        else:
            out, stages = CO.rn_encode_image(sd, img, return_stages=True)
            for k, t in stages.items():
                sums, sample = summary(t.permute(0, 2, 3, 1))
                assert np.allclose(sample, g[f"{k}_sample"], rtol=2e-5, atol=2e-5 * np.abs(g[f"{k}_sample"]).max())
                assert abs(sums[1] - g[f"{k}_sums"][1]) <= 2e-5 * g[f"{k}_sums"][1]
        assert relerr(out, g["embedding"]) < 5e-6
        txt = CO.encode_text(sd, torch.from_numpy(g["tokens"]))
        assert relerr(txt, g["text_embedding"]) < 5e-6
        W = synth.text_matrix(seed + 1, out.shape[1], 2, "zs")
        logits, _, pred = CO.zeroshot_tail(torch.from_numpy(g["embedding"]), W)
        assert np.abs(logits.numpy() - g["zs_logits"]).max() < 1e-4
        assert (pred.numpy() == g["zs_pred"]).all()


def _text_mats(D=1024):
    return (synth.text_matrix(1, D, 2, "class"), synth.text_matrix(1, D, 2, "spurious"),
            synth.text_matrix(1, D, 4, "group"))


@pytest.mark.parametrize("D,B", [(1024, 4), (1024, 256), (1024, 1024), (512, 256), (512, 4096), (768, 256), (768, 8192)])
def test_adapter_steps(D, B, golden):
    g = golden("adapter.npz" if D == 1024 else f"adapter_D{D}.npz")
    H = 128
    tcls, tsp, tgrp = _text_mats(D)
    x = synth.normal(5, f"x{B}", (B, D), 0.5)
    y, c, grp = synth.labels(6, B)

    def check(tag, name, t, tol=3e-5):
        if tag.endswith("/grad") and name.endswith("layers.0.bias"):
            # the bias in front of a train-mode BatchNorm has an analytically zero gradient:
            # both sides hold rounding noise only
            assert t.abs().max() < 1e-6 and np.abs(g[f"{tag}/{name}"]).max() < 1e-6
        elif f"{tag}/{name}" in g.files:
            assert relerr(t, g[f"{tag}/{name}"]) < tol, (tag, name)
        else:
            sums, sample = summary(t)
            ref = g[f"{tag}/{name}_sample"]
            assert np.abs(sample - ref).max() <= tol * max(np.abs(ref).max(), 1e-6), (tag, name)

    stage1 = None
    for use_group in (False, True):
        tag = f"custom_B{B}_{'group' if use_group else 'class'}"
        sd = {"adapter." + k: v.clone() for k, v in synth.adapter_state_dict(3, D, H).items()}
        bufs = {}
        labels, text = (grp, tgrp) if use_group else (y, tcls)
        for step in range(3):
            loss, logits, grads = AO.train_step(sd, bufs, x, labels, text, 0.1)
            if step == 0:
                check(tag + "/step0", "logits", logits)
                assert abs(loss.item() - float(g[tag + "/step0/loss"])) < 1e-5 * max(1, abs(loss.item()))
                for k, v in grads.items():
                    check(tag + "/step0/grad", k, v)
        for k, v in sd.items():
            if v.dtype.is_floating_point:
                check(tag + "/after3", k, v)
        assert int(sd["adapter.layers.1.num_batches_tracked"]) == int(g[tag + "/after3/adapter.layers.1.num_batches_tracked"])
        ev = AO.custom_clip_logits(sd, x, tcls, 0.01, train=False)
        check(tag + "/eval", "logits", ev)
        check(tag + "/eval", "logits_spurious", AO.custom_clip_logits(sd, x, tsp, 0.01, train=False))
        if not use_group:
            assert (AO.group_counts(ev, y, grp) == g[tag + "/counts"]).all()
        if B == min(b for b in (256, 1024, 4096, 8192) if f"custom_B{b}_group/step0/loss" in g.files) and use_group:
            for k, v in sd.items():                  # the stored stage-1 end state is this run's (reference vs oracle: 3e-5)
                if v.dtype.is_floating_point:
                    assert relerr(v, g["stage1/" + k]) < 3e-5, k
    # stage 2 starts from the reference's stored stage-1 end state (bit for bit), new adapter from the stored seed
    stage1 = {k[len("stage1/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("stage1/")}
    for ni in (True, False):
        for use_group in (False, True):
            tag = f"multi_B{B}_{'ni' if ni else 'rn'}_{'group' if use_group else 'class'}"
            sd = {"old_cls." + k: v.clone() for k, v in stage1.items()}
            new = {k: v.clone() for k, v in stage1.items()} if ni else \
                {"adapter." + k: v for k, v in synth.adapter_state_dict(int(g[tag + "/new_seed"]), D, H).items()}
            sd.update({k.replace("adapter.", "new_adapter.", 1): v.clone() for k, v in new.items()})
            bufs = {}
            labels, text = (grp, tgrp) if use_group else (y, tcls)
            for step in range(3):
                loss, logits, grads = AO.train_step(sd, bufs, x, labels, text, 0.05, multiple=True)
                if step == 0:
                    check(tag + "/step0", "logits", logits)
                    for k, v in grads.items():
                        check(tag + "/step0/grad", k, v)
            # every committed trajectory is well conditioned (the generator picks the start seed): the reference's own
            # 1-ulp input sensitivity (x 4) stays at its 2e-5 floor, which is also what the oracle is held to
            ttol = max(3e-5, float(g[tag + "/traj_tol"]))
            assert ttol <= 1e-4, (tag, ttol)
            for k, v in sd.items():
                if v.dtype.is_floating_point:
                    check(tag + "/after3", k, v, ttol)
            check(tag + "/eval", "logits", AO.multiple_adapter_logits(sd, x, tcls, 0.01, train=False), 3e-5)


@pytest.mark.parametrize("arch", ARCHS)
def test_fp16_path_fixtures(arch, golden):
    """clip_<arch>_f16.npz = the reference's own fp16 path run on the CPU (oracle/make_golden.py gen_clip_f16): values are
    fp16-representable, sit one fp16 rounding history (~1e-3 of the maximum) from the fp32 fixture, and the recorded
    distance is the real one."""
    g, h = golden(gname(arch)), golden(gname(arch).replace(".npz", "_f16.npz"))
    for k in ("embedding", "text_embedding"):
        v = torch.from_numpy(h[k])
        assert torch.equal(v.half().float(), v) and v.shape == g[k].shape
    assert (h["tokens"] == g["tokens"]).all() and int(h["seed"]) == int(g["seed"])
    d = relerr(h["embedding"], g["embedding"])
    assert abs(d - float(h["f16_vs_f32"])) < 1e-9 and 1e-4 < d < 3e-3
    dt = relerr(h["text_embedding"], g["text_embedding"])
    assert abs(dt - float(h["text_f16_vs_f32"])) < 1e-9 and 1e-4 < dt < 3e-3


def test_indices(golden):
    g = golden("indices.npz")
    _, _, grp = AO.group_index(g["raw_y"], g["raw_c"])
    assert grp.dtype == np.int64 and (grp == g["group"]).all()
    for bsr in (16, 100000):
        np.random.seed(42)
        idx, bs = AO.balance_val_indices(g["balance_garr"][200:900], 4, bsr)
        assert (idx == g[f"balance_idx_{bsr}"]).all() and bs == int(g[f"balance_bs_{bsr}"])
    t, s, p = (torch.from_numpy(g[k]) for k in ("minor_t", "minor_s", "minor_p"))
    for ds, pre in (("waterbirds", "wb"), ("celeba", "ca")):
        m, mp = CO.minority_flags(ds, t, s, p)
        assert (m.numpy() == g[pre + "_is_minor"]).all() and (mp.numpy() == g[pre + "_is_minor_pred"]).all()


def test_checkpoint_contract():
    keys = json.load(open(os.path.join(GOLDEN, "multiple_adapter_keys.json")))
    assert len(keys) == 18
    assert keys["new_adapter.layers.0.weight"][0] == [128, 1024]


def test_build_model_rounds_the_same_keys_to_fp16_as_the_reference():
    """fixture = which parameters the REFERENCE's build_model left fp16-rounded / untouched when
    fed non-representable weights (oracle/make_golden.py fp16keys); our loader must classify
    every key the same way (and then do the rounding: checked on the values)."""
    import json, os
    import torch
    from conftest import ROOT
    from dbmm_amd import synth
    from dbmm_amd.clip.model import _as_loaded
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "build_model_fp16_keys.json")))
    for arch, cls in gold.items():
        sd = {k: (v * 1.0001 + 1e-5 if v.is_floating_point() else v) for k, v in synth.clip_state_dict(2, arch).items()}
        assert set(cls) == {k for k, v in sd.items() if v.is_floating_point()}
        for k, c in cls.items():
            got = _as_loaded(k, sd[k])
            want = sd[k].half().float() if c == "fp16" else sd[k].float()
            assert torch.equal(got, want), (arch, k, c)
