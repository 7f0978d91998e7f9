"""Generate tests/golden/*.npz by running the REFERENCE's own modules on synthetic inputs.

Runs only in the build container (needs /root/reference, read-only).  Nothing from the
reference is copied: its modules are imported in place, fed with
debiasing-multi-modal_amd/synth.py tensors, and only numeric outputs are saved.  The
same run asserts that oracle/*.py reproduces the reference (that is what pins the oracle).

Import recipe: clip/model.py by file path (clip/__init__ needs
torchvision/ftfy, absent here); final_main.py by path with sys.modules stubs for
torchvision / visualizer_supcon and .cuda() patched to identity.

    python oracle/make_golden.py            # writes tests/golden/
"""
import importlib.util
import json
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import dbmm_amd  # noqa: E402  (root shim -> debiasing-multi-modal_amd/)
from dbmm_amd import synth  # noqa: E402
import clip_oracle as CO  # noqa: E402
import adapter_oracle as AO  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def ref_model_module():
    return _load_by_path("ref_clip_model", os.path.join(REF, "clip", "model.py"))


def ref_final_main():
    for n in ("torchvision", "torchvision.transforms", "visualizer_supcon"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["visualizer_supcon"].skim_dataloader_by_group = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    return _load_by_path("ref_final_main", os.path.join(REF, "final_main.py"))


def ref_tokenizer():
    ftfy = types.ModuleType("ftfy"); ftfy.fix_text = lambda s: s   # identity on ASCII prompts
    sys.modules["ftfy"] = ftfy
    return _load_by_path("ref_simple_tokenizer", os.path.join(REF, "clip", "simple_tokenizer.py"))


def summary(t, nsample=256):
    """checksum triple that pins a large tensor without committing it."""
    f = t.detach().double().flatten()
    step = max(1, f.numel() // nsample)
    return np.array([f.sum().item(), f.abs().sum().item()]), f[::step][:nsample].float().numpy()


def relerr(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# ---------------------------------------------------------------------------------------

def gen_clip(M, only=None):
    cases = [("tiny-RN", 3, 2), ("tiny-RN-w32", 3, 2), ("RN50", 2, 2), ("tiny-ViT", 3, 2), ("ViT-B/32", 2, 2),
             ("ViT-L/14@336px", 2, 1)]          # full depth (24 layers, 577 tokens): BASELINE configs[4]'s tower
    if only:
        cases = [c for c in cases if c[0] in only]
    for arch, seed, B in cases:
        sd = synth.clip_state_dict(seed, arch)
        model = M.build_model({k: v.clone() for k, v in sd.items()}).float().eval()
        res = model.visual.input_resolution
        img = synth.images(seed + 100, B, res)
        with torch.no_grad():
            ref = model.encode_image(img)
            out = {"embedding": ref.numpy(), "seed": seed, "batch": B, "res": res}
            if arch.startswith(("RN", "tiny-RN")):
                mine, stages = CO.rn_encode_image(sd, img, return_stages=True)
                # reference stage tensors via forward hooks-free recomputation
                v = model.visual
                x = img.type(v.conv1.weight.dtype)
                x = v.relu1(v.bn1(v.conv1(x))); x = v.relu2(v.bn2(v.conv2(x))); x = v.relu3(v.bn3(v.conv3(x)))
                x = v.avgpool(x); rstages = {"stem": x}
                for li in (1, 2, 3, 4):
                    x = getattr(v, f"layer{li}")(x); rstages[f"layer{li}"] = x
                for k, t in rstages.items():
                    assert relerr(stages[k], t) < 5e-6, (arch, k, relerr(stages[k], t))
                    out[f"{k}_sums"], out[f"{k}_sample"] = summary(t.permute(0, 2, 3, 1))  # NHWC order
            else:
                mine = CO.vit_encode_image(sd, img)
            e = relerr(mine, ref)
            print(f"[clip] {arch:12s} encode_image oracle-vs-reference rel err {e:.2e}")
            assert e < 5e-6, (arch, e)
            # text tower
            vocab = sd["token_embedding.weight"].shape[0]
            n = 4
            toks = torch.zeros(n, 77, dtype=torch.int32)
            lens = [5, 9, 12, 77]
            for i, L in enumerate(lens):
                toks[i, 0] = vocab - 2
                toks[i, 1:L - 1] = synth.integers(seed + 7 + i, "tok", (L - 2,), vocab - 2).int()
                toks[i, L - 1] = vocab - 1
            tref = model.encode_text(toks)
            tmine = CO.encode_text(sd, toks)
            e = relerr(tmine, tref)
            print(f"[clip] {arch:12s} encode_text  oracle-vs-reference rel err {e:.2e}")
            assert e < 5e-6, (arch, e)
            out["tokens"] = toks.numpy(); out["text_embedding"] = tref.numpy()
            # zero-shot tail (clip_inference.py:207-216) on the reference embedding
            W = synth.text_matrix(seed + 1, ref.shape[1], 2, "zs")
            f = ref.clone(); f /= f.norm(dim=-1, keepdim=True)
            logits = f @ W / 0.02
            probs = logits.softmax(dim=-1)
            _, pred = torch.max(probs, dim=1)
            out["zs_logits"] = logits.numpy(); out["zs_pred"] = pred.numpy()
        np.savez_compressed(os.path.join(GOLD, "clip_" + arch.replace("/", "-").replace("@", "-") + ".npz"), **out)


def gen_clip_f16(M, only=None):
    """The reference's fp16 path (what `clip.load` leaves on a GPU: `build_model` WITHOUT `.float()`,
    clip/clip.py:139-141 -- conv / linear / attention weights fp16 via convert_weights, clip/model.py:375-396,433;
    activations fp16 because encode_image casts the image to `self.dtype`, :341; LayerNorm in fp32 and cast back,
    :157-163; BatchNorm parameters fp32) run HERE on the CPU, same seeds / inputs as the fp32 fixtures.
    Writes clip_<arch>_f16.npz: image + text embeddings (fp16 values stored as fp32), RN per-stage samples, and
    `f16_vs_f32` = the distance of the reference's own fp16 result from its fp32 result (relative to the embedding
    maximum) -- the size of one rounding history; the tests' tolerance is a small multiple of it."""
    cases = [("tiny-RN", 3, 2), ("tiny-RN-w32", 3, 2), ("RN50", 2, 2), ("tiny-ViT", 3, 2), ("ViT-B/32", 2, 2),
             ("ViT-L/14@336px", 2, 1)]
    if only:
        cases = [c for c in cases if c[0] in only]
    for arch, seed, B in cases:
        sd = synth.clip_state_dict(seed, arch)
        model = M.build_model({k: v.clone() for k, v in sd.items()}).eval()          # no .float(): the GPU-path dtypes
        assert model.dtype == torch.float16
        res = model.visual.input_resolution
        img = synth.images(seed + 100, B, res)
        g32 = np.load(os.path.join(GOLD, "clip_" + arch.replace("/", "-").replace("@", "-") + ".npz"))
        with torch.no_grad():
            ref = model.encode_image(img)
            assert ref.dtype == torch.float16
            out = {"embedding": ref.float().numpy(), "seed": seed, "batch": B, "res": res,
                   "f16_vs_f32": np.float64(relerr(ref.float(), torch.from_numpy(g32["embedding"])))}
            if arch.startswith(("RN", "tiny-RN")):
                v = model.visual
                x = img.type(v.conv1.weight.dtype)
                x = v.relu1(v.bn1(v.conv1(x))); x = v.relu2(v.bn2(v.conv2(x))); x = v.relu3(v.bn3(v.conv3(x)))
                x = v.avgpool(x); rstages = {"stem": x}
                for li in (1, 2, 3, 4):
                    x = getattr(v, f"layer{li}")(x); rstages[f"layer{li}"] = x
                for k, t in rstages.items():
                    out[f"{k}_sums"], out[f"{k}_sample"] = summary(t.float().permute(0, 2, 3, 1))  # NHWC order
            toks = torch.from_numpy(g32["tokens"])
            tref = model.encode_text(toks)
            out["tokens"] = toks.numpy(); out["text_embedding"] = tref.float().numpy()
            out["text_f16_vs_f32"] = np.float64(relerr(tref.float(), torch.from_numpy(g32["text_embedding"])))
        print(f"[clip f16] {arch:14s} image f16-vs-f32 {out['f16_vs_f32']:.2e}  text {out['text_f16_vs_f32']:.2e}")
        np.savez_compressed(os.path.join(GOLD, "clip_" + arch.replace("/", "-").replace("@", "-") + "_f16.npz"), **out)


def gen_tokens():
    T = ref_tokenizer()
    tok = T.SimpleTokenizer()
    prompts = []
    for mod in ("classic_celeba_templates", "classic_waterbirds_templates"):
        m = _load_by_path("ref_" + mod, os.path.join(REF, mod + ".py"))
        for kws in (m.classes, m.spurious_attributes, m.group_attributes):
            prompts += ["a photo of a {}.".format(k) for k in kws]
    prompts += ["A photo of a dog's  bowl -- it's 3.5 metres!", ""]
    sot, eot = tok.encoder["<|startoftext|>"], tok.encoder["<|endoftext|>"]
    rows = np.zeros((len(prompts), 77), dtype=np.int32)
    for i, p in enumerate(prompts):
        ids = [sot] + tok.encode(p) + [eot]
        rows[i, :len(ids)] = ids
    np.savez_compressed(os.path.join(GOLD, "tokens.npz"), tokens=rows)
    with open(os.path.join(GOLD, "tokens_prompts.json"), "w") as f:
        json.dump(prompts, f, indent=0)
    print(f"[tokens] {len(prompts)} prompts, sot={sot} eot={eot}")


def _write_text_json(path, mat, names):
    with open(path, "w") as f:
        json.dump({n: mat[:, i].numpy().tolist() for i, n in enumerate(names)}, f)


TIE_MARGIN = 1e-5      # |BatchNorm output| below this at step 0 = a ReLU input whose sign rounding decides


def gen_adapter(FM, D=1024, Bs=(4, 256, 1024), fname="adapter.npz"):
    """D = 1024 is the reference's hard-coded RN50 width (final_main.py:31,304); D = 512 / 768 are the
    ViT-B/32 and ViT-L/14 widths BASELINE configs[3] / [4] need (D becomes a parameter),
    run through the reference's own Adapter(D, 128) / CustomCLIP / MultipleAdapter classes.

    Stored WITH every case, all measured on the reference itself:
      * `ties`  [n, 2] int64 (row, hidden unit): BatchNorm outputs of the trainable adapter within TIE_MARGIN of 0 at
        step 0.  ReLU'(+-1e-7) is a tie; which side an implementation lands on decides that unit's rank-one term of the
        layer-0 / BatchNorm gradients, so tests leave exactly those units out and hold everything else tight.
      * `traj_tol` / `eval_tol`: the reference re-run from the same start on an input perturbed by ONE ulp -- largest
        relative parameter difference after the three steps (x 4) / largest absolute eval-logit difference (x 4).  At
        T = 0.01 some trajectories amplify 6e-8 to 1e-3; the stage-2 start seed is chosen (and stored, `new_seed`) so
        that the committed case is well conditioned where one of a few seeds is.
      * `stage1/<key>`: the complete stage-1 end state of the B = 256 run (or the smallest B >= 256): every stage-2 case
        of the file starts from it, so that step-0 logits of stage 2 are a pure forward pass on identical parameters."""
    from demo.util import set_optimizer, set_optimizer_reg  # reference's own helpers
    import copy, io, contextlib
    tmp = tempfile.mkdtemp(prefix="dbmm_golden_")
    H = 128
    tcls, tsp, tgrp = (synth.text_matrix(1, D, 2, "class"), synth.text_matrix(1, D, 2, "spurious"),
                       synth.text_matrix(1, D, 4, "group"))
    paths = [os.path.join(tmp, n) for n in ("clip_class.json", "clip_spurious.json", "clip_group.json")]
    _write_text_json(paths[0], tcls, ["c0", "c1"]); _write_text_json(paths[1], tsp, ["s0", "s1"])
    _write_text_json(paths[2], tgrp, ["g0", "g1", "g2", "g3"])
    crit = torch.nn.CrossEntropyLoss()
    opt_ns = SimpleNamespace(learning_rate=0.1, learning_rate_reg=0.05, momentum=0.9, weight_decay=5e-5)
    out = {}

    def record(tag, tensors):
        for k, t in tensors.items():
            if t.numel() <= 8192:
                out[f"{tag}/{k}"] = t.detach().numpy().copy()
            else:
                out[f"{tag}/{k}_sums"], out[f"{tag}/{k}_sample"] = summary(t)

    def three_steps(model, optim, bn, xin, labels, use_group, tag=None, osd=None, obufs=None, text=None, lr=None, multiple=False):
        """the reference's step body x 3 (final_main.py:455-466 / 610-623); with `tag`, records step 0 and checks the oracle"""
        model.train()
        seen = []
        hook = bn.register_forward_hook(lambda m, i, o: seen.append(o.detach().clone()))
        for step in range(3):
            logits = model(xin.detach(), use_group)
            loss = crit(logits, labels)
            optim.zero_grad(); loss.backward()
            if step == 0:
                hook.remove()
                if tag is not None:
                    out[f"{tag}/step0/ties"] = (seen[0].abs() < TIE_MARGIN).nonzero().numpy().astype(np.int64).reshape(-1, 2)
                    out[f"{tag}/step0/relu_margin"] = np.float64(seen[0].abs().min().item())
                    record(tag + "/step0", {"logits": logits, "loss": loss})
                    record(tag + "/step0/grad", {n: p.grad for n, p in model.named_parameters() if p.grad is not None})
            optim.step()
            if osd is not None:
                ol, ologits, _ = AO.train_step(osd, obufs, xin, labels, text, lr, multiple=multiple)
                assert relerr(ologits, logits.detach()) < 2e-5, tag
                if not multiple:
                    assert abs(ol.item() - loss.item()) < 1e-5 * max(1, abs(loss.item())), tag

    def sensitivity(make, xin, labels, use_group, base_model):
        """(4 x max relative parameter difference, 4 x max |eval logit| difference) of the reference re-run on x * (1 + 2^-23)"""
        m2, o2, bn2 = make()
        three_steps(m2, o2, bn2, xin * (1.0 + 2.0 ** -23), labels, use_group)
        sens = max(relerr(m2.state_dict()[k], v) for k, v in base_model.state_dict().items() if v.dtype.is_floating_point)
        m2.eval(); base_model.eval()
        with torch.no_grad():
            es = (m2(xin) - base_model(xin)).abs().max().item()
        return max(2e-5, 4 * sens), max(2e-4, 4 * es)

    stage1_B = min(b for b in Bs if b >= 256)
    stage1 = None
    inputs = {}
    for B in Bs:
        x = synth.normal(5, f"x{B}", (B, D), 0.5)
        y, c, g = synth.labels(6, B)
        inputs[B] = (x, y, g)
        # ---- stage 1: CustomCLIP, 3 SGD steps in train mode (class / group prompts) -------------
        for use_group in (False, True):
            tag = f"custom_B{B}_{'group' if use_group else 'class'}"

            def make():
                ad = FM.Adapter(D, H); ad.load_state_dict(synth.adapter_state_dict(3, D, H))
                clf = FM.CustomCLIP(ad, *paths, temperature=0.01)
                return clf, set_optimizer(opt_ns, clf), clf.adapter.layers[1]
            clf, optim, bn = make()
            osd = {"adapter." + k: v.clone() for k, v in synth.adapter_state_dict(3, D, H).items()}
            labels = g if use_group else y
            text = tgrp if use_group else tcls
            three_steps(clf, optim, bn, x, labels, use_group, tag, osd, {}, text, 0.1)
            record(tag + "/after3", dict(clf.state_dict()))
            for k, v in clf.state_dict().items():
                if v.dtype.is_floating_point:
                    assert relerr(osd[k], v) < 2e-5, (tag, k, relerr(osd[k], v))
            tt, et = sensitivity(make, x, labels, use_group, clf)
            out[f"{tag}/traj_tol"], out[f"{tag}/eval_tol"] = np.float64(tt), np.float64(et)
            # eval-mode logits (validate path, running stats)
            clf.eval()
            with torch.no_grad():
                ev = clf(x); evs = clf.forward_spurious(x)
                og = AO.custom_clip_logits(osd, x, tcls, 0.01, train=False)
                assert relerr(og, ev) < 2e-5, tag
            record(tag + "/eval", {"logits": ev, "logits_spurious": evs})
            if not use_group:
                out[f"{tag}/counts"] = AO.group_counts(ev, y, g)
                # update_dict / get_results of the reference on the same logits
                meters = {i: FM.AverageMeter() for i in range(4)}
                FM.update_dict(meters, y, g, ev)
                from functools import partial
                res = FM.get_results(meters, partial(FM.get_y_p, n_places=2))
                ometers = {i: AO.Meter() for i in range(4)}
                cnt = AO.group_counts(ev, y, g)
                for gi in range(4):
                    if cnt[gi, 0]:
                        ometers[gi].update(cnt[gi, 1] / cnt[gi, 0], int(cnt[gi, 0]))
                ores = AO.results_from_meters({k: m for k, m in ometers.items()})
                for k in res:
                    assert res[k] == ores[k], (k, res[k], ores[k])
                out[f"{tag}/results"] = np.array([res[k] for k in sorted(res)], dtype=np.float64)
            if B == stage1_B and use_group:
                stage1 = clf
                for k, v in clf.state_dict().items():
                    out[f"stage1/{k}"] = v.detach().numpy().copy()          # complete, bit for bit
    # ---- stage 2: MultipleAdapter on top of the trained stage-1 classifier (the stored one), every B ---------
    for B in Bs:
        x, y, g = inputs[B]
        for near_identity in (True, False):
            for use_group in (False, True):
                tag = f"multi_B{B}_{'ni' if near_identity else 'rn'}_{'group' if use_group else 'class'}"
                labels = g if use_group else y
                text = tgrp if use_group else tcls
                best = None
                for new_seed in (4, 14, 24, 34):
                    def make():
                        new_ad = FM.Adapter(D, H); new_ad.load_state_dict(synth.adapter_state_dict(new_seed, D, H))
                        with contextlib.redirect_stdout(io.StringIO()):
                            ma = FM.MultipleAdapter(copy.deepcopy(stage1), new_ad, init_near_identity=near_identity, ebd_weight=0.5)
                        return ma, set_optimizer_reg(opt_ns, ma), ma.new_adapter.layers[1]
                    ma, optim, bn = make()
                    three_steps(ma, optim, bn, x, labels, use_group)
                    tt, et = sensitivity(make, x, labels, use_group, ma)
                    if best is None or tt < best[1]:
                        best = (new_seed, tt, et)
                    if tt <= 1e-4:
                        break
                new_seed, tt, et = best
                ma, optim, bn = make()
                osd = {k: v.clone() for k, v in ma.state_dict().items()}
                three_steps(ma, optim, bn, x, labels, use_group, tag, osd, {}, text, 0.05, multiple=True)
                record(tag + "/after3", dict(ma.state_dict()))
                out[f"{tag}/new_seed"] = np.int64(new_seed)
                out[f"{tag}/traj_tol"], out[f"{tag}/eval_tol"] = np.float64(tt), np.float64(et)
                if tt > 1e-4:
                    print(f"[adapter] {tag}: ill-conditioned for every seed tried, best seed {new_seed}: traj_tol {tt:.2e} eval_tol {et:.2e}")
                for k, v in ma.state_dict().items():
                    if v.dtype.is_floating_point:
                        assert relerr(osd[k], v) < tt, (tag, k, relerr(osd[k], v))
                ma.eval()
                with torch.no_grad():
                    record(tag + "/eval", {"logits": ma(x), "logits_spurious": ma.forward_spurious(x)})
        print(f"[adapter] D={D} B={B} done")
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    return FM


def gen_indices(FM):
    out = {}
    # group = 2y + c with the -1 -> 0 CelebA recode (data/celeba_embeddings_reg.py:34-38)
    raw_y = synth.integers(11, "raw_y", (4096,), 2).numpy() * 2 - 1
    raw_c = synth.integers(12, "raw_c", (4096,), 2).numpy() * 2 - 1
    y = raw_y.copy(); c = raw_c.copy(); y[y == -1] = 0; c[c == -1] = 0
    out["raw_y"], out["raw_c"], out["group"] = raw_y, raw_c, (y * 2 + c).astype("int")
    oy, oc, og = AO.group_index(raw_y, raw_c)
    assert (og == out["group"]).all()
    # balance_val on a seeded synthetic group array, reference function on a fake loader
    garr = synth.integers(13, "garr", (1000,), 7).numpy()
    garr = np.minimum(garr, 3)                                    # skewed 4 groups
    ds = SimpleNamespace(n_groups=4, group_array=garr)
    sub = SimpleNamespace(dataset=ds, indices=np.arange(200, 900))
    loader = SimpleNamespace(dataset=sub)
    orig_subset, orig_loader = FM.Subset, FM.DataLoader
    FM.Subset = lambda d, idx: SimpleNamespace(d=d, idx=idx)
    FM.DataLoader = lambda s, shuffle, batch_size: SimpleNamespace(subset=s, batch_size=batch_size)
    for bsr in (16, 100000):
        np.random.seed(42)
        bl = FM.balance_val(loader, SimpleNamespace(batch_size_reg=bsr))
        np.random.seed(42)
        oidx, obs = AO.balance_val_indices(garr[200:900], 4, bsr)
        assert (oidx == bl.subset.idx).all() and obs == bl.batch_size
        out[f"balance_idx_{bsr}"] = np.asarray(bl.subset.idx, dtype=np.int64)
        out[f"balance_bs_{bsr}"] = np.int64(bl.batch_size)
    FM.Subset, FM.DataLoader = orig_subset, orig_loader
    out["balance_garr"] = garr
    # minority flags (clip_inference.py:219-233) are pure boolean algebra: pin truth tables
    t = torch.tensor([0, 0, 0, 0, 1, 1, 1, 1]); s = torch.tensor([0, 0, 1, 1, 0, 0, 1, 1]); p = torch.tensor([0, 1] * 4)
    out["minor_t"], out["minor_s"], out["minor_p"] = t.numpy(), s.numpy(), p.numpy()
    out["wb_is_minor_pred"] = (((t == 0) & (p == 1)) | ((t == 1) & (p == 0))).long().numpy()
    out["wb_is_minor"] = (((t == 0) & (s == 1)) | ((t == 1) & (s == 0))).long().numpy()
    out["ca_is_minor_pred"] = ((t == 1) & (p == 1)).long().numpy()
    out["ca_is_minor"] = ((t == 1) & (s == 1)).long().numpy()
    np.savez_compressed(os.path.join(GOLD, "indices.npz"), **out)
    print("[indices] done")


def gen_split():
    """stratified_split_dataset (data/celeba_embeddings_reg.py:95-107) of the reference on seeded
    synthetic group arrays (sklearn train_test_split, random_state=42)."""
    for n in ("torchvision", "torchvision.transforms"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    mod = _load_by_path("ref_celeba_embeddings_reg", os.path.join(REF, "data", "celeba_embeddings_reg.py"))
    out = {}
    for name, n, skew in (("a", 1000, 7), ("b", 19867, 11)):       # b: the size of CelebA's val split
        garr = np.minimum(synth.integers(21, "split_" + name, (n,), skew).numpy(), 3)
        for ts in (0.5, 0.25):
            reg, val = mod.stratified_split_dataset(SimpleNamespace(group_array=garr), test_size=ts)
            out[f"{name}_garr"] = garr
            out[f"{name}_reg_{ts}"] = np.asarray(reg.indices, dtype=np.int64)
            out[f"{name}_val_{ts}"] = np.asarray(val.indices, dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "split.npz"), **out)
    print("[split] done")


def gen_lr():
    """LR schedule helpers of the reference (demo/util.py:70-115) on a grid of epochs / batches."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from demo import util as U
    base = dict(learning_rate=0.1, learning_rate_reg=1.0, lr_decay_rate=0.1, lr_decay_epochs=[30, 60, 90], epochs=100,
                epochs_feature_learning=50, cosine=False, warm=True, warm_epochs=10, warmup_from=0.001, warmup_to=0.1,
                warm_reg=True, warm_epochs_reg=2, warmup_from_reg=0.01, warmup_to_reg=1.0)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.5)
    lr = lambda: opt.param_groups[0]["lr"]
    rows = []
    for cosine in (False, True):
        a = SimpleNamespace(**dict(base, cosine=cosine))
        for epoch in (0, 1, 2, 10, 30, 31, 60, 61, 90, 91, 100):
            U.adjust_learning_rate(a, opt, epoch)
            rows.append(["adjust_learning_rate", cosine, epoch, 0, 0, lr()])
            try:
                U.adjust_learning_rate_reg(a, opt, epoch)
                rows.append(["adjust_learning_rate_reg", cosine, epoch, 0, 0, lr()])
            except AttributeError as e:
                rows.append(["adjust_learning_rate_reg", cosine, epoch, 0, 0, "AttributeError"])
    a = SimpleNamespace(**base)
    for epoch in (1, 2, 3, 10, 11):
        for batch_id in (0, 1, 7, 18):
            opt.param_groups[0]["lr"] = 0.5
            U.warmup_learning_rate(a, epoch, batch_id, 19, opt)
            rows.append(["warmup_learning_rate", False, epoch, batch_id, 19, lr()])
            opt.param_groups[0]["lr"] = 0.5
            U.warmup_learning_rate_reg(a, epoch, batch_id, 19, opt)
            rows.append(["warmup_learning_rate_reg", False, epoch, batch_id, 19, lr()])
    with open(os.path.join(GOLD, "lr_schedule.json"), "w") as f:
        json.dump({"args": base, "rows": rows}, f, indent=0)
    print(f"[lr] {len(rows)} rows")


def gen_checkpoint_contract():
    """state-dict key/shape contract of the shipped MultipleAdapter checkpoint (weights_only load)."""
    d = os.path.join(REF, "trained_model")
    pth = [f for f in os.listdir(d) if f.endswith(".pth")][0]
    sd = torch.load(os.path.join(d, pth), map_location="cpu", weights_only=True)
    with open(os.path.join(GOLD, "multiple_adapter_keys.json"), "w") as f:
        json.dump({k: [list(v.shape), str(v.dtype)] for k, v in sd.items()}, f, indent=1)
    print(f"[ckpt] {len(sd)} tensors")


def gen_fp16_keys(M):
    """Which state-dict entries the reference's build_model rounds to fp16 on load
    (convert_weights before load_state_dict, clip/model.py:375-396,433): feed it weights that are
    NOT fp16-representable and record, per key, whether the loaded (then .float()-ed) parameter
    equals the fp16-rounded or the original value."""
    out = {}
    for arch in ("tiny-RN", "tiny-ViT"):
        sd = {k: (v * 1.0001 + 1e-5 if v.is_floating_point() else v) for k, v in synth.clip_state_dict(2, arch).items()}
        model = M.build_model({k: v.clone() for k, v in sd.items()}).float()
        got = model.state_dict()
        cls = {}
        for k, v in sd.items():
            if not v.is_floating_point():
                continue
            if torch.equal(got[k], v.half().float()) and not torch.equal(v.half().float(), v):
                cls[k] = "fp16"
            elif torch.equal(got[k], v):
                cls[k] = "fp32"
            else:
                raise SystemExit(f"{arch} {k}: neither fp16-rounded nor kept")
        out[arch] = cls
        print(f"[fp16 keys] {arch}: {sum(c == 'fp16' for c in cls.values())} rounded, "
              f"{sum(c == 'fp32' for c in cls.values())} kept")
    with open(os.path.join(GOLD, "build_model_fp16_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)



# ---------------------------------------------------------------------------------------
# two-stage training schedule: the reference's OWN driver, train_all_epochs (final_main.py:805-1128)
# ---------------------------------------------------------------------------------------

TWO_STAGE = dict(seed=21, n_train=2048, n_val=1600, n_test=1024, dim=1024,
                 argv=["--dataset", "celeba", "--tl_method", "adapter_reg_seq_alter", "--add_adapter", "--balance_val", "--continue_from_best",
                       "--warm_reg", "--epochs", "8", "--epochs_feature_learning", "3", "--batch_size", "256", "--batch_size_reg", "16",
                       "--learning_rate", "0.1", "--learning_rate_reg", "0.05", "--lr_decay_epochs", "6,7", "--lr_decay_rate", "0.5",
                       "--random_seed", "42"])


# a second schedule through the OTHER branches of train_all_epochs: no MultipleAdapter (stage 2 keeps training the stage-1 classifier
# with a fresh optimiser over all its parameters), no re-balancing (the reg loader itself, shuffle = True), no restart from the best
# model, group prompts in every stage-2 epoch (`adapter_reg_seq` without --use_cls_prompt_in_reg), step decay inside stage 2
TWO_STAGE_B = dict(seed=33, n_train=1536, n_val=1200, n_test=768, dim=1024,
                   argv=["--dataset", "celeba", "--tl_method", "adapter_reg_seq", "--warm_reg", "--epochs", "6", "--epochs_feature_learning", "2",
                         "--batch_size", "512", "--batch_size_reg", "64", "--learning_rate", "0.2", "--learning_rate_reg", "0.02",
                         "--lr_decay_epochs", "4,5", "--lr_decay_rate", "0.5", "--random_seed", "7"])


def _synthetic_embedding_module(FM, cfg, scale=1.0, log=None):
    """A stand-in for data/celeba_embeddings{,_reg}.py (which read the absent CSV / JSON files): the same dataset protocol
    (data/celeba_embeddings_reg.py:40-84: attributes and the __getitem__ tuple) over synth.embedding_dataset, and the same four loaders
    as load_celeba_embeddings (:109-131; num_workers = 0) built with the REFERENCE's stratified_split_dataset."""
    from torch.utils.data import DataLoader, Dataset
    real = _load_by_path("ref_celeba_embeddings_reg", os.path.join(REF, "data", "celeba_embeddings_reg.py"))
    sizes = {"train": cfg["n_train"], "val": cfg["n_val"], "test": cfg["n_test"]}

    class CelebaEmbeddings(Dataset):
        def __init__(self, data_dir=None, split="train", embedding_dir=None, transform=None):
            x, y, c = synth.embedding_dataset(cfg["seed"], split, sizes[split], cfg["dim"])
            self.split, self.x = split, x * scale
            self.y_array, self.confounder_array = y.numpy().copy(), c.numpy().copy()
            self.group_array = (self.y_array * 2 + self.confounder_array).astype("int")
            self.filename_array = np.array([f"{split}_{i:06d}.jpg" for i in range(len(y))])
            self.targets, self.targets_group = torch.tensor(self.y_array), torch.tensor(self.group_array)
            self.targets_spurious = torch.tensor(self.confounder_array)
            self.n_classes, self.n_groups, self.n_places = 2, 4, 2
            self.group_counts = (torch.arange(self.n_groups).unsqueeze(1) == torch.from_numpy(self.group_array)).sum(1).float()
            self.group_ratio = self.group_counts / len(self)

        def __len__(self):
            return len(self.filename_array)

        def __getitem__(self, idx):
            if log is not None:
                log.append((self.split, int(idx)))
            return self.x[idx], {"class": self.targets[idx], "group": self.targets_group[idx], "spurious": self.targets_spurious[idx],
                                 "ebd_y_pred": 0}, self.filename_array[idx]

    def load_celeba_embeddings(data_dir, embedding_dir, bs_train=512, bs_val=512, num_workers=0, transform=None):
        train_loader = DataLoader(CelebaEmbeddings(split="train"), batch_size=bs_train, shuffle=True)
        reg_set, val_set = real.stratified_split_dataset(CelebaEmbeddings(split="val"), test_size=0.5)
        return (train_loader, DataLoader(reg_set, batch_size=bs_val, shuffle=True), DataLoader(val_set, batch_size=bs_val, shuffle=False),
                DataLoader(CelebaEmbeddings(split="test"), batch_size=bs_val, shuffle=False))

    def load_celeba_embeddings_plain(data_dir, embedding_dir, bs_train=512, bs_val=512, num_workers=0, transform=None):
        """data/celeba_embeddings.py's loader (three loaders, no reg split): train_all_epochs builds these first and then replaces
        them (final_main.py:843-848)"""
        return (DataLoader(CelebaEmbeddings(split="train"), batch_size=bs_train, shuffle=True),
                DataLoader(CelebaEmbeddings(split="val"), batch_size=bs_val, shuffle=False),
                DataLoader(CelebaEmbeddings(split="test"), batch_size=bs_val, shuffle=False))

    mod, plain = types.ModuleType("synthetic_celeba_embeddings_reg"), types.ModuleType("synthetic_celeba_embeddings")
    mod.CelebaEmbeddings, mod.load_celeba_embeddings = CelebaEmbeddings, load_celeba_embeddings
    plain.CelebaEmbeddings, plain.load_celeba_embeddings = CelebaEmbeddings, load_celeba_embeddings_plain
    return mod, plain


def run_reference_two_stage(FM, cfg, paths, scale=1.0):
    """parse_option() + train_all_epochs() of the reference, unmodified, on the synthetic embedding set; its loop functions are wrapped
    (not replaced) to record what every epoch saw and produced"""
    log = []
    mod, plain = _synthetic_embedding_module(FM, cfg, scale, log)
    saved_mods = {k: sys.modules.get(k) for k in ("data.celeba_embeddings", "data.celeba_embeddings_reg")}
    sys.modules["data.celeba_embeddings"], sys.modules["data.celeba_embeddings_reg"] = plain, mod
    import data as _data_pkg                                                       # `from data.x import ...` resolves through sys.modules
    argv = sys.argv
    sys.argv = ["final_main.py"] + cfg["argv"] + ["--text_embedding_dir", paths[0], "--text_spurious_embedding_dir", paths[1],
                                                 "--text_group_embedding_dir", paths[2], "--image_embedding_dir", "/nonexistent/e.json",
                                                 "--data_dir", "/nonexistent"]
    rec = {"epochs": [], "inits": [], "lr": []}
    cur = {}
    names = ("train_one_epoch", "train_reg_seq_one_epoch", "validate", "validate_zs", "update_dict", "set_model", "set_model_multiple_adapter",
             "warmup_learning_rate", "warmup_learning_rate_reg", "balance_val")
    orig = {n: getattr(FM, n) for n in names}

    def phase(kind, fn):
        def wrapped(*a, **k):
            cur.clear(); cur.update(kind=kind, counts=np.zeros((4, 2), dtype=np.int64), start=len(log), lr=[])
            out = fn(*a, **k)
            loss, acc, gacc = out
            rec["epochs"].append(dict(kind=kind, label=k.get("print_label", ""), use_group=bool(k.get("use_group", False)),
                                      target=k.get("target"), loss=float(loss), acc=float(acc), counts=cur["counts"].copy(),
                                      group_acc={kk: float(v) for kk, v in gacc.items()}, idx=[i for _, i in log[cur["start"]:]],
                                      split=log[cur["start"]][0] if len(log) > cur["start"] else "", lr=list(cur["lr"])))
            return out
        return wrapped

    def update_dict(acc_groups, y, g, logits):
        cur["counts"] += AO.group_counts(logits.detach(), y, g)
        return orig["update_dict"](acc_groups, y, g, logits)

    def warm(fn):
        def wrapped(args, epoch, batch_id, total, optimizer):
            fn(args, epoch, batch_id, total, optimizer)
            cur["lr"].append(float(optimizer.param_groups[0]["lr"]))
        return wrapped

    def model_maker(fn, which):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            m = out[0]
            ad = m.new_adapter if which == "stage2" else m.adapter
            rec["inits"].append({kk: v.detach().clone().numpy() for kk, v in ad.state_dict().items()})
            return out
        return wrapped

    def balance(fn):
        def wrapped(loader, opt, print_procedure=False):
            out = fn(loader, opt, print_procedure)
            rec.setdefault("balanced", []).append((np.asarray(out.dataset.indices).copy(), int(out.batch_size)))
            return out
        return wrapped
    FM.train_one_epoch = phase("train1", orig["train_one_epoch"])
    FM.train_reg_seq_one_epoch = phase("train2", orig["train_reg_seq_one_epoch"])
    FM.validate = phase("validate", orig["validate"])
    FM.validate_zs = phase("validate_zs", orig["validate_zs"])
    FM.update_dict = update_dict
    FM.warmup_learning_rate, FM.warmup_learning_rate_reg = warm(orig["warmup_learning_rate"]), warm(orig["warmup_learning_rate_reg"])
    FM.set_model, FM.set_model_multiple_adapter = model_maker(orig["set_model"], "stage1"), model_maker(orig["set_model_multiple_adapter"], "stage2")
    FM.balance_val = balance(orig["balance_val"])
    try:
        opt = FM.parse_option()                                                   # set_seed(opt.random_seed) runs in here
        rec["opt"] = {k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, bool, list))}
        import contextlib, io
        # set_model_multiple_adapter only binds its return value under `if torch.cuda.is_available()` (final_main.py:338-343: the
        # reference is GPU-only); .cuda() is the identity here (ref_final_main), so answering True keeps everything on the CPU
        cuda_avail, torch.cuda.is_available = torch.cuda.is_available, (lambda: True)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                rec["final"] = FM.train_all_epochs(opt)
        finally:
            torch.cuda.is_available = cuda_avail
    finally:
        sys.argv = argv
        for n, f in orig.items():
            setattr(FM, n, f)
        for k, v in saved_mods.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return rec


def gen_two_stage(FM, cfg=None, fname="two_stage.npz"):
    """tests/golden/two_stage.npz: per-epoch (n, correct) counters, losses, accuracies, learning rates, batch index streams, the two
    adapter initialisations and the final / zero-shot results of the reference's train_all_epochs on the synthetic embedding set, and
    the same run on embeddings scaled by (1 + 2^-23) -- the reference's own sensitivity to one ulp of its input."""
    cfg = cfg or TWO_STAGE
    tmp = tempfile.mkdtemp()
    tcls, tspu, tgrp = synth.embedding_text(cfg["seed"], cfg["dim"])
    paths = [os.path.join(tmp, n) for n in ("clip_class.json", "clip_spurious.json", "clip_group.json")]
    _write_text_json(paths[0], tcls, ["c0", "c1"]); _write_text_json(paths[1], tspu, ["s0", "s1"])
    _write_text_json(paths[2], tgrp, ["g0", "g1", "g2", "g3"])
    rec = run_reference_two_stage(FM, cfg, paths)
    pert = run_reference_two_stage(FM, cfg, paths, scale=1.0 + 2.0 ** -23)
    pert8 = run_reference_two_stage(FM, cfg, paths, scale=1.0 + 2.0 ** -20)       # 8 ulp: the size of an fp32 kernel's rounding differences
    keys = ["weighted_mean_acc", "worst_acc", "acc_0_0", "acc_0_1", "acc_1_0", "acc_1_1", "mean_acc"]
    out = {"config": np.array(json.dumps({k: v for k, v in cfg.items()})), "opt": np.array(json.dumps(rec["opt"])),
           "n_phases": np.int64(len(rec["epochs"])), "acc_keys": np.array(keys)}
    for i, (e, pe) in enumerate(zip(rec["epochs"], pert["epochs"])):
        assert e["kind"] == pe["kind"] and e["idx"] == pe["idx"]
        out[f"p{i}/kind"] = np.array(e["kind"]); out[f"p{i}/label"] = np.array(e["label"]); out[f"p{i}/split"] = np.array(e["split"])
        out[f"p{i}/use_group"] = np.bool_(e["use_group"]); out[f"p{i}/target"] = np.array(str(e["target"]))
        out[f"p{i}/loss"] = np.float64(e["loss"]); out[f"p{i}/acc"] = np.float64(e["acc"])
        out[f"p{i}/counts"] = e["counts"]; out[f"p{i}/counts_1ulp"] = pe["counts"]; out[f"p{i}/loss_1ulp"] = np.float64(pe["loss"])
        out[f"p{i}/counts_8ulp"] = pert8["epochs"][i]["counts"]; out[f"p{i}/loss_8ulp"] = np.float64(pert8["epochs"][i]["loss"])
        out[f"p{i}/group_acc"] = np.array([e["group_acc"].get(k, np.nan) for k in keys], dtype=np.float64)
        out[f"p{i}/group_acc_1ulp"] = np.array([pe["group_acc"].get(k, np.nan) for k in keys], dtype=np.float64)
        out[f"p{i}/idx"] = np.asarray(e["idx"], dtype=np.int64); out[f"p{i}/lr"] = np.asarray(e["lr"], dtype=np.float64)
    for i, (bi, bs) in enumerate(rec.get("balanced", [])):
        out[f"balanced{i}/indices"], out[f"balanced{i}/batch_size"] = bi.astype(np.int64), np.int64(bs)
    for i, sd in enumerate(rec["inits"]):                                          # both sides draw them from the seeded global RNG:
        for k, v in sd.items():                                                    # checksums + 256 strided samples pin them
            out[f"init{i}/{k}_sums"], out[f"init{i}/{k}_sample"] = summary(torch.from_numpy(np.asarray(v)))
    (btr, bva, bte), (zs, zss) = rec["final"]
    out["final/best_test"] = np.array([bte[k] for k in keys]); out["final/best_val"] = np.array([bva[k] for k in keys])
    out["final/zs_class"] = np.array([zs[k] for k in keys]); out["final/zs_spurious"] = np.array([zss[k] for k in keys])
    (_, _, pte), _ = pert["final"]
    out["final/best_test_1ulp"] = np.array([pte[k] for k in keys])
    kinds = [e["kind"] for e in rec["epochs"]]
    print("[two_stage] phases:", " ".join(k[0] + k[-1] for k in kinds))
    for i, e in enumerate(rec["epochs"]):
        d = (np.abs(e["counts"] - pert["epochs"][i]["counts"]).max(), np.abs(e["counts"] - pert8["epochs"][i]["counts"]).max())
        print(f"[two_stage] p{i:02d} {e['kind']:11s} n={e['counts'][:, 0].sum():5d} loss {e['loss']:.4f} acc {e['acc']:.4f} worst "
              f"{e['group_acc'].get('worst_acc', float('nan')):.4f}  group correct {e['counts'][:, 1].tolist()}  |1 / 8 ulp count diff| {d}")
    np.savez_compressed(os.path.join(GOLD, fname), **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which_all = ["clip", "clip_vitl", "clip_f16", "tokens", "adapter", "adapter_vit", "indices", "ckpt", "fp16keys", "split", "lr", "two_stage"]
    which = [a for a in sys.argv[1:] if a in which_all] or ["clip", "clip_f16", "tokens", "adapter", "adapter_vit", "indices", "ckpt",
                                                              "fp16keys", "split", "lr"]
    if "fp16keys" in which:
        gen_fp16_keys(ref_model_module())
    if "clip" in which:
        gen_clip(ref_model_module())
    if "clip_vitl" in which:                      # only the full-depth ViT-L/14@336px case (minutes on CPU)
        gen_clip(ref_model_module(), only=("ViT-L/14@336px",))
    if "clip_f16" in which:                       # the reference's fp16 path on the CPU (needs the fp32 fixtures above)
        gen_clip_f16(ref_model_module(), only=[a for a in sys.argv[1:] if a not in which_all] or None)
    if "split" in which:
        gen_split()
    if "lr" in which:
        gen_lr()
    if "tokens" in which:
        gen_tokens()
    if "adapter" in which or "indices" in which or "adapter_vit" in which or "two_stage" in which:
        FM = ref_final_main()
        if "two_stage" in which:                  # the reference's own train_all_epochs on a synthetic embedding set
            gen_two_stage(FM)
            gen_two_stage(FM, TWO_STAGE_B, "two_stage_b.npz")
        if "adapter" in which:
            gen_adapter(FM)
        if "adapter_vit" in which:                # BASELINE configs[3] / [4] adapter widths and global batches
            gen_adapter(FM, D=512, Bs=(256, 4096), fname="adapter_D512.npz")
            gen_adapter(FM, D=768, Bs=(256, 8192), fname="adapter_D768.npz")
        if "indices" in which:
            gen_indices(FM)
    if "ckpt" in which:
        gen_checkpoint_contract()
