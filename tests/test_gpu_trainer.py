"""Fused step body, row gather and the device-resident epoch loop against the autograd path
and the oracle."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import adapter_oracle as AO
from conftest import relerr
from dbmm_amd import adapter, ops, optim, synth, trainer

pytestmark = pytest.mark.gpu
D, H = 1024, 128


@pytest.fixture(scope="module")
def text_paths(tmp_path_factory):
    d = tmp_path_factory.mktemp("text")
    paths = []
    for name, C, tag in (("clip_class", 2, "class"), ("clip_spurious", 2, "spurious"), ("clip_group", 4, "group")):
        m = synth.text_matrix(1, D, C, tag)
        p = os.path.join(d, name + ".json")
        json.dump({f"{tag}{i}": m[:, i].numpy().tolist() for i in range(C)}, open(p, "w"))
        paths.append(p)
    return paths


def _ns():
    from types import SimpleNamespace
    return SimpleNamespace(learning_rate=0.1, learning_rate_reg=0.05, momentum=0.9, weight_decay=5e-5)


def test_gather_rows():
    t = synth.normal(1, "t", (100, 64)).cuda()
    idx = torch.tensor([5, 99, 0, 5, 42], dtype=torch.int64).cuda()
    assert torch.equal(ops.gather_rows(t, idx), t[idx])


@pytest.mark.parametrize("B", [4, 256])
@pytest.mark.parametrize("multiple,use_group", [(False, False), (False, True), (True, False), (True, True)])
def test_train_step_equals_autograd_path(B, multiple, use_group, text_paths):
    x = synth.normal(5, f"x{B}", (B, D), 0.5).cuda()
    y, c, g = (t.cuda() for t in synth.labels(6, B))
    labels = g if use_group else y

    def make():
        ad = adapter.Adapter(D, H); ad.load_state_dict(synth.adapter_state_dict(3, D, H))
        clf = adapter.CustomCLIP(ad, *text_paths)
        if multiple:
            new = adapter.Adapter(D, H); new.load_state_dict(synth.adapter_state_dict(4, D, H))
            clf = adapter.MultipleAdapter(clf, new, init_near_identity=False)
            return clf.cuda().train(), optim.set_optimizer_reg(_ns(), clf)
        return clf.cuda().train(), optim.set_optimizer(_ns(), clf)
    a, oa = make(); b, ob = make()
    for step in range(3):
        la, logits_a, _ = a.loss(x, labels, use_group)
        oa.zero_grad(); la.backward(); oa.step()
        lb, logits_b, rows_b = b.train_step(x, labels, ob, use_group)
        assert torch.equal(logits_a, logits_b) and torch.equal(la.detach(), lb)     # same kernels, same order
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k
    # the two entry points can be mixed on one optimiser
    la, _, _ = b.loss(x, labels, use_group); ob.zero_grad(); la.backward(); ob.step()
    la2, _, _ = a.train_step(x, labels, oa, use_group)
    assert torch.equal(la.detach(), la2)


def test_train_epoch_matches_oracle_loop(text_paths):
    """device-resident epoch (DataLoader-order batches, fused steps, device counters) against the
    oracle driven with the same index stream."""
    N, bs = 600, 128
    emb = synth.normal(8, "table", (N, D), 0.5)
    y, c, g = synth.labels(9, N)
    emb = emb + 0.3 * synth.normal(10, "sig", (1, D)) * (2 * y.float().unsqueeze(1) - 1)
    table = trainer.EmbeddingTable(emb.numpy(), y.numpy(), c.numpy(), device="cuda")
    assert np.array_equal(table.group_array, g.numpy())
    ad = adapter.Adapter(D, H); ad.load_state_dict(synth.adapter_state_dict(3, D, H))
    clf = adapter.CustomCLIP(ad, *text_paths).cuda()
    opt = optim.set_optimizer(_ns(), clf)
    tcls = synth.text_matrix(1, D, 2, "class")
    osd = {"adapter." + k: v.clone() for k, v in synth.adapter_state_dict(3, D, H).items()}
    obufs = {}
    for epoch in range(2):
        torch.manual_seed(100 + epoch)
        loss, acc, gacc = trainer.train_epoch(table, clf, opt, bs)
        torch.manual_seed(100 + epoch)
        order = trainer.dataloader_shuffle_order(N)
        ol, correct = 0.0, 0
        cnt = np.zeros((4, 2), dtype=np.int64)
        for i in range(0, N, bs):
            idx = order[i:i + bs]
            l, logits, _ = AO.train_step(osd, obufs, emb[idx], y[idx], tcls, 0.1)
            ol += l.item() * len(idx)
            cnt += AO.group_counts(logits, y[idx], g[idx])
        assert abs(loss - ol / N) < 2e-3 * max(1.0, abs(ol / N))
        assert acc == cnt[:, 1].sum() / cnt[:, 0].sum()
        for gi in range(4):
            assert gacc[f"acc_{gi // 2}_{gi % 2}"] == np.round(cnt[gi, 1] / cnt[gi, 0], 4)
    vloss, vacc, vg = trainer.validate(table, clf, 256, table.group_ratio.numpy())
    oev = AO.custom_clip_logits(osd, emb, tcls, 0.01, train=False)
    ocnt = AO.group_counts(oev, y, g)
    assert vacc == ocnt[:, 1].sum() / ocnt[:, 0].sum()
    assert abs(vloss - torch.nn.functional.cross_entropy(oev, y).item()) < 2e-3 * max(1.0, vloss)
    assert set(vg) == set(trainer.NEW_ORDER_FOR_PRINT)


def test_embed_adapter_step_fused_equals_autograd(text_paths):
    """dp.EmbedAdapterStep: the one-call step body and the autograd path leave identical
    parameters, BN statistics, momentum and group counters"""
    from dbmm_amd import dp
    B = 64
    emb = synth.normal(7, "e", (B, D), 0.5).cuda()
    y, c, g = (t.cuda() for t in synth.labels(8, B))

    def run(fused):
        ad = adapter.Adapter(D, H); ad.load_state_dict(synth.adapter_state_dict(3, D, H))
        clf = adapter.CustomCLIP(ad, *text_paths).cuda().train()
        opt = optim.set_optimizer(_ns(), clf)
        st = dp.EmbedAdapterStep(lambda t: t, clf, opt, fused=fused)
        for _ in range(3):
            loss, logits, _ = st.step(emb, y, g)
        return clf.state_dict(), loss, logits, st.counts
    sa, la, ga, ca = run(True); sb, lb, gb, cb = run(False)
    assert torch.equal(la.detach(), lb.detach()) and torch.equal(ga, gb) and torch.equal(ca, cb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_deepcopy_and_pickle_after_fused_step(text_paths, tmp_path):
    """final_main.py:943, 1006-1008 deep-copies the classifier every epoch and starts stage 2 from the copy; the fused step's cached
    argument block (raw device addresses) must not travel with the module: the copy builds its own and matches the autograd path."""
    B = 64
    x = synth.normal(5, f"x{B}", (B, D), 0.5).cuda()
    y, c, g = (t.cuda() for t in synth.labels(6, B))
    ad = adapter.Adapter(D, H); ad.load_state_dict(synth.adapter_state_dict(3, D, H))
    clf = adapter.CustomCLIP(ad, *text_paths).cuda().train()
    opt = optim.set_optimizer(_ns(), clf)
    for _ in range(3):
        clf.train_step(x, y, opt)
    best = copy.deepcopy(clf)                                        # raised ValueError (ctypes pointers) in round 3
    assert "_step_plan" not in best.__dict__ and clf.__dict__["_step_plan"] is not None
    for (k, va), (_, vb) in zip(clf.state_dict().items(), best.state_dict().items()):
        assert torch.equal(va, vb) and (va.data_ptr() != vb.data_ptr() or va.numel() == 0), k
    torch.save(best, tmp_path / "clf.pt")                            # pickles the module (same rule)
    # stage 2 from the copy, fused vs autograd (both start from identical copies)
    def stage2(src):
        new = adapter.Adapter(D, H); new.load_state_dict(synth.adapter_state_dict(4, D, H))
        ma = adapter.MultipleAdapter(copy.deepcopy(src), new, init_near_identity=False).cuda().train()
        return ma, optim.set_optimizer_reg(_ns(), ma)
    a, oa = stage2(best); b, ob = stage2(best)
    for step in range(3):
        la, logits_a, _ = a.loss(x, g, True); oa.zero_grad(); la.backward(); oa.step()
        lb, logits_b, _ = b.train_step(x, g, ob, True)
        assert torch.equal(logits_a, logits_b) and torch.equal(la.detach(), lb)
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k
    # the copy of a stepped MultipleAdapter steps on ITS tensors: the original must not move
    before = {k: v.clone() for k, v in b.state_dict().items()}
    b2 = copy.deepcopy(b); ob2 = optim.set_optimizer_reg(_ns(), b2)
    b2.train_step(x, g, ob2, True)
    for k, v in b.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert b2.__dict__["_step_plan"]["args"] != b.__dict__["_step_plan"]["args"]


@pytest.mark.parametrize("what", ["w2_data", "running_mean", "momentum", "old_reload"])
def test_fused_step_follows_replaced_storage(what, text_paths):
    """every pointer of the cached argument block is re-checked: replacing a parameter's storage, a BatchNorm buffer, a momentum
    buffer, or reloading the frozen old adapter between steps must not leave the kernel writing through stale addresses"""
    B = 32
    x = synth.normal(5, f"x{B}", (B, D), 0.5).cuda()
    y, c, g = (t.cuda() for t in synth.labels(6, B))

    def make():
        ad = adapter.Adapter(D, H); ad.load_state_dict(synth.adapter_state_dict(3, D, H))
        old = adapter.CustomCLIP(ad, *text_paths)
        new = adapter.Adapter(D, H); new.load_state_dict(synth.adapter_state_dict(4, D, H))
        ma = adapter.MultipleAdapter(old, new, init_near_identity=False).cuda().train()
        return ma, optim.set_optimizer_reg(_ns(), ma)

    def disturb(m, o):
        keep = []
        if what == "w2_data":
            p = m.new_adapter.layers[3].weight
            keep.append(p.data); p.data = p.data.clone()
        elif what == "running_mean":
            bn = m.new_adapter.layers[1]
            keep.append(bn.running_mean); bn.running_mean = bn.running_mean.clone()
        elif what == "momentum":
            for p in o.param_groups[0]["params"]:
                keep.append(o.state[p]["momentum_buffer"]); o.state[p]["momentum_buffer"] = o.state[p]["momentum_buffer"].clone()
        else:
            sd = {k: v.clone() for k, v in m.old_cls.adapter.state_dict().items()}
            keep += [p.data for p in m.old_cls.adapter.parameters()]
            for p in m.old_cls.adapter.parameters():
                p.data = p.data.clone()
            m.old_cls.adapter.load_state_dict(sd)
        return keep                                                  # old storages stay alive: a stale write would go unnoticed otherwise

    a, oa = make(); b, ob = make()
    for step in range(4):
        if step == 2:
            ka, kb = disturb(a, oa), disturb(b, ob)
            stale = [t.clone() for t in kb]
        la, logits_a, _ = a.loss(x, g, True); oa.zero_grad(); la.backward(); oa.step()
        lb, logits_b, _ = b.train_step(x, g, ob, True)
        assert torch.equal(logits_a, logits_b) and torch.equal(la.detach(), lb), step
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k
    for pa, pb in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
        assert torch.equal(oa.state[pa]["momentum_buffer"], ob.state[pb]["momentum_buffer"])
    for t, s in zip(kb, stale):
        assert torch.equal(t, s)                                     # nothing wrote through the replaced storages
