"""N>1 path on CPU: world_size-2 gloo.  Checks the sharding + all-gather plumbing of
dbmm_amd.dp (row order, label packing, replicated state) with a stand-in encoder/classifier
(the HIP kernels need a GPU; the collective logic does not)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _ToyClassifier(torch.nn.Module):
    """offers the .loss(features, labels, use_group) protocol with plain torch ops."""
    def __init__(self, d):
        super().__init__()
        torch.manual_seed(0)
        self.lin = torch.nn.Linear(d, 4)
        self.bn = torch.nn.BatchNorm1d(4)          # batch statistics: global-batch dependent

    def loss(self, f, labels, use_group=False):
        logits = self.bn(self.lin(f))
        rows = torch.nn.functional.cross_entropy(logits, labels, reduction="none")
        return rows.mean(), logits.detach(), rows.detach()


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import dbmm_amd  # noqa: F401
    from dbmm_amd import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, D = 16, 8
        torch.manual_seed(1)
        images = torch.randn(B, D)                 # "images" already flattened for the toy encoder
        y = torch.randint(0, 2, (B,)); g = 2 * y + torch.randint(0, 2, (B,))
        lo, hi = dp.shard_rows(B, world, rank)
        enc = lambda x: x * 2.0 + 1.0
        clf = _ToyClassifier(D)
        opt = torch.optim.SGD(clf.parameters(), lr=0.1, momentum=0.9)
        step = dp.EmbedAdapterStep(enc, clf, opt)
        step.counts = torch.zeros(4, 2, dtype=torch.int64)
        emb, yy, gg = step.gather(enc(images[lo:hi]), y[lo:hi], g[lo:hi])
        ok = torch.equal(emb, enc(images)) and torch.equal(yy, y) and torch.equal(gg, g)
        # two replicated steps (group counting needs the HIP kernel -> exercise use_group path)
        for _ in range(2):
            loss, logits, _ = step.step(images[lo:hi], y[lo:hi], g[lo:hi], use_group=True)
        # micro-batched gather (the overlap path's data flow; gloo runs it synchronously): same rows in the same order
        mstep = dp.EmbedAdapterStep(enc, clf, opt, micro_batches=4)
        emb4, y4, g4 = mstep.encode_gather_overlapped(images[lo:hi], y[lo:hi], g[lo:hi])
        ok = ok and torch.equal(emb4, enc(images)) and torch.equal(y4, y) and torch.equal(g4, g)
        flat = torch.cat([p.detach().flatten() for p in clf.parameters()] + [clf.bn.running_mean, clf.bn.running_var])
        q.put((rank, ok, loss.item(), flat.numpy().tobytes()))   # bytes, not a tensor: shared-memory
        # tensor handles die with the worker and race the parent's q.get
    finally:
        dist.destroy_process_group()


def test_gather_order_and_replicated_step():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    # every rank holds bit-identical parameters / BN stats / loss (no gradient all-reduce needed)
    assert res[0][2] == res[1][2]
    assert res[0][3] == res[1][3]
    # ... and they equal a single-process run over the whole batch
    import dbmm_amd  # noqa: F401
    torch.manual_seed(1)
    images = torch.randn(16, 8); y = torch.randint(0, 2, (16,)); g = 2 * y + torch.randint(0, 2, (16,))
    clf = _ToyClassifier(8); opt = torch.optim.SGD(clf.parameters(), lr=0.1, momentum=0.9)
    for _ in range(2):
        loss, _, _ = clf.loss(images * 2.0 + 1.0, g)
        opt.zero_grad(); loss.backward(); opt.step()
    flat = torch.cat([p.detach().flatten() for p in clf.parameters()] + [clf.bn.running_mean, clf.bn.running_var])
    assert flat.numpy().tobytes() == res[0][3] and loss.item() == res[0][2]


def test_shard_rows():
    import dbmm_amd  # noqa: F401
    from dbmm_amd import dp
    assert [dp.shard_rows(1024, 4, r) for r in range(4)] == [(0, 256), (256, 512), (512, 768), (768, 1024)]
    with pytest.raises(ValueError):
        dp.shard_rows(10, 4, 0)
