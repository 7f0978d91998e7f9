"""Two ranks on ONE MI355X (gloo rendezvous, embeddings staged through the host -- dp.all_gather_rows' rehearsal
path; RCCL needs one GPU per rank and is exercised by the driver's multi-GPU bench only): the REAL step --
model.encode_image -> gather of embeddings + packed (y, g) -> adapter.CustomCLIP.train_step (the fused C step) with
optim.SGD -- must leave bit-identical parameters, momentum buffers, BatchNorm statistics, group counters and loss on
both ranks, equal to a single process that runs the adapter step on the whole batch.
The single-process reference encodes the two shards separately as the ranks do: kernel tile schedules, and therefore
fp32 summation order, depend on the batch a launch sees (tests/test_gpu_clip.py::test_large_batch_property_rn50)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
ARCH, B, STEPS = "tiny-RN", 16, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(tmp):
    import json
    import sys
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dbmm_amd  # noqa: F401
    from types import SimpleNamespace
    from dbmm_amd import adapter, optim, synth
    from dbmm_amd.clip.model import build_model
    torch.cuda.set_device(0)
    model = build_model(synth.clip_state_dict(3, ARCH)).cuda()
    D, R = model.visual.output_dim, model.visual.input_resolution
    paths = []
    for nm, C in (("c", 2), ("s", 2), ("g", 4)):
        p = os.path.join(tmp, f"{nm}.json")
        if not os.path.exists(p):
            m = synth.text_matrix(1, D, C, nm)
            with open(p, "w") as f:
                json.dump({f"{nm}{i}": m[:, i].tolist() for i in range(C)}, f)
        paths.append(p)
    ad = adapter.Adapter(D, 32); ad.load_state_dict(synth.adapter_state_dict(3, D, 32))
    clf = adapter.CustomCLIP(ad, *paths, temperature=0.01).cuda().train()
    opt = optim.set_optimizer(SimpleNamespace(learning_rate=0.1, momentum=0.9, weight_decay=5e-5), clf)
    images = synth.images(7, B, R).cuda()
    y, c, g = synth.labels(6, B)
    return model, clf, opt, images, y.cuda(), g.cuda()


def _state_bytes(clf, opt, step, loss):
    parts = [v.detach().float().flatten() for v in clf.state_dict().values()]
    parts += [opt.state[p]["momentum_buffer"].flatten() for p in clf.parameters()]
    parts += [step.counts.float().flatten(), loss.detach().reshape(1)]
    return torch.cat([t.cpu() for t in parts]).numpy().tobytes()


def _worker(rank, world, port, tmp, q, micro=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dbmm_amd import dp
        model, clf, opt, images, y, g = _setup(tmp)
        lo, hi = dp.shard_rows(B, world, rank)
        step = dp.EmbedAdapterStep(model.encode_image, clf, opt, micro_batches=micro)
        for _ in range(STEPS):
            loss, logits, emb = step.step(images[lo:hi].contiguous(), y[lo:hi], g[lo:hi])
        torch.cuda.synchronize()
        q.put((rank, tuple(emb.shape), _state_bytes(clf, opt, step, loss)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("micro", [1, 2])
def test_two_ranks_one_gpu_equal_single_process(tmp_path, micro):
    """micro = 2: dp.EmbedAdapterStep(micro_batches=2), the gather-overlap path of BASELINE configs[3] (each rank encodes its
    shard in two chunks and gathers each as soon as it is encoded; under gloo the gathers run synchronously, the data flow
    and row order are the overlap path's)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q, micro)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][1][0] == B            # every rank holds the gathered global batch
    assert res[0][2] == res[1][2]                                   # replicated state is bit-identical
    # single process, whole batch (shards encoded separately like the ranks do), same fused step
    from dbmm_amd import dp
    model, clf, opt, images, y, g = _setup(str(tmp_path))
    n_chunks = world * micro                                         # the launches the ranks made, in global row order
    encode = lambda x: torch.cat([model.encode_image(c.contiguous()) for c in x.chunk(n_chunks)])
    step = dp.EmbedAdapterStep(encode, clf, opt)
    for _ in range(STEPS):
        loss, _, _ = step.step(images, y, g)
    torch.cuda.synchronize()
    assert _state_bytes(clf, opt, step, loss) == res[0][2]
    assert int(step.counts[:, 0].sum()) == B * STEPS


def _rccl_worker(port, tmp, q):
    """ONE rank on the `nccl` backend (= RCCL): a one-GPU box cannot host two RCCL ranks, but a world of one still loads RCCL, builds
    its communicator on this GPU and runs every collective as an RCCL kernel on RCCL's stream -- the calls bench.py and
    dp.EmbedAdapterStep make at N > 1, including the side-stream / async_op / record_stream ordering of the overlap path."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from dbmm_amd import dp
        out = {"backend": dist.get_backend()}
        for micro in (1, 2):
            model, clf, opt, images, y, g = _setup(tmp)
            step = dp.EmbedAdapterStep(model.encode_image, clf, opt, micro_batches=micro, always_collective=True)
            for _ in range(STEPS):
                loss, logits, emb = step.step(images, y, g)
            dist.barrier(device_ids=[0])
            torch.cuda.synchronize()
            out[micro] = _state_bytes(clf, opt, step, loss)
        t = torch.tensor([1.25], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # bench.py's max-over-ranks of the timed region
        who = [None]
        dist.all_gather_object(who, {"rank": 0, "device": torch.cuda.current_device()})
        # a gather much larger than a step's (64 MiB) racing a kernel that rewrites its source right after: stream ordering
        big = torch.arange(16 << 20, device=dev, dtype=torch.float32)
        got = dp.all_gather_rows(big.view(-1, 1024), always=True)
        big.zero_()
        torch.cuda.synchronize()
        out["allreduce"], out["who"] = t.item(), who
        out["big_ok"] = bool(torch.equal(got.flatten(), torch.arange(16 << 20, device=dev, dtype=torch.float32)))
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_runs_the_collectives_of_the_step_on_one_gpu(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), str(tmp_path), q))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out["backend"] == "nccl" and out["allreduce"] == 1.25 and out["who"] == [{"rank": 0, "device": 0}] and out["big_ok"]
    # the same steps without any process group, chunked like the micro-batched rank encodes
    from dbmm_amd import dp
    for micro in (1, 2):
        model, clf, opt, images, y, g = _setup(str(tmp_path))
        encode = lambda x: torch.cat([model.encode_image(c.contiguous()) for c in x.chunk(micro)])
        step = dp.EmbedAdapterStep(encode, clf, opt)
        for _ in range(STEPS):
            loss, _, _ = step.step(images, y, g)
        torch.cuda.synchronize()
        assert _state_bytes(clf, opt, step, loss) == out[micro], micro


def test_bench_starts_its_own_ranks(tmp_path):
    """the driver's command shape, `python bench.py --gpus 2 ...` with no launcher environment: the parent starts two ranks (gloo
    rehearsal: both on this GPU), rank 0's line carries the dist block and the fixed-global-batch leg, return code 0"""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DBMM_DIST_BACKEND="gloo", DBMM_BENCH_PROFILE="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["dist"]["world_size"] == 2 and line["dist"]["backend"] == "gloo"
    assert line["config"]["global_batch"] == 2048 and line["fixed_global_batch"]["config"]["global_batch"] == 1024
    assert line["value"] > 0 and line["fixed_global_batch"]["value"] > 0
