"""`python bench.py --gpus N` starts its own N ranks (the driver's command shape): the parent launches torch.distributed.run as a
child before anything touches the GPU, forwards the ranks' output and returns their exit code."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


@pytest.fixture
def bench(monkeypatch):
    import bench as B
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    return B


def test_no_launch_for_one_gpu_or_inside_a_launcher(bench, monkeypatch):
    assert bench.self_launch(["--steps", "2"]) is None
    assert bench.self_launch(["--gpus", "1"]) is None
    monkeypatch.setenv("WORLD_SIZE", "2")                    # already a rank of torch.distributed.run
    assert bench.self_launch(["--gpus", "2"]) is None


def test_launch_command(bench, monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setenv("DBMM_DIST_BACKEND", "gloo")
    rc = bench.self_launch(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert rc == 7                                           # the children's return code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_rccl_needs_one_gpu_per_rank(bench, monkeypatch, capsys):
    import torch
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: pytest.fail("must not launch"))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("DBMM_DIST_BACKEND", raising=False)
    assert bench.self_launch(["--gpus", "2"]) == 2
    assert "needs 2 GPUs" in capsys.readouterr().err


def test_self_launched_ranks_meet_and_print_one_line():
    """end to end on the CPU: `python bench.py --gpus 2` with no launcher environment -> two ranks under torch.distributed.run on
    127.0.0.1 (gloo), rank 0's JSON line on the parent's stdout, return code 0"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DBMM_DIST_BACKEND="gloo", DBMM_BENCH_DRYRUN="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert lines[0]["n_gpus"] == 2 and lines[0]["max_over_ranks"] == 2.0
    assert sorted(r["rank"] for r in lines[0]["ranks"]) == [0, 1]


def test_failing_rank_fails_the_parent():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DBMM_DIST_BACKEND="gloo", DBMM_BENCH_DRYRUN="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-such-flag"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
