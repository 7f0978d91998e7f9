import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import dbmm_amd  # noqa: E402,F401  (root shim -> debiasing-multi-modal_amd/)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def relerr(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def summary(t, nsample=256):
    f = t.detach().double().flatten().cpu()
    step = max(1, f.numel() // nsample)
    return np.array([f.sum().item(), f.abs().sum().item()]), f[::step][:nsample].float().numpy()


@pytest.fixture
def option():
    """set a library option (dbmm_set_option) for the duration of a test: option("igemm_halo", 0)"""
    from dbmm_amd import ops
    saved = {}

    def set_(name, value):
        old = ops.set_option(name, int(value))
        saved.setdefault(name, old)
    yield set_
    for name, old in saved.items():
        ops.set_option(name, old)
