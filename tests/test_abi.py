"""The C-ABI library loads and exports every symbol include/dbmm.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from dbmm_amd import _lib


@pytest.fixture(scope="module")
def built():
    return _lib.build()


def _declared():
    txt = open(os.path.join(ROOT, "include", "dbmm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dbmm_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(built):
    L = ctypes.CDLL(built)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dbmm.h but not exported"


def test_binding_table_matches_header(built):
    assert sorted(_lib.EXPORTS) == _declared()


def test_version_and_error_strings(built):
    L = _lib.lib()
    assert L.dbmm_version() >= 100
    assert L.dbmm_error_string(0) == b"ok"
    assert b"align" in L.dbmm_error_string(-2)


def test_argument_validation_without_gpu(built):
    """shape / null checks return negative codes before anything touches a device."""
    L = _lib.lib()
    assert L.dbmm_gemm_bias_act(None, 4, 0, None, 4, 0, None, None, 0, None, 4, 4, 4, 4, 1.0, 0, None) == -4
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    # K not a multiple of 4 -> shape error (or alignment if the ctypes buffer is not 16-B aligned)
    assert L.dbmm_gemm_bias_act(p, 4, 0, p, 4, 0, None, None, 0, p, 4, 4, 4, 3, 1.0, 0, None) in (-1, -2)
    assert L.dbmm_avgpool2d(p, p, 1, 3, 3, 4, 2, None) in (-1, -2)
    assert L.dbmm_mha_core(p, p, 1, 4, 100, 2, 0, None) == -1          # E != heads*64
    assert L.dbmm_l2norm_sim_ce_fwd(p, None, 0.5, p, None, 0.01, p, None, None, None, None, 4, 8, 9, None) == -1
    assert L.dbmm_sgd_momentum(0, None, None, None, None, 0.1, 0.9, 0.0, 1, None) == -4
