"""The C ABI sees raw pointers, sizes and leading dimensions; the Python wrappers therefore refuse operands that do not hold what the
kernels will index, before anything is launched (CPU tensors are refused earlier still).  These checks run without a GPU: a shape error
must surface before the device is touched."""
import pytest
import torch

from dbmm_amd import ops


def test_wrappers_refuse_cpu_tensors_before_any_launch():
    with pytest.raises(Exception):
        ops.gemm(torch.zeros(4, 8), torch.zeros(4, 8))
    with pytest.raises(Exception):
        ops.attnpool(torch.zeros(1, 2, 2, 64), torch.zeros(5, 64), torch.zeros(64, 64), torch.zeros(64), torch.zeros(128, 64), torch.zeros(128),
                     torch.zeros(32, 64), torch.zeros(32), 1)


def test_sized_helper():
    ops._sized("bias", None, 5)
    ops._sized("bias", torch.zeros(5), 5)
    with pytest.raises(RuntimeError, match="bias"):
        ops._sized("bias", torch.zeros(4), 5)
