"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/tfasr_hip.h declares."""
import os
import re

from tensorflowasr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "tfasr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tfasr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared_symbols()
    assert "tfasr_rnnt_loss" in names and "tfasr_gemm" in names
    for n in names:
        assert hasattr(lib, n), f"libtfasr_hip.so does not export {n}"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in tensorflowasr_amd/_lib.py"
    assert lib.tfasr_abi_version() == _lib.ABI_VERSION
    assert lib.tfasr_status_string(0) == b"success"


def test_workspace_query_needs_no_gpu():
    import ctypes

    n = ctypes.c_size_t(0)
    assert _lib.load().tfasr_rnnt_loss_workspace_size(2, 5, 4, 8, ctypes.byref(n)) == 0
    assert n.value >= 5 * 2 * 5 * 4 * 4
    assert _lib.load().tfasr_rnnt_loss_workspace_size(0, 5, 4, 8, ctypes.byref(n)) == 1


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch

    from tensorflowasr_amd import kernels

    x = torch.zeros(1, 2, 2, 4)
    with pytest.raises(Exception):
        kernels.rnnt_loss_fwd_bwd(x, torch.zeros(1, 1, dtype=torch.int32), torch.ones(1, dtype=torch.int32),
                                  torch.ones(1, dtype=torch.int32))
