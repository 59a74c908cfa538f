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


def test_block_executor_sizing_is_host_only():
    """tfasr_block_workspace_sizes is a dry run of the executor's allocation sequence: callable without a GPU."""
    import ctypes

    from tensorflowasr_amd import _lib

    lib = _lib.load()
    assert lib.tfasr_block_ctx_bytes() > 0
    k = _lib.BlockCfg()
    k.B, k.T, k.d, k.H, k.dh, k.dff, k.ksize = 4, 100, 256, 4, 64, 1024, 31
    k.dtype, k.training, k.save, k.use_mask, k.world, k.drop_p = _lib.TFASR_BF16, 1, 1, 1, 1, 0.1
    a, b, c = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.tfasr_block_workspace_sizes(ctypes.byref(k), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    rows = 400
    assert a.value > rows * (2 * 2 * 1024 + 768 + 512) * 2  # at least z, h of both FFNs + qkv + pw1 output
    assert c.value > rows * 1024 * 2
    k.force_unfused = 1
    a2 = ctypes.c_size_t(0)
    assert lib.tfasr_block_workspace_sizes(ctypes.byref(k), ctypes.byref(a2), ctypes.byref(b), ctypes.byref(c)) == 0
    assert a2.value > a.value  # the unfused path stores the probabilities
    assert lib.tfasr_block_workspace_sizes(None, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) != 0


def test_package_import_asks_for_enough_hardware_queues():
    """tensorflowasr_amd sets GPU_MAX_HW_QUEUES=8 at import unless the user chose a value (with HIP's default of 4 the prediction
    network's stream shares a hardware queue with the main stream once RCCL's streams exist: +3.3 ms per data-parallel step)."""
    import subprocess
    import sys

    code = "import os; import tensorflowasr_amd; print(os.environ.get('GPU_MAX_HW_QUEUES'))"
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root)
    assert out.stdout.strip() == "8", out.stderr[-500:]
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, GPU_MAX_HW_QUEUES="2"), cwd=root)
    assert out.stdout.strip() == "2"


def test_round4_entry_points_validate_their_arguments_on_the_host():
    """The entry points that finish deferred per-block work for a whole step refuse bad arguments before anything is launched."""
    import ctypes

    lib = _lib.load()
    INVALID = 1
    assert lib.tfasr_block_ln_fold_all(None, 4, 256, None) == INVALID
    arr = (ctypes.c_void_p * 2)(None, None)
    assert lib.tfasr_block_ln_fold_all(arr, 2, 256, None) == INVALID  # a NULL ctx
    assert lib.tfasr_block_ln_fold_all(arr, 0, 256, None) == INVALID
    assert lib.tfasr_layernorm_bwd_fold_sets(arr, 129, 8, 256, arr, arr, None) == INVALID  # more sets than the kernel's table holds
    assert lib.tfasr_layernorm_bwd_fold_sets(arr, 2, 8, 256, arr, arr, None) == INVALID    # NULL partial-sum pointers
    assert lib.tfasr_cast_colsum_many(None, None, 1024, 2, 8, 64, arr, None) == INVALID
    k = _lib.BlockCfg()
    assert lib.tfasr_block_dwconv_wgrad_all(ctypes.byref(k), arr, arr, arr, 2, None, 0, None) == INVALID  # NULL params / ctx entries
    assert lib.tfasr_block_dwconv_wgrad_all(ctypes.byref(k), arr, arr, arr, 33, None, 0, None) == INVALID
    # a block io without the options = a zeroed tail: the structure the executor reads must match the header's
    io = _lib.BlockIO()
    assert not io.pext_pre and io.defer_pos_grad == 0 and not io.ln_part_ext and not io.dcv_keep and not io.ds_keep and not io.qv_keep


def test_ctypes_structures_have_the_header_sizes(tmp_path):
    """The structures passed by pointer through the C ABI are declared twice (include/tfasr_hip.h and tensorflowasr_amd/_lib.py): their
    sizes must agree (a field added on one side only shifts everything behind it)."""
    import ctypes
    import shutil
    import subprocess

    import pytest

    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "tfasr_hip.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", sizeof(tfasr_block_io), '
                   'sizeof(tfasr_block_cfg), sizeof(tfasr_block_params), sizeof(tfasr_gemm_args)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(_lib.BlockIO), ctypes.sizeof(_lib.BlockCfg), ctypes.sizeof(_lib.BlockParams), ctypes.sizeof(_lib.GemmArgs)]
