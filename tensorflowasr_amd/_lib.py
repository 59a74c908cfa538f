"""ctypes binding of libtfasr_hip.so (include/tfasr_hip.h).

The product path has NO fallback: if the HIP library cannot be loaded every op raises. PyTorch is used
only for device memory, streams and torch.distributed; all arithmetic on the hot path is in the .so.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_long, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFASR_LIB") or os.path.join(HERE, "lib", "libtfasr_hip.so")  # TFASR_LIB: A/B runs of two builds on one box

TFASR_F32, TFASR_BF16 = 0, 1
ACT_NONE, ACT_SWISH, ACT_TANH, ACT_SIGMOID, ACT_TANH_OUT, ACT_FACTOR = 0, 1, 2, 3, 4, 5


class TfasrError(RuntimeError):
    pass


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("D", c_void_p),
        ("bias", c_void_p), ("res", c_void_p), ("dact_z", c_void_p), ("prez", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("ldb", c_int), ("ldd", c_int),
        ("trans_a", c_int), ("trans_b", c_int),
        ("nb1", c_int), ("nb2", c_int),
        ("sA1", c_long), ("sA2", c_long), ("sB1", c_long), ("sB2", c_long), ("sD1", c_long), ("sD2", c_long),
        ("alpha", c_float), ("beta", c_float),
        ("act", c_int), ("dact", c_int), ("dtype", c_int), ("out_f32", c_int), ("accumulate", c_int), ("split_k", c_int),
        ("drop_p", c_float), ("drop_seed", c_long),
        ("ws", c_void_p), ("ws_elems", c_long), ("colsum", c_void_p),
        ("lse_part", c_void_p), ("lse_parts", c_int), ("row_label", c_void_p), ("pick", c_void_p),
        ("seg_a_off", c_void_p), ("seg_b_off", c_void_p), ("seg_k", c_int),
        ("rgrad_coef", c_void_p),
        ("bns_x", c_void_p), ("bns_fin", c_void_p), ("bns_out", c_void_p), ("bns_copies", c_int), ("bns_c", c_int),
    ]


BLOCK_PARAM_NAMES = [  # order of tfasr_block_param_t (include/tfasr_hip.h); names relative to "enc/block{i}/"
    "ff1/ln/g", "ff1/ln/b", "ff1/d1/w", "ff1/d1/b", "ff1/d2/w", "ff1/d2/b",
    "ff2/ln/g", "ff2/ln/b", "ff2/d1/w", "ff2/d1/b", "ff2/d2/w", "ff2/d2/b",
    "mhsa/ln/g", "mhsa/ln/b", "mhsa/qkv/w", "mhsa/qkv/b", "mhsa/pos/w", "mhsa/pos/b", "mhsa/o/w", "mhsa/o/b", "/enc/u", "/enc/v",
    "conv/ln/g", "conv/ln/b", "conv/pw1/w", "conv/pw1/b", "conv/dw/w", "conv/dw/b", "conv/bn/g", "conv/bn/b", "conv/pw2/w", "conv/pw2/b",
    "ln/g", "ln/b",
]
PHASE_A, PHASE_B = 1, 2


class BlockCfg(Structure):
    _fields_ = [
        ("B", c_int), ("T", c_int), ("d", c_int), ("H", c_int), ("dh", c_int), ("dff", c_int), ("ksize", c_int),
        ("dtype", c_int), ("training", c_int), ("save", c_int), ("use_mask", c_int), ("force_unfused", c_int), ("world", c_int),
        ("site0", c_int), ("drop_epoch", c_long),
        ("drop_p", c_float), ("ffm_res", c_float), ("mhsa_res", c_float), ("conv_res", c_float), ("ln_eps", c_float), ("bn_eps", c_float),
        ("bn_momentum", c_float),
        ("chunk_size", c_int), ("history_size", c_int), ("dw_norm_layer", c_int), ("dh_logical", c_int),
    ]


class BlockParams(Structure):
    _fields_ = [
        ("flat", c_void_p), ("shadow", c_void_p), ("grad", c_void_p), ("bn_mm", c_void_p), ("bn_mv", c_void_p), ("pe", c_void_p),
        ("off", c_long * len(BLOCK_PARAM_NAMES)),
    ]


class BlockIO(Structure):
    _fields_ = [
        ("x_in", c_void_p), ("x_out", c_void_p), ("dy", c_void_p), ("dx", c_void_p), ("lengths", c_void_p),
        ("bn_stats", c_void_p), ("bn_bstats", c_void_p),
        ("stash", c_void_p), ("stash_bytes", c_size_t), ("scratch", c_void_p), ("scratch_bytes", c_size_t),
        ("prezeroed", c_int), ("dpext_zero", c_void_p), ("wgrad_slot", c_int),
        ("pext_pre", c_void_p), ("defer_pos_grad", c_int), ("ln_part_ext", c_void_p), ("ln_part_ext_floats", c_size_t),
        ("dcv_keep", c_void_p), ("ds_keep", c_void_p), ("qv_keep", c_void_p), ("bn_stats_copies", c_int),
    ]


def _parse_header():
    """Derive the ctypes signature table from include/tfasr_hip.h (single source of truth for the ABI)."""
    import re

    hdr = os.path.join(HERE, "..", "include", "tfasr_hip.h")
    if not os.path.exists(hdr):
        hdr = os.path.join(HERE, "tfasr_hip.h")
    src = open(hdr).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    sigs = {}
    for m in re.finditer(r"(const\s+char\s*\*|int|size_t)\s+(tfasr_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        res = c_char_p if "char" in ret else (c_size_t if ret == "size_t" else c_int)
        at = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "tfasr_gemm_args" in a:
                    at.append(POINTER(GemmArgs))
                elif "size_t*" in a.replace(" ", ""):
                    at.append(POINTER(c_size_t))
                elif "*" in a:
                    at.append(c_void_p)
                elif a.startswith("size_t"):
                    at.append(c_size_t)
                elif a.startswith("long"):
                    at.append(c_long)
                elif a.startswith("float"):
                    at.append(c_float)
                elif a.startswith("int") or a.startswith("int32_t"):
                    at.append(c_int)
                else:
                    raise TfasrError(f"cannot map C parameter '{a}' of {name}")
        sigs[name] = (res, at)
    return sigs


SIGNATURES = _parse_header()

_lib = None


def load(build_if_missing=True):
    """Load (building first if the .so is absent and hipcc exists). Raises TfasrError on failure."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise TfasrError(f"{LIB_PATH} is missing: run `python -m tensorflowasr_amd.build`")
        from . import build as _build

        _build.build(verbose=False)
    # The process must hold ONE HIP runtime: PyTorch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's), and
    # whichever copy is mapped first serves both.  Import torch first so device memory, streams and our kernels all live
    # in the runtime torch initialised (loading ours first made torch's later HIP calls see "no ROCm-capable device").
    import torch  # noqa: F401

    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # loud, no fallback
        raise TfasrError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TfasrError(f"{LIB_PATH} does not export {name}; rebuild with `python -m tensorflowasr_amd.build --force`") from e
        fn.restype = res
        fn.argtypes = args
    if lib.tfasr_abi_version() != ABI_VERSION:
        raise TfasrError(f"ABI mismatch: library {lib.tfasr_abi_version()} vs python {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


ABI_VERSION = 43


STATUS_UNSUPPORTED = 3


class TfasrUnsupported(TfasrError):
    pass


def check(status, what=""):
    if status != 0:
        msg = load().tfasr_status_string(status).decode()
        raise TfasrError(f"{what}: {msg} (status {status})")
