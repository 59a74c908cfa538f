"""Thin torch-tensor -> C-ABI adapters (raw device pointers + the current HIP stream).

Every function here requires CUDA(HIP) tensors and fails loudly otherwise: there is no CPU fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_SIGMOID, ACT_SWISH, ACT_TANH, TFASR_BF16, TFASR_F32, GemmArgs, check  # noqa: F401

_WS_CACHE = {}


def _dt(t):
    if t.dtype == torch.float32:
        return TFASR_F32
    if t.dtype == torch.bfloat16:
        return TFASR_BF16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.TfasrError("tensorflowasr_amd ops need HIP device tensors (no CPU fallback on the product path)")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes, device, tag="ws"):
    """Grow-only per-(device, tag) scratch buffer (uint8)."""
    key = (str(device), tag)
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _WS_CACHE[key] = buf
    return buf


# --------------------------------------------------------------------------------------------- RNN-T
def rnnt_loss_workspace_size(B, T, U1, V):
    n = ctypes.c_size_t(0)
    check(_lib.load().tfasr_rnnt_loss_workspace_size(B, T, U1, V, ctypes.byref(n)), "rnnt_loss_workspace_size")
    return n.value


def rnnt_loss_fwd_bwd(logits, labels, label_len, logit_len, grad_scale=None, grads=None, want_grads=True, blank=0):
    """logits [B,T,U1,V] (f32|bf16, contiguous) -> (costs [B] f32, grads or None). grads may be `logits` (in place)."""
    assert logits.dim() == 4 and logits.is_contiguous()
    B, T, U1, V = logits.shape
    assert labels.shape == (B, U1 - 1) and labels.dtype == torch.int32 and labels.is_contiguous()
    assert label_len.dtype == torch.int32 and logit_len.dtype == torch.int32
    costs = torch.empty(B, dtype=torch.float32, device=logits.device)
    if want_grads and grads is None:
        grads = torch.empty_like(logits)
    nbytes = rnnt_loss_workspace_size(B, T, U1, V)
    ws = workspace(nbytes, logits.device, "rnnt")
    check(
        _lib.load().tfasr_rnnt_loss(
            _p(logits), _p(grads) if want_grads else None, _p(labels), _p(label_len), _p(logit_len),
            _p(grad_scale), B, T, U1, V, blank, _dt(logits), _p(costs), _p(ws), ws.numel(), _stream()),
        "rnnt_loss")
    return costs, (grads if want_grads else None)


# ---------------------------------------------------------------------------------------------- GEMM
def gemm(A, B, out, M, N, K, lda, ldb, ldd, trans_a=False, trans_b=False, bias=None, res=None, dact_z=None,
         prez=None, alpha=1.0, beta=1.0, act=ACT_NONE, dact=ACT_NONE, nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sD=(0, 0),
         accumulate=False, split_k=1):
    """Raw strided (two-level batched) GEMM; see include/tfasr_hip.h."""
    a = GemmArgs()
    a.A, a.B, a.D = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.dact_z = dact_z.data_ptr() if dact_z is not None else None
    a.prez = prez.data_ptr() if prez is not None else None
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldb, a.ldd = lda, ldb, ldd
    a.trans_a, a.trans_b = int(trans_a), int(trans_b)
    a.nb1, a.nb2 = nb1, nb2
    a.sA1, a.sA2, a.sB1, a.sB2, a.sD1, a.sD2 = sA[0], sA[1], sB[0], sB[1], sD[0], sD[1]
    a.alpha, a.beta = alpha, beta
    a.act, a.dact = act, dact
    a.dtype = _dt(A)
    assert B.dtype == A.dtype and A.is_cuda and B.is_cuda and out.is_cuda
    a.out_f32 = int(out.dtype == torch.float32 and A.dtype != torch.float32) or int(accumulate)
    a.accumulate = int(accumulate)
    a.split_k = split_k
    if accumulate:
        assert out.dtype == torch.float32
    check(_lib.load().tfasr_gemm(ctypes.byref(a), _stream()), "gemm")
    return out


def matmul(A, B, trans_a=False, trans_b=False, out=None, out_dtype=None, **kw):
    """2-D convenience: op(A)[M,K] @ op(B)[K,N]."""
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    N = B.shape[0] if trans_b else B.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or A.dtype, device=A.device)
    return gemm(A, B, out, M, N, K, A.stride(0), B.stride(0), out.stride(0), trans_a, trans_b, **kw)
