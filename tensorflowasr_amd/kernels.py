"""Thin torch-tensor -> C-ABI adapters (raw device pointers + the current HIP stream).

Every function here requires CUDA(HIP) tensors and fails loudly otherwise: there is no CPU fallback.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_FACTOR, ACT_NONE, ACT_SIGMOID, ACT_SWISH, ACT_TANH, ACT_TANH_OUT, TFASR_BF16, TFASR_F32, GemmArgs, check  # noqa: F401

_WS_CACHE = {}
_SPLITK_WS = False  # k-slices of a split product through a workspace (deterministic sums; measured slower than the f32 atomics: tests set it)


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
    if not t.is_contiguous():
        raise _lib.TfasrError(f"non-contiguous tensor passed to a HIP kernel (shape {tuple(t.shape)}, strides {t.stride()})")
    return ctypes.c_void_p(t.data_ptr())


def _pv(t):
    """pointer of a strided VIEW (the kernel receives the strides explicitly)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.TfasrError("tensorflowasr_amd ops need HIP device tensors (no CPU fallback on the product path)")
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The HIP stream torch currently launches on (raw hipStream_t).  torch.cuda.current_stream() builds a Stream object per
    call (~9 us); the raw query is one C call, which matters at ~1500 launches per train step."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
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


def rnnt_row_labels(labels, label_len, logit_len, cell_off, total_cells, T, V):
    """[total_cells] i32: the label id each packed lattice row can emit (-1: none)."""
    B, U = labels.shape
    out = torch.empty(total_cells, dtype=torch.int32, device=labels.device)
    check(_L().tfasr_rnnt_row_labels(_p(labels), _p(label_len), _p(logit_len), _p(cell_off), total_cells, B, T, U + 1, V, _p(out), _stream()),
          "rnnt_row_labels")
    return out


def rnnt_loss_packed(logits, labels, label_len, logit_len, cell_off, total_cells, T, grad_scale=None, grads=None, want_grads=True, blank=0,
                     stats=None):
    """logits [total_cells, V] over the packed lattice (see include/tfasr_hip.h) -> (costs [B], grads).  stats = (lse_part, pick)
    from the projection GEMM's epilogue (kernels.gemm lse=...): skips the first pass over the logits."""
    assert logits.dim() == 2 and logits.is_contiguous() and cell_off.dtype == torch.int64
    B, U = labels.shape
    V = logits.shape[1]
    costs = torch.empty(B, dtype=torch.float32, device=logits.device)
    if want_grads and grads is None:
        grads = torch.empty_like(logits)
    nbytes = rnnt_loss_workspace_size(1, total_cells, 1, V)
    ws = workspace(nbytes, logits.device, "rnnt")
    if stats is not None:
        part, pick = stats
        check(_lib.load().tfasr_rnnt_loss_packed_stats(
            _p(logits), _p(grads) if want_grads else None, _p(labels), _p(label_len), _p(logit_len), _p(grad_scale), _p(cell_off),
            total_cells, _p(part), part.shape[1], _p(pick), B, T, U + 1, V, blank, _dt(logits), _p(costs), _p(ws), ws.numel(), _stream()),
            "rnnt_loss_packed_stats")
        return costs, (grads if want_grads else None)
    check(_lib.load().tfasr_rnnt_loss_packed(
        _p(logits), _p(grads) if want_grads else None, _p(labels), _p(label_len), _p(logit_len), _p(grad_scale), _p(cell_off),
        total_cells, B, T, U + 1, V, blank, _dt(logits), _p(costs), _p(ws), ws.numel(), _stream()), "rnnt_loss_packed")
    return costs, (grads if want_grads else None)


def rnnt_loss_packed_coef(labels, label_len, logit_len, cell_off, total_cells, T, V, stats, grad_scale=None, blank=0):
    """The loss WITHOUT logits (joint "recompute" variant): costs [B] and the per-row coefficients [total_cells, 4] f32 from which a
    re-computed logit tile becomes the gradient (gemm(..., rgrad=(coef, row_label)))."""
    B, U = labels.shape
    part, pick = stats
    costs = torch.empty(B, dtype=torch.float32, device=part.device)
    coef = torch.empty(total_cells, 4, dtype=torch.float32, device=part.device)
    ws = workspace(rnnt_loss_workspace_size(1, total_cells, 1, V), part.device, "rnnt")
    check(_lib.load().tfasr_rnnt_loss_packed_coef(_p(labels), _p(label_len), _p(logit_len), _p(grad_scale), _p(cell_off), total_cells, _p(part),
                                                  part.shape[1], _p(pick), B, T, U + 1, V, blank, _p(costs), _p(coef), _p(ws), ws.numel(), _stream()),
          "rnnt_loss_packed_coef")
    return costs, coef


# ---------------------------------------------------------------------------------------------- GEMM
def gemm(A, B, out, M, N, K, lda, ldb, ldd, trans_a=False, trans_b=False, bias=None, res=None, dact_z=None,
         prez=None, alpha=1.0, beta=1.0, act=ACT_NONE, dact=ACT_NONE, nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sD=(0, 0),
         accumulate=False, split_k=1, drop_p=0.0, drop_seed=0, colsum=None, lse=None, seg=None, rgrad=None, bns=None):
    """Raw strided (two-level batched) GEMM; see include/tfasr_hip.h.  bns = (x [M, N], fin [4N] f32, out [copies, 2N] f32): the BatchNorm
    backward sums of the output in the epilogue (tfasr_gemm_args.bns_*; raises TfasrUnsupported when the product is not the plain NT one).  lse = (lse_part [M, parts, 2] f32, row_label [M] i32,
    pick [M, 2] f32): fused log-softmax statistics of the output rows (raises TfasrUnsupported when the fast path cannot); with it
    `out` may be None (statistics only).  rgrad = (coef [M, 4] f32, row_label [M] i32): the product is a re-computed logit tile and the
    epilogue stores the RNN-T loss gradient instead (tfasr_gemm_args.rgrad_coef)."""
    a = _gemm_args(A, B, out, M, N, K, lda, ldb, ldd, trans_a, trans_b, bias, res, dact_z, prez, alpha, beta, act, dact, nb1, nb2, sA, sB, sD,
                   accumulate, split_k, drop_p, drop_seed, colsum)
    if rgrad is not None:
        coef, row_label = rgrad
        assert coef.dtype == torch.float32 and coef.is_contiguous() and row_label.dtype == torch.int32
        a.rgrad_coef, a.row_label = coef.data_ptr(), row_label.data_ptr()
    if seg is not None:  # (seg_a_off i64 device tensor, seg_b_off or None, seg_k): K-segmented operands
        a.seg_a_off, a.seg_b_off, a.seg_k = seg[0].data_ptr(), (seg[1].data_ptr() if seg[1] is not None else None), int(seg[2])
        assert seg[0].dtype == torch.int64 and seg[0].is_cuda
    if bns is not None:
        bx, bfin, bout = bns[:3]
        bc = int(bns[3]) if len(bns) > 3 else N   # channels (the columns are (position, channel) pairs when < N)
        assert bfin.dtype == torch.float32 and bout.dtype == torch.float32 and bout.is_contiguous()
        a.bns_x, a.bns_fin, a.bns_out, a.bns_copies = bx.data_ptr(), bfin.data_ptr(), bout.data_ptr(), bout.numel() // (2 * bc)
        a.bns_c = 0 if bc == N else bc
    if lse is not None:
        part, row_label, pick = lse
        assert part.dtype == torch.float32 and pick.dtype == torch.float32 and row_label.dtype == torch.int32
        a.lse_part, a.lse_parts, a.row_label, a.pick = part.data_ptr(), part.shape[1], row_label.data_ptr(), pick.data_ptr()
    st = _lib.load().tfasr_gemm(ctypes.byref(a), _stream())
    if (lse is not None or seg is not None or rgrad is not None or bns is not None) and st == _lib.STATUS_UNSUPPORTED:
        raise _lib.TfasrUnsupported("gemm: fused row statistics / K-segments / gradient epilogue are not available for this product")
    check(st, "gemm")
    return out


def gemm_group(calls):
    """Several independent products in one launch when they qualify (tfasr_gemm_group: bf16 weight gradients), else one by
    one.  `calls` = list of dicts with gemm()'s keyword arguments."""
    arr = (GemmArgs * len(calls))()
    for i, kw in enumerate(calls):
        arr[i] = _gemm_args(**kw)
    check(_lib.load().tfasr_gemm_group(arr, len(calls), _stream()), "gemm_group")


def _gemm_args(A, B, out, M, N, K, lda, ldb, ldd, trans_a=False, trans_b=False, bias=None, res=None, dact_z=None,
               prez=None, alpha=1.0, beta=1.0, act=ACT_NONE, dact=ACT_NONE, nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sD=(0, 0),
               accumulate=False, split_k=1, drop_p=0.0, drop_seed=0, colsum=None):
    a = GemmArgs()
    a.A, a.B, a.D = A.data_ptr(), B.data_ptr(), (out.data_ptr() if out is not None else None)
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
    assert B.dtype == A.dtype and A.is_cuda and B.is_cuda and (out is None or out.is_cuda)
    a.out_f32 = int(out is not None and out.dtype == torch.float32)
    a.accumulate = int(accumulate)
    a.split_k = split_k
    a.drop_p, a.drop_seed = drop_p, drop_seed
    a.colsum = colsum.data_ptr() if colsum is not None else None
    if accumulate:
        assert out.dtype == torch.float32
        if _SPLITK_WS and split_k > 1 and nb1 * nb2 == 1:  # opt-in: k-slices reduce through a workspace (deterministic sums)
            ws = workspace(4 * split_k * M * N, out.device, "splitk_%d" % (_raw_stream(torch.cuda.current_device()) if _raw_stream else 0))
            a.ws, a.ws_elems = ws.data_ptr(), ws.numel() // 4
    return a


def matmul(A, B, trans_a=False, trans_b=False, out=None, out_dtype=None, **kw):
    """2-D convenience: op(A)[M,K] @ op(B)[K,N]."""
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    N = B.shape[0] if trans_b else B.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or A.dtype, device=A.device)
    return gemm(A, B, out, M, N, K, A.stride(0), B.stride(0), out.stride(0), trans_a, trans_b, **kw)


# ------------------------------------------------------------------------------------- norms
def _L():
    return _lib.load()


def layernorm_fwd(x, gamma, beta, eps=1e-3, save_stats=True):
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    check(_L().tfasr_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, C, eps, _dt(x), _stream()), "layernorm_fwd")
    return y, mean, rstd


def ffn_fused_fwd(x, gamma, beta, W1, b1, W2, b2, res_factor, drop_p=0.0, seed1=0, seed2=0, save_z=True, eps=1e-3, z_factor=False):
    """tfasr_ffn_fused_fwd2: FFModule forward in one launch.  Returns (y, ln, mean, rstd, z, h), or None when the shape is outside the fused
    kernel's range (the caller keeps the layernorm + two GEMM route).  z_factor: `z` is the data gradient's factor swish'(z) * mask1 / (1 - p)
    (use it as dact_z with dact=ACT_FACTOR and no dropout term) instead of the pre-activation."""
    rows, d = x.shape
    F = W1.shape[1]
    y, ln = torch.empty_like(x), torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    z = torch.empty(rows, F, dtype=x.dtype, device=x.device) if save_z else None
    h = torch.empty(rows, F, dtype=x.dtype, device=x.device)
    st = _L().tfasr_ffn_fused_fwd2(_p(x), _p(gamma), _p(beta), _p(W1), _p(b1), _p(W2), _p(b2), _p(y), _p(ln), _p(mean), _p(rstd), _pv(z), int(bool(z_factor)),
                                   _p(h), rows, d, F, eps, float(res_factor), float(drop_p), int(seed1), int(seed2), _dt(x), _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return None
    check(st, "ffn_fused_fwd")
    return y, ln, mean, rstd, z, h


def ln_dense_fwd(x, gamma, beta, W, b, eps=1e-3):
    """tfasr_ln_dense_fwd: out = LayerNorm(x) W + b in one launch.  Returns (out, ln, mean, rstd) or None (shape outside the kernel's range)."""
    rows, d = x.shape
    N = W.shape[1]
    out = torch.empty(rows, N, dtype=x.dtype, device=x.device)
    ln = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    st = _L().tfasr_ln_dense_fwd(_p(x), _p(gamma), _p(beta), _p(W), _p(b), _p(out), _p(ln), _p(mean), _p(rstd), rows, d, N, eps, _dt(x), _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return None
    check(st, "ln_dense_fwd")
    return out, ln, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, add=None, dx=None, dropped=None, drop_p=0.0, drop_seed=0):
    """dropped (optional, same shape as dx): also receives dropout(dx, drop_p, drop_seed) from the same kernel."""
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    if dx is None:
        dx = torch.empty_like(x)
    if dropped is not None:
        check(_L().tfasr_layernorm_bwd_drop(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(add), _p(dx), _p(dgamma), _p(dbeta), _p(dropped),
                                            float(drop_p), int(drop_seed), rows, C, _dt(x), _stream()), "layernorm_bwd_drop")
        return dx
    check(_L().tfasr_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(add), _p(dx), _p(dgamma), _p(dbeta), rows, C, _dt(x), _stream()), "layernorm_bwd")
    return dx


def layernorm_bwd_fold(dy, x, gamma, mean, rstd, dgamma, dbeta, add=None):
    """layernorm_bwd with the gamma / beta gradients through per-block partial sums + tfasr_layernorm_bwd_fold (what the native block
    executor does for the five LayerNorms of a block with ONE fold launch); returns dx, or None when the shape has no such kernel."""
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    nblk = _L().tfasr_layernorm_bwd_part_blocks(rows, C, _dt(x))
    if nblk <= 0:
        return None
    dx = torch.empty_like(x)
    part = torch.empty(nblk * 2 * C, dtype=torch.float32, device=x.device)
    check(_L().tfasr_layernorm_bwd_part(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(add), _p(dx), _p(part), None, 0.0, 0, rows, C, _dt(x),
                                        _stream()), "layernorm_bwd_part")
    dg = (ctypes.c_void_p * 1)(dgamma.data_ptr())
    db = (ctypes.c_void_p * 1)(dbeta.data_ptr())
    check(_L().tfasr_layernorm_bwd_fold(_p(part), 1, nblk, C, dg, db, _stream()), "layernorm_bwd_fold")
    return dx


def dense_ln_bwd(dy, W, x, gamma, mean, rstd, dgamma, dbeta, add=None, dropped=None, drop_p=0.0, drop_seed=0, alpha=1.0):
    """tfasr_dense_ln_bwd: dx of `Dense(LayerNorm(x))` in one launch (dln = alpha * dy @ W^T never stored), gamma / beta gradients through
    partial sums + tfasr_layernorm_bwd_fold.  dy [rows, K], W [d, K] (the Dense kernel as stored).  Returns dx, or None when the shape is
    outside the fused kernel's range (the caller keeps gemm + layernorm_bwd)."""
    rows, d = x.numel() // x.shape[-1], x.shape[-1]
    K = dy.shape[-1]
    nblk = _L().tfasr_layernorm_bwd_part_blocks(rows, d, _dt(x))
    if nblk <= 0:
        return None
    dx = torch.empty_like(x)
    part = torch.empty(nblk * 2 * d, dtype=torch.float32, device=x.device)
    st = _L().tfasr_dense_ln_bwd(_p(dy), _p(W), K, _p(x), _p(gamma), _p(mean), _p(rstd), _p(add), _p(dx), _p(part), nblk, _p(dropped), float(drop_p),
                                 int(drop_seed), rows, d, float(alpha), _dt(x), _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return None
    check(st, "dense_ln_bwd")
    dg = (ctypes.c_void_p * 1)(dgamma.data_ptr())
    db = (ctypes.c_void_p * 1)(dbeta.data_ptr())
    check(_L().tfasr_layernorm_bwd_fold(_p(part), 1, nblk, d, dg, db, _stream()), "layernorm_bwd_fold")
    return dx


def bn_stats(x, stats):
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    check(_L().tfasr_bn_stats(_p(x), _p(stats), rows, C, _dt(x), _stream()), "bn_stats")


def bn_finalize(stats, count, gamma, beta, fin, moving_mean, moving_var, momentum=0.99, eps=1e-3, training=True):
    C = gamma.numel()
    check(_L().tfasr_bn_finalize(_p(stats), float(count), _p(gamma), _p(beta), _p(fin), _p(moving_mean), _p(moving_var), momentum, eps, C, int(training), _stream()), "bn_finalize")


def bn_finalize_apply_fwd(x, stats, count, gamma, beta, fin, moving_mean, moving_var, act=ACT_NONE, y=None, momentum=0.99, eps=1e-3, training=True,
                          copies=1):
    """tfasr_bn_finalize + tfasr_bn_apply_fwd in one launch (fin and the moving statistics written as bn_finalize would); returns y, or None
    when the channel count is outside the row kernel's range (the caller runs the two launches).  copies > 1: stats is [copies, 2C] (the
    layout dwconv_fwd_stats accumulates into), summed on the fly."""
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    if y is None:
        y = torch.empty_like(x)
    st = _L().tfasr_bn_finalize_apply_fwd_copies(_p(x), _p(stats), int(copies), float(count), _p(gamma), _p(beta), _p(fin), _p(moving_mean), _p(moving_var),
                                                 momentum, eps, _p(y), rows, C, act, int(training), _dt(x), _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return None
    check(st, "bn_finalize_apply_fwd")
    return y


def bn_apply_fwd(x, fin, act=ACT_NONE, y=None):
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    if y is None:
        y = torch.empty_like(x)
    check(_L().tfasr_bn_apply_fwd(_p(x), _p(fin), _p(y), rows, C, act, _dt(x), _stream()), "bn_apply_fwd")
    return y


def bn_bwd_stats(x, dy, fin, bstats, act=ACT_NONE):
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    check(_L().tfasr_bn_bwd_stats(_p(x), _p(dy), _p(fin), _p(bstats), rows, C, act, _dt(x), _stream()), "bn_bwd_stats")


def bn_apply_bwd(x, dy, fin, bstats, count, act=ACT_NONE, dx=None, dgamma=None, dbeta=None, grad_scale=1.0, copies=1):
    """dgamma / dbeta (f32 views of the gradient buffer): += grad_scale * the two statistics, in the same launch.  copies > 1: bstats is
    [copies, 2C] (the layout the gemm's bns epilogue accumulates into), added up on the fly."""
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    if dx is None:
        dx = torch.empty_like(x)
    check(_L().tfasr_bn_apply_bwd_grads_copies(_p(x), _p(dy), _p(fin), _p(bstats), int(copies), float(count), _p(dx), rows, C, act,
                                               _p(dgamma) if dgamma is not None else None, _p(dbeta) if dbeta is not None else None, float(grad_scale),
                                               _dt(x), _stream()), "bn_apply_bwd")
    return dx


# --------------------------------------------------------------------------------- pointwise
def cast(src, dst):
    check(_L().tfasr_cast(_p(src), _p(dst), src.numel(), _dt(src), _dt(dst), _stream()), "cast")
    return dst


def cast_colsum_many(x, y, stride, nmat, rows, C, colsums):
    """f32 matrices x + b*stride [rows, C] -> bf16 copies in y (same strides) and colsums[b] += their f32 column sums (one launch)."""
    # (the kernel takes its pointer table as an argument: at most 64 matrices per launch - a model with more blocks takes several)
    for b0 in range(0, int(nmat), 64):
        n = min(64, int(nmat) - b0)
        arr = (ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in colsums[b0:b0 + n]])
        check(_L().tfasr_cast_colsum_many(x.data_ptr() + b0 * int(stride) * x.element_size(), y.data_ptr() + b0 * int(stride) * y.element_size(),
                                          int(stride), n, int(rows), int(C), arr, _stream()), "cast_colsum_many")
    return y


def dropout(x, p, seed, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(_L().tfasr_dropout(_p(x), _p(out), x.numel(), p, seed, _dt(x), _stream()), "dropout")
    return out


def colsum(x2d, out, scale=1.0, rows=None, C=None, ld=None):
    rows = x2d.shape[0] if rows is None else rows
    C = x2d.shape[1] if C is None else C
    ld = x2d.stride(0) if ld is None else ld
    check(_L().tfasr_colsum(_pv(x2d), ld, _p(out), rows, C, scale, _dt(x2d), _stream()), "colsum")


def glu_fwd(x):
    rows, C2 = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty(*x.shape[:-1], C2 // 2, dtype=x.dtype, device=x.device)
    check(_L().tfasr_glu_fwd(_p(x), _p(y), rows, C2 // 2, _dt(x), _stream()), "glu_fwd")
    return y


def glu_bwd(x, dy):
    rows, C2 = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x)
    check(_L().tfasr_glu_bwd(_p(x), _p(dy), _p(dx), rows, C2 // 2, _dt(x), _stream()), "glu_bwd")
    return dx


def dwconv_fwd(x, w, bias):
    B, T, C = x.shape
    y = torch.empty_like(x)
    check(_L().tfasr_dwconv_fwd(_p(x), _p(w), _p(bias), _p(y), B, T, C, w.shape[0], _dt(x), _stream()), "dwconv_fwd")
    return y


def dwconv_fwd_stats(x, w, bias, stats):
    """Depthwise conv + the BatchNorm statistics of its output in one launch: stats [copies, 2C] f32 is accumulated into (sum | sum of
    squares of the bf16-rounded outputs per channel, spread over the copies).  Returns y, or None when the fused kernel does not take the
    shape / dtype (the caller runs dwconv_fwd + bn_stats)."""
    B, T, C = x.shape
    y = torch.empty_like(x)
    st = _L().tfasr_dwconv_fwd_stats(_p(x), _p(w), _p(bias), _p(y), _p(stats), int(stats.numel() // (2 * C)), B, T, C, w.shape[0], _dt(x), _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return None
    check(st, "dwconv_fwd_stats")
    return y


def bn_dwconv_bwd_data_glu(bn_x, dsw, fin, bstats, count, w, glu_x, dgamma=None, dbeta=None, grad_scale=1.0):
    """BatchNorm backward apply pass (swish) + depthwise data gradient + GLU backward in one launch; bstats [copies, 2C].  Returns
    (dcv [B, T, C], dglu [B, T, 2C]) or None when the fused kernel does not take the shape."""
    B, T, C = bn_x.shape
    dcv = torch.empty_like(bn_x)
    dglu = torch.empty(B, T, 2 * C, dtype=bn_x.dtype, device=bn_x.device)
    st = _L().tfasr_bn_dwconv_bwd_data_glu(_p(bn_x), _p(dsw), _p(fin), _p(bstats), int(bstats.numel() // (2 * C)), float(count),
                                           _p(dgamma) if dgamma is not None else None, _p(dbeta) if dbeta is not None else None, float(grad_scale), _p(dcv),
                                           _p(w), _p(glu_x), _p(dglu), B, T, C, w.shape[0], _dt(bn_x), _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return None
    check(st, "bn_dwconv_bwd_data_glu")
    return dcv, dglu


def glu_dwconv_fwd_stats(glu_x, w, bias, stats):
    """GLU + depthwise conv + BatchNorm statistics in one launch: glu_x [B, T, 2C] -> (g [B, T, C], y [B, T, C]); stats [copies, 2C] is
    accumulated into.  None when the fused kernel does not take the shape (glu_fwd + dwconv_fwd_stats)."""
    B, T, C2 = glu_x.shape
    C = C2 // 2
    g = torch.empty(B, T, C, dtype=glu_x.dtype, device=glu_x.device)
    y = torch.empty_like(g)
    st = _L().tfasr_glu_dwconv_fwd_stats(_p(glu_x), _p(g), _p(w), _p(bias), _p(y), _p(stats), int(stats.numel() // (2 * C)), B, T, C, w.shape[0], _dt(glu_x), _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return None
    check(st, "glu_dwconv_fwd_stats")
    return g, y


def dwconv_bwd_data(dy, w):
    B, T, C = dy.shape
    dx = torch.empty_like(dy)
    check(_L().tfasr_dwconv_bwd_data(_p(dy), _p(w), _p(dx), B, T, C, w.shape[0], _dt(dy), _stream()), "dwconv_bwd_data")
    return dx


def dwconv_bwd_data_glu(dy, w, glu_x):
    """Depthwise data gradient + the backward of the GLU in front of the conv in one launch (bf16); None if the fused kernel does not apply."""
    B, T, C = dy.shape
    dglu = torch.empty_like(glu_x)
    st = _L().tfasr_dwconv_bwd_data_glu(_p(dy), _p(w), _p(glu_x), _p(dglu), B, T, C, w.shape[0], _dt(dy), _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return None
    check(st, "dwconv_bwd_data_glu")
    return dglu


def dwconv_bwd_weight(x, dy, dw, dbias):
    B, T, C = x.shape
    n = ctypes.c_size_t(0)
    check(_L().tfasr_dwconv_bwd_weight_workspace_size(B, T, C, dw.shape[0], ctypes.byref(n)), "dwconv_bwd_weight_workspace_size")
    ws = workspace(n.value, x.device, "dwconv_wgrad_%d" % (_raw_stream(torch.cuda.current_device()) if _raw_stream else 0))
    check(_L().tfasr_dwconv_bwd_weight_ws(_p(x), _p(dy), _p(dw), _p(dbias), B, T, C, dw.shape[0], _dt(x), _p(ws), ws.numel(), _stream()),
          "dwconv_bwd_weight_ws")


def dwconv_bwd_weight_many(items):
    """Depthwise weight gradients of several layers of ONE shape as one launch pair: items = [(x [B,T,C], dy [B,T,C], dw [K,C] f32, dbias or None)]."""
    n = len(items)
    B, T, C = items[0][0].shape
    Kk = items[0][2].shape[0]
    need = ctypes.c_size_t(0)
    check(_L().tfasr_dwconv_bwd_weight_workspace_size(B, T, C, Kk, ctypes.byref(need)), "dwconv_bwd_weight_workspace_size")
    ws = workspace(n * need.value, items[0][0].device, "dw_wgrad_many")
    xa = (ctypes.c_void_p * n)(*[_p(t[0]).value for t in items])
    da = (ctypes.c_void_p * n)(*[_p(t[1]).value for t in items])
    wa = (ctypes.c_void_p * n)(*[t[2].data_ptr() for t in items])
    ba = (ctypes.c_void_p * n)(*[(t[3].data_ptr() if t[3] is not None else None) for t in items])
    check(_L().tfasr_dwconv_bwd_weight_many(xa, da, wa, ba, n, B, T, C, Kk, _dt(items[0][0]), _p(ws), ws.numel(), _stream()), "dwconv_bwd_weight_many")


def bias2_fwd(x, ldx, u, v, rows, C):
    y1 = torch.empty(rows, C, dtype=x.dtype, device=x.device)
    y2 = torch.empty(rows, C, dtype=x.dtype, device=x.device)
    check(_L().tfasr_bias2_fwd(_pv(x), ldx, _p(u), _p(v), _p(y1), _p(y2), rows, C, _dt(x), _stream()), "bias2_fwd")
    return y1, y2


def bias2_bwd(d1, d2, dx, lddx, du, dv, rows, C):
    check(_L().tfasr_bias2_bwd(_p(d1), _p(d2), _pv(dx), lddx, _p(du), _p(dv), rows, C, _dt(d1), _stream()), "bias2_bwd")


def embedding_fwd(idx, table, dtype):
    rows = idx.numel()
    V, E = table.shape
    out = torch.empty(*idx.shape, E, dtype=dtype, device=table.device)
    check(_L().tfasr_embedding_fwd(_p(idx), _p(table), _p(out), rows, E, V, _dt(out), _stream()), "embedding_fwd")
    return out


def embedding_bwd(idx, dout, dtable):
    V, E = dtable.shape
    check(_L().tfasr_embedding_bwd(_p(idx), _p(dout), _p(dtable), idx.numel(), E, V, _dt(dout), _stream()), "embedding_bwd")


def joint_fwd(enc, pred):
    B, T, J = enc.shape
    U1 = pred.shape[1]
    h = torch.empty(B, T, U1, J, dtype=enc.dtype, device=enc.device)
    check(_L().tfasr_joint_fwd(_p(enc), _p(pred), _p(h), B, T, U1, J, _dt(enc), _stream()), "joint_fwd")
    return h


def joint_bwd(h, dh):
    B, T, U1, J = h.shape
    denc = torch.empty(B, T, J, dtype=h.dtype, device=h.device)
    dpred = torch.empty(B, U1, J, dtype=h.dtype, device=h.device)
    check(_L().tfasr_joint_bwd(_p(h), _p(dh), _p(denc), _p(dpred), B, T, U1, J, _dt(h), _stream()), "joint_bwd")
    return denc, dpred


def joint_fwd_packed(enc, pred, cell_off, label_len, total_cells):
    B, T, J = enc.shape
    U1 = pred.shape[1]
    h = torch.empty(total_cells, J, dtype=enc.dtype, device=enc.device)
    check(_L().tfasr_joint_fwd_packed(_p(enc), _p(pred), _p(h), _p(cell_off), _p(label_len), total_cells, B, T, U1, J, _dt(enc), _stream()), "joint_fwd_packed")
    return h


def joint_bwd_packed(h, dh, cell_off, label_len, logit_len, B, T, U1):
    """h = None: dh already carries the tanh' factor (produced with dact=ACT_TANH_OUT)."""
    J = dh.shape[1]
    denc = torch.empty(B, T, J, dtype=dh.dtype, device=dh.device)
    dpred = torch.empty(B, U1, J, dtype=dh.dtype, device=dh.device)
    check(_L().tfasr_joint_bwd_packed(_p(h) if h is not None else None, _p(dh), _p(denc), _p(dpred), _p(cell_off), _p(label_len), _p(logit_len), B, T, U1, J, _dt(dh), _stream()), "joint_bwd_packed")
    return denc, dpred


def adam(p, g, m, v, n_reg, lr, step, beta1=0.9, beta2=0.999, eps=1e-7, weight_decay=0.0, l2=0.0, grad_scale=1.0, shadow=None):
    """shadow (bf16, p's shape): also receives the updated parameters rounded to bf16, in the same launch"""
    if shadow is not None:
        assert shadow.dtype == torch.bfloat16 and shadow.numel() == p.numel()
    check(_L().tfasr_adam_shadow(_p(p), _p(g), _p(m), _p(v), p.numel(), n_reg, lr, beta1, beta2, eps, weight_decay, l2, grad_scale, step, _p(shadow),
                                 _stream()), "adam")


def gauss_noise(x, stddev, seed):
    """x (f32, contiguous) += stddev * N(0,1), counter-based (seed, index)."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    check(_L().tfasr_gauss_noise(_p(x), x.numel(), float(stddev), int(seed) & 0x7FFFFFFFFFFFFFFF, _stream()), "gauss_noise")
    return x


def axpy(y, x, alpha=1.0):
    check(_L().tfasr_axpy(_p(y), _p(x), alpha, x.numel(), _stream()), "axpy")


def sumsq(p, n, out):
    check(_L().tfasr_sumsq(_p(p), n, _p(out), _stream()), "sumsq")


def specaugment(x, fmask, tmask, mask_value=0.0):
    B, T, F = x.shape[:3]
    nf = 0 if fmask is None else fmask.shape[1]
    nt = 0 if tmask is None else tmask.shape[1]
    check(_L().tfasr_specaugment(_p(x), _p(fmask), _p(tmask), nf, nt, B, T, F, mask_value, _dt(x), _stream()), "specaugment")
    return x


# --------------------------------------------------------------------------------- attention
def relattn_softmax_fwd(content, pos, lengths, T, use_mask=True, probs=None, chunk_size=None, history_size=None):
    """content [B,H,T,ldc], pos [B,H,T,ldp] (row strides may be padded); chunk_size / history_size = streaming mask."""
    B, H, _, ldc = content.shape
    ldp = pos.shape[3]
    if probs is None:
        probs = torch.empty_like(content)
    chunk = int(chunk_size) if chunk_size else 0
    hist = int(history_size) if history_size is not None else 0
    check(_L().tfasr_relattn_softmax_fwd_streaming(_p(content), _p(pos), _p(lengths), _p(probs), B, H, T, ldc, ldp, int(use_mask), chunk, hist,
                                                   _dt(content), _stream()), "relattn_softmax_fwd")
    return probs


def relattn_softmax_bwd(probs, dprobs, lengths, T, ldp, use_mask=True, dcontent=None, dpos=None):
    B, H, _, ldc = probs.shape
    if dcontent is None:
        dcontent = torch.empty_like(probs)
    if dpos is None:
        dpos = torch.empty(B, H, T, ldp, dtype=probs.dtype, device=probs.device)
    check(_L().tfasr_relattn_softmax_bwd(_p(probs), _p(dprobs), _p(lengths), _p(dcontent), _p(dpos), B, H, T, ldc, ldp, int(use_mask), _dt(probs), _stream()), "relattn_softmax_bwd")
    return dcontent, dpos


def _window(chunk_size, history_size):
    """(chunk, hist) of the streaming attention mask for the C ABI: chunk 0 = off, hist < 0 = unlimited history"""
    if not chunk_size:
        return 0, 0
    return int(chunk_size), (-1 if history_size is None else int(history_size))


def relattn_fused_fwd(qkv, ubias, vbias, pext, lengths, B, H, T, dh, scale, use_mask=True, chunk_size=None, history_size=None):
    out = torch.empty(B * T, H * dh, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    ck, hs = _window(chunk_size, history_size)
    check(_L().tfasr_relattn_fused_fwd(_p(qkv), _p(ubias), _p(vbias), _p(pext), _p(lengths), _p(out), _p(lse), B, H, T, dh, scale,
                                       int(use_mask), ck, hs, _dt(qkv), _stream()), "relattn_fused_fwd")
    return out, lse


def relattn_fused_bwd_q3(qkv, ubias, vbias, pext, lengths, o, dout, lse, dqkv, du, dv, dpext, B, H, T, dh, scale, use_mask=True, chunk_size=None,
                         history_size=None, ds=None, qv=None):
    """Query side of the fused attention backward (tfasr_relattn_fused_bwd_q3): writes dq = dqu + dqv into the q columns of dqkv
    [B*T, 3*H*dh], adds the u / v bias gradients into du / dv [H*dh] f32 and the bias row's share into dpext [2T, H*dh] f32.
    -> (ds [B,H,T,lds] unskewed score gradient, dvec [2,B,H,T], qu, qv = q + u / q + v for the key side and tfasr_relattn_dpext)."""
    lds = -(-T // 8) * 8
    if ds is None:
        ds = torch.empty(B, H, T, lds, dtype=qkv.dtype, device=qkv.device)
    dvec = torch.empty(2, B, H, T, dtype=torch.float32, device=qkv.device)  # rowsum(dout * o) | the bias-row score of every query
    qu = torch.empty(B * T, H * dh, dtype=qkv.dtype, device=qkv.device)
    if qv is None:
        qv = torch.empty(B * T, H * dh, dtype=qkv.dtype, device=qkv.device)
    check(_L().tfasr_relattn_fused_bwd_q3(_p(qkv), _p(ubias), _p(vbias), _p(pext), _p(lengths), _p(o), _p(dout), _p(lse), _p(dqkv), dqkv.stride(0),
                                          _p(du), _p(dv), _p(ds), _p(dvec), _p(dpext), _p(qu), _p(qv), B, H, T, dh, lds, scale, int(use_mask),
                                          *_window(chunk_size, history_size), _dt(qkv), _stream()), "relattn_fused_bwd_q3")
    return ds, dvec, qu, qv


def relattn_dpext(ds, qv, lengths, dpext, B, H, T, dh, use_mask=True):
    check(_L().tfasr_relattn_dpext(_p(ds), _p(qv), _p(lengths), _p(dpext), B, H, T, dh, ds.shape[3], int(use_mask), _dt(ds), _stream()), "relattn_dpext")
    return dpext


def relattn_fused_bwd_k(qkv, qu, qv, pext, lengths, dout, lse, dvec, dqkv, B, H, T, dh, scale, use_mask=True, chunk_size=None, history_size=None):
    check(_L().tfasr_relattn_fused_bwd_k(_p(qkv), _p(qu), _p(qv), _p(pext), _p(lengths), _p(dout), _p(lse), _p(dvec), _p(dqkv), B, H, T, dh,
                                         scale, int(use_mask), *_window(chunk_size, history_size), _dt(qkv), _stream()), "relattn_fused_bwd_k")
    return dqkv


# -------------------------------------------------------------------------------------- LSTM
def lstm_step_fwd(xg_t, hr, h_prev, c_prev, lengths, t, gates_t, c_out, h_out, y_out, B, P):
    check(_L().tfasr_lstm_step_fwd(
        _pv(xg_t), xg_t.stride(0), _pv(hr), _pv(h_prev), 0 if h_prev is None else h_prev.stride(0), _pv(c_prev),
        0 if c_prev is None else c_prev.stride(0), _pv(lengths), t, _pv(gates_t), 0 if gates_t is None else gates_t.stride(0),
        _pv(c_out), c_out.stride(0), _pv(h_out), h_out.stride(0), _pv(y_out), 0 if y_out is None else y_out.stride(0), B, P,
        _dt(xg_t), _stream()), "lstm_step_fwd")


def lstm_step_bwd(dy_t, dhr, dh_carry, dc_carry, gates_t, c_t, c_prev, lengths, t, dz_t, B, P):
    check(_L().tfasr_lstm_step_bwd(
        _pv(dy_t), dy_t.stride(0), _pv(dhr), _pv(dh_carry), _pv(dc_carry), _pv(gates_t), gates_t.stride(0), _pv(c_t), c_t.stride(0),
        _pv(c_prev), 0 if c_prev is None else c_prev.stride(0), _pv(lengths), t, _pv(dz_t), dz_t.stride(0), B, P, _dt(dy_t),
        _stream()), "lstm_step_bwd")


def lstm_set_persist(mode):
    """-1: TFASR_LSTM_PERSIST from the environment, 0: per-step kernels, 1: persistent kernels; returns the previous override."""
    return int(_L().tfasr_lstm_set_persist(int(mode)))


def lstm_seq_fwd(xg, rk, h0, c0, lengths, gates, cseq, hseq, yseq, hr):
    """All U1 steps of the recurrence in one host call (contiguous [B,U1,*] buffers)."""
    B, U1, P4 = xg.shape
    P = P4 // 4
    assert xg.is_contiguous() and gates.is_contiguous() and cseq.is_contiguous() and hseq.is_contiguous() and (yseq is None or yseq.is_contiguous())
    check(_L().tfasr_lstm_seq_fwd(_p(xg), _p(rk), _pv(h0), 0 if h0 is None else h0.stride(0), _pv(c0), 0 if c0 is None else c0.stride(0), _pv(lengths),
                                  _p(gates), _p(cseq), _p(hseq), _pv(yseq), _p(hr), B, U1, P, _dt(xg), _stream()), "lstm_seq_fwd")


def lstm_seq_fwd_range(xg, rk, h0, c0, lengths, gates, cseq, hseq, yseq, hr, t0, t1):
    """steps [t0, t1) of lstm_seq_fwd with the per-step kernels (tfasr_lstm_seq_fwd_range)"""
    B, U1, P4 = xg.shape
    P = P4 // 4
    check(_L().tfasr_lstm_seq_fwd_range(_p(xg), _p(rk), _pv(h0), 0 if h0 is None else h0.stride(0), _pv(c0), 0 if c0 is None else c0.stride(0), _pv(lengths),
                                        _p(gates), _p(cseq), _p(hseq), _pv(yseq), _p(hr), B, U1, P, _dt(xg), int(t0), int(t1), _stream()),
          "lstm_seq_fwd_range")


def lstm_seq_bwd_range(dy, rk, gates, cseq, lengths, dz, dh_carry, dc_carry, dhr, t0, t1):
    """steps t1-1 .. t0 of lstm_seq_bwd with the per-step kernels (slices in descending order)"""
    B, U1, P = dy.shape
    check(_L().tfasr_lstm_seq_bwd_range(_p(dy), _p(rk), _p(gates), _p(cseq), _pv(lengths), _p(dz), _p(dh_carry), _p(dc_carry), _p(dhr), B, U1, P, _dt(dy),
                                        int(t0), int(t1), _stream()), "lstm_seq_bwd_range")


def lstm_persist_sync(device):
    """64-byte synchronisation record of the persistent LSTM kernels; word [1] != 0 after a sync = a wait timed out"""
    return torch.zeros(int(_L().tfasr_lstm_persist_sync_bytes()) // 4, dtype=torch.int32, device=device)


def lstm_persist_fwd(xg, rk, h0, c0, lengths, gates, cseq, hseq, yseq, sync):
    """tfasr_lstm_persist_fwd: the whole forward recurrence as ONE launch (raises TfasrUnsupported outside its shape range)."""
    B, U1, P4 = xg.shape
    P = P4 // 4
    assert xg.is_contiguous() and gates.is_contiguous() and cseq.is_contiguous() and hseq.is_contiguous() and (yseq is None or yseq.is_contiguous())
    check(_L().tfasr_lstm_persist_fwd(_p(xg), _p(rk), _pv(h0), 0 if h0 is None else h0.stride(0), _pv(c0), 0 if c0 is None else c0.stride(0), _pv(lengths),
                                      _p(gates), _p(cseq), _p(hseq), _pv(yseq), B, U1, P, _dt(xg), _p(sync), _stream()), "lstm_persist_fwd")


def lstm_persist_bwd(dy, rk, gates, cseq, lengths, dz, dh_carry, dc_carry, sync):
    B, U1, P = dy.shape
    assert dy.is_contiguous() and gates.is_contiguous() and cseq.is_contiguous() and dz.is_contiguous()
    check(_L().tfasr_lstm_persist_bwd(_p(dy), _p(rk), _p(gates), _p(cseq), _pv(lengths), _p(dz), _p(dh_carry), _p(dc_carry), B, U1, P, _dt(dy),
                                      _p(sync), _stream()), "lstm_persist_bwd")


def lstm_seq_bwd(dy, rk, gates, cseq, lengths, dz, dh_carry, dc_carry, dhr):
    B, U1, P = dy.shape
    assert dy.is_contiguous() and gates.is_contiguous() and cseq.is_contiguous() and dz.is_contiguous()
    check(_L().tfasr_lstm_seq_bwd(_p(dy), _p(rk), _p(gates), _p(cseq), _pv(lengths), _p(dz), _p(dh_carry), _p(dc_carry), _p(dhr), B, U1, P, _dt(dy),
                                  _stream()), "lstm_seq_bwd")


# --------------------------------------------------------------------------------- subsampling
def conv1_fwd(x, w, bias):
    B, T0, F0 = x.shape[:3]
    C = w.shape[-1]
    y = torch.empty(B, (T0 + 1) // 2, (F0 + 1) // 2, C, dtype=x.dtype, device=x.device)
    check(_L().tfasr_conv1_fwd(_p(x), _p(w), _p(bias), _p(y), B, T0, F0, C, _dt(x), _stream()), "conv1_fwd")
    return y


def conv1_bwd_weight(x, dy, dw, db):
    B, T0, F0 = x.shape[:3]
    C = dy.shape[-1]
    check(_L().tfasr_conv1_bwd_weight(_p(x), _p(dy), _p(dw), _p(db), B, T0, F0, C, _dt(x), _stream()), "conv1_bwd_weight")


# haloed space-to-depth ("S") layout helpers (include/tfasr_hip.h): y / dy are [B, T2+1, F2+1, 4, C] buffers
def conv1_fwd_s2d(x, w, bias, y):
    B, T0, F0 = x.shape[:3]
    check(_L().tfasr_conv1_fwd_s2d(_p(x), _p(w), _p(bias), _p(y), B, T0, F0, w.shape[-1], _dt(x), _stream()), "conv1_fwd_s2d")
    return y


def conv1_bwd_weight_s2d(x, dy, dw, db, C):
    B, T0, F0 = x.shape[:3]
    check(_L().tfasr_conv1_bwd_weight_s2d(_p(x), _p(dy), _p(dw), _p(db), B, T0, F0, C, _dt(x), _stream()), "conv1_bwd_weight_s2d")


def conv1_stats(x, w, bias, stats):
    B, T0, F0 = x.shape[:3]
    check(_L().tfasr_conv1_stats(_p(x), _p(w), _p(bias), _p(stats), B, T0, F0, w.shape[-1], _dt(x), _stream()), "conv1_stats")


def conv1_bn_apply_s2d(x, w, bias, fin, y):
    B, T0, F0 = x.shape[:3]
    check(_L().tfasr_conv1_bn_apply_s2d(_p(x), _p(w), _p(bias), _p(fin), _p(y), B, T0, F0, w.shape[-1], _dt(x), _stream()), "conv1_bn_apply")
    return y


def conv1_bn_bwd_stats_s2d(x, w, bias, fin, dy, bstats):
    B, T0, F0 = x.shape[:3]
    check(_L().tfasr_conv1_bn_bwd_stats_s2d(_p(x), _p(w), _p(bias), _p(fin), _p(dy), _p(bstats), B, T0, F0, w.shape[-1], _dt(x), _stream()),
          "conv1_bn_bwd_stats")


def conv1_bn_bwd_apply_s2d(x, w, bias, fin, bstats, count, dy, dw, db):
    B, T0, F0 = x.shape[:3]
    check(_L().tfasr_conv1_bn_bwd_apply_s2d(_p(x), _p(w), _p(bias), _p(fin), _p(bstats), float(count), _p(dy), _p(dw), _p(db), B, T0, F0,
                                            w.shape[-1], _dt(x), _stream()), "conv1_bn_bwd_apply")


CONV1_GRAM_DOUBLES = 768  # TFASR_CONV1_GRAM_DOUBLES (include/tfasr_hip.h)


def conv1_gram(x, gram):
    """gram: CONV1_GRAM_DOUBLES f64 (8 partial copies, stride 96, of G[9][9], s[9], N) of conv1's 3x3 patches over every output position
    of the feature map x [B, T0, F0]."""
    B, T0, F0 = x.shape[:3]
    assert gram.dtype == torch.float64 and gram.numel() >= CONV1_GRAM_DOUBLES
    check(_L().tfasr_conv1_gram(_p(x), _p(gram), B, T0, F0, _dt(x), _stream()), "conv1_gram")
    return gram


def conv1_stats_from_gram(gram, w, bias, stats):
    check(_L().tfasr_conv1_stats_from_gram(_p(gram), _p(w), _p(bias), _p(stats), w.shape[-1], _stream()), "conv1_stats_from_gram")


def conv1_bn_bwd_onepass_s2d(x, w, bias, fin, dy, bstats, pbuf):
    B, T0, F0 = x.shape[:3]
    check(_L().tfasr_conv1_bn_bwd_onepass_s2d(_p(x), _p(w), _p(bias), _p(fin), _p(dy), _p(bstats), _p(pbuf), B, T0, F0, w.shape[-1], _dt(x),
                                              _stream()), "conv1_bn_bwd_onepass")


def conv1_bn_bwd_finalize(gram, w, bias, fin, bstats, count, pbuf, dw, db):
    check(_L().tfasr_conv1_bn_bwd_finalize(_p(gram), _p(w), _p(bias), _p(fin), _p(bstats), float(count), _p(pbuf), _p(dw), _p(db), w.shape[-1],
                                           _stream()), "conv1_bn_bwd_finalize")


def halo_zero(x, B, T2, F2, W):
    check(_L().tfasr_halo_zero(_p(x), B, T2, F2, W, _dt(x), _stream()), "halo_zero")
    return x


def s2d_edge_zero(x, B, T1, F1, C):
    check(_L().tfasr_s2d_edge_zero(_p(x), B, T1, F1, C, _dt(x), _stream()), "s2d_edge_zero")
    return x


def im2col_3x3s2(x, col=None):
    B, T1, F1, C = x.shape
    T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
    if col is None:
        col = torch.empty(B * T2 * F2, 9 * C, dtype=x.dtype, device=x.device)
    check(_L().tfasr_im2col_3x3s2(_p(x), _p(col), B, T1, F1, C, _dt(x), _stream()), "im2col")
    return col


def col2im_3x3s2(dcol, B, T1, F1, C):
    dx = torch.empty(B, T1, F1, C, dtype=dcol.dtype, device=dcol.device)
    check(_L().tfasr_col2im_3x3s2(_p(dcol), _p(dx), B, T1, F1, C, _dt(dcol), _stream()), "col2im")
    return dx


# ------------------------------------------------------------------------------------ frontend
def logmel(signal, window, melw, band, frame_step, nfft, preemph, eps, out_dtype):
    B, N = signal.shape
    T0 = -(-N // frame_step)
    F = melw.shape[1]
    out = torch.empty(B, T0, F, dtype=out_dtype, device=signal.device)
    assert signal.dtype == torch.float32
    check(_L().tfasr_logmel(_p(signal), B, N, preemph, _p(window), window.numel(), frame_step, nfft, _p(melw), _p(band), F,
                            eps, _p(out), T0, _dt(out), _stream()), "logmel")
    return out


# -------------------------------------------------------------------------------------- greedy search
def decode_prepare(encj, nframes, frame_idx, tok_idx, active, ecur, max_tokens, mode):
    B, T, J = encj.shape
    check(_L().tfasr_decode_prepare(_p(encj), _p(nframes), _p(frame_idx), _p(tok_idx), _p(active), _p(ecur), B, T, J, max_tokens,
                                    mode, _dt(encj), _stream()), "decode_prepare")


def decode_pack(emb, lstm_k, lstm_rk, wjp, wv):
    """The f32 weights of a search step in the MFMA kernels' tile order plus G = emb @ lstm_k (csrc/decode_step.hip); None when the shapes
    have no MFMA route (the step then runs on the row-major masters).  Made once per recognize call: the weights are constants of the search."""
    V, E = emb.shape
    P = lstm_rk.shape[0]
    J = wv.shape[0]
    n = int(_L().tfasr_decode_pack_floats(E, P, J, V))
    if n == 0:
        return None
    packed = torch.empty(n, dtype=torch.float32, device=lstm_k.device)
    st = _L().tfasr_decode_pack(_p(emb), _p(lstm_k), _p(lstm_rk), _p(wjp), _p(wv), _p(packed), E, P, J, V, _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return None
    check(st, "decode_pack")
    return packed


def decode_step(emb, lstm_k, lstm_rk, lstm_b, ln_g, ln_b, wjp, bjp, wv, bv, encj, nframes, frame_idx, tok_idx, prev_tok, h, c, active, h_new, c_new,
                z, logits, max_tokens, mode, ln_eps=1e-3, packed=None):
    """Fused search step (three launches up to the f32 logits); returns False when the shapes need the per-op route."""
    B, T, J = encj.shape
    V, E = emb.shape
    P = h.shape[1]
    st = _L().tfasr_decode_step(_p(emb), _p(lstm_k), _p(lstm_rk), _p(lstm_b), _p(ln_g), _p(ln_b), _p(wjp), _p(bjp), _p(wv), _p(bv), _p(packed),
                                _p(encj), _p(nframes), _p(frame_idx), _p(tok_idx), _p(prev_tok), _p(h), _p(c), _p(active), _p(h_new), _p(c_new),
                                _p(z), _p(logits), B, T, E, P, J, V, max_tokens, mode, ln_eps, _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return False
    check(st, "decode_step")
    return True


def decode_steps(emb, lstm_k, lstm_rk, lstm_b, ln_g, ln_b, wjp, bjp, wv, bv, encj, nframes, frame_idx, tok_idx, prev_tok, h, c, active, h_new, c_new,
                 z, logits, tokens, per_frame, max_tokens, blank, mode, max_tokens_per_frame, iters, ln_eps=1e-3, packed=None):
    """`iters` fused search iterations from one host call; False when the shapes need the per-op route."""
    B, T, J = encj.shape
    V, E = emb.shape
    P = h.shape[1]
    st = _L().tfasr_decode_steps(_p(emb), _p(lstm_k), _p(lstm_rk), _p(lstm_b), _p(ln_g), _p(ln_b), _p(wjp), _p(bjp), _p(wv), _p(bv), _p(packed),
                                 _p(encj), _p(nframes), _p(frame_idx), _p(tok_idx), _p(prev_tok), _p(h), _p(c), _p(active), _p(h_new), _p(c_new),
                                 _p(z), _p(logits), _p(tokens), _p(per_frame), B, T, E, P, J, V, max_tokens, blank, mode, max_tokens_per_frame,
                                 ln_eps, int(iters), _stream())
    if st == _lib.STATUS_UNSUPPORTED:
        return False
    check(st, "decode_steps")
    return True


def decode_update(logits, active, nframes, frame_idx, prev_tok, tok_idx, tokens, per_frame, h_new, c_new, h, c, max_tokens,
                  blank, mode, max_tokens_per_frame):
    B, V = logits.shape
    P = h.shape[1]
    check(_L().tfasr_decode_update(_p(logits), _p(active), _p(nframes), _p(frame_idx), _p(prev_tok), _p(tok_idx), _p(tokens),
                                   _p(per_frame), _p(h_new), _p(c_new), _p(h), _p(c), B, V, P, max_tokens, blank, mode,
                                   max_tokens_per_frame, _dt(logits), _stream()), "decode_update")


# ------------------------------------------------------------------------------------------------ CTC
def ctc_loss_fwd_bwd(logits, labels, label_len, logit_len, grad_scale=None, grads=None, want_grads=True, blank=0):
    """logits [B,T,V] -> (costs [B], grads or None); tf.nn.ctc_loss semantics (losses/ctc_loss.py:57-66)."""
    B, T, V = logits.shape
    U = labels.shape[1]
    # kernel limits (one lattice state per thread, per-class occupancy in LDS): say so instead of a bare INVALID_VALUE
    if 2 * U + 1 > 1024:
        raise ValueError(f"tfasr_ctc_loss: padded label length {U} > 511 (2U+1 lattice states must fit one 1024-thread workgroup); crop the label padding")
    if V * 4 > 64 * 1024:
        raise ValueError(f"tfasr_ctc_loss: vocabulary {V} > 16384 classes (the per-class occupancy buffer of the gradient kernel lives in LDS)")
    costs = torch.empty(B, dtype=torch.float32, device=logits.device)
    if want_grads and grads is None:
        grads = torch.empty_like(logits)
    n = ctypes.c_size_t(0)
    check(_L().tfasr_ctc_loss_workspace_size(B, T, U, V, ctypes.byref(n)), "ctc_ws")
    ws = workspace(n.value, logits.device, "ctc")
    check(_L().tfasr_ctc_loss(_p(logits), _p(grads) if want_grads else None, _p(labels), _p(label_len), _p(logit_len), _p(grad_scale),
                              B, T, U, V, blank, _dt(logits), _p(costs), _p(ws), ws.numel(), _stream()), "ctc_loss")
    return costs, (grads if want_grads else None)


def ctc_beam_search(logits, logit_len, beam_width=10, blank_index=None):
    """tf.nn.ctc_beam_search_decoder semantics (top path, dense, 0 padded).  Host routine (as in the reference): the logits
    are copied to host memory.  blank_index=None reproduces the reference call (TF convention: the LAST class is blank)."""
    import numpy as np

    x = logits.detach().float().cpu().contiguous().numpy()
    B, T, V = x.shape
    ln = np.ascontiguousarray(logit_len.detach().cpu().numpy() if hasattr(logit_len, "detach") else logit_len, dtype=np.int32)
    toks = np.zeros((B, T), np.int32)
    n = np.zeros(B, np.int32)
    lp = np.zeros(B, np.float32)
    bi = V - 1 if blank_index is None else int(blank_index)
    check(_L().tfasr_ctc_beam_search_host(x.ctypes.data, ln.ctypes.data, B, T, V, int(beam_width), bi, toks.ctypes.data, n.ctypes.data,
                                          lp.ctypes.data), "ctc_beam_search")
    return torch.from_numpy(toks), torch.from_numpy(n), torch.from_numpy(lp)


def ctc_greedy_decode(logits, logit_len, blank=0):
    B, T, V = logits.shape
    am = torch.empty(B * T, dtype=torch.int32, device=logits.device)
    tokens = torch.empty(B, T, dtype=torch.int32, device=logits.device)
    tlen = torch.empty(B, dtype=torch.int32, device=logits.device)
    check(_L().tfasr_ctc_greedy_decode(_p(logits), _p(logit_len), _p(am), _p(tokens), _p(tlen), B, T, V, blank, _dt(logits), _stream()), "ctc_greedy")
    return tokens, tlen


# --------------------------------------------------------------------------------- native Conformer-block executor
def block_ctx():
    return ctypes.create_string_buffer(int(_L().tfasr_block_ctx_bytes()))


def block_workspace_sizes(cfg):
    a, b, c = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    check(_L().tfasr_block_workspace_sizes(ctypes.byref(cfg), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "block_workspace_sizes")
    return a.value, b.value, c.value


def block_fwd(cfg, params, io, ctx, phase):
    check(_L().tfasr_block_fwd(ctypes.byref(cfg), ctypes.byref(params), ctypes.byref(io), ctx, phase, _stream()), "block_fwd")


def block_bwd(cfg, params, io, ctx, phase):
    check(_L().tfasr_block_bwd(ctypes.byref(cfg), ctypes.byref(params), ctypes.byref(io), ctx, phase, _stream()), "block_bwd")


def launch_count():
    """kernel launches the library has queued since it was loaded (tfasr_launch_count)"""
    return int(_L().tfasr_launch_count())


def block_side_stream():
    """the block executor's low-priority second stream (tfasr_block_side_stream) as a raw hipStream_t value"""
    ptr = ctypes.c_void_p(0)
    check(_L().tfasr_block_side_stream(ctypes.byref(ptr)), "block_side_stream")
    return int(ptr.value)


def block_wgrad_probe(enable):
    check(_L().tfasr_block_wgrad_probe(int(bool(enable))), "block_wgrad_probe")


def block_wgrad_probe_read():
    """(summed milliseconds, launches) of the grouped weight-gradient launches recorded since the probe was enabled / last read"""
    ms, n = ctypes.c_float(0.0), ctypes.c_int(0)
    check(_L().tfasr_block_wgrad_probe_read(ctypes.byref(ms), ctypes.byref(n)), "block_wgrad_probe_read")
    return float(ms.value), int(n.value)


def block_bwd_left(ctx):
    """bit mask of what the last block_bwd with this ctx left to the caller (tfasr_block_bwd_left)"""
    return int(_L().tfasr_block_bwd_left(ctx))


def block_wgrad_join(slot_mask=3):
    """The current stream waits for the grouped weight-gradient launches queued on the executor's second stream (tfasr_block_io.wgrad_slot)."""
    check(_L().tfasr_block_wgrad_join(int(slot_mask), _stream()), "block_wgrad_join")


def block_ln_fold_all(ctxs, d):
    """One launch for the LayerNorm gamma / beta gradients of every block whose backward ran with io.ln_part_ext (tfasr_block_ln_fold_all)."""
    arr = (ctypes.c_void_p * len(ctxs))(*[ctypes.addressof(c) for c in ctxs])
    check(_L().tfasr_block_ln_fold_all(arr, len(ctxs), int(d), _stream()), "block_ln_fold_all")


def block_dwconv_wgrad_all(cfg, params, ctxs, dcvs, device):
    """Depthwise-conv weight / bias gradients of every block whose backward ran with io.dcv_keep: one launch pair (tfasr_block_dwconv_wgrad_all)."""
    need = ctypes.c_size_t(0)
    check(_L().tfasr_dwconv_bwd_weight_workspace_size(cfg.B, cfg.T, cfg.d, cfg.ksize, ctypes.byref(need)), "dwconv_bwd_weight_workspace_size")
    for b0 in range(0, len(ctxs), 32):  # (at most 32 blocks per launch pair: deeper encoders take several)
        n = min(32, len(ctxs) - b0)
        ws = workspace(n * need.value, device, "dw_wgrad_all")
        pa = (ctypes.c_void_p * n)(*[ctypes.addressof(p) for p in params[b0:b0 + n]])
        ca = (ctypes.c_void_p * n)(*[ctypes.addressof(c) for c in ctxs[b0:b0 + n]])
        da = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dcvs[b0:b0 + n]])
        check(_L().tfasr_block_dwconv_wgrad_all(ctypes.byref(cfg), pa, ca, da, n, _p(ws), ws.numel(), _stream()), "block_dwconv_wgrad_all")


def layernorm_bwd_part_blocks(rows, C, dtype):
    return int(_L().tfasr_layernorm_bwd_part_blocks(int(rows), int(C), {torch.float32: 0, torch.bfloat16: 1}[dtype] if not isinstance(dtype, int) else dtype))


# --------------------------------------------------------------------------------- ContextNet pieces
def rows_subsample_fwd(x, stride):
    B, T, C = x.shape
    y = torch.empty(B, -(-T // stride), C, dtype=x.dtype, device=x.device)
    check(_L().tfasr_rows_subsample_fwd(_p(x), _p(y), B, T, C, stride, _dt(x), _stream()), "rows_subsample_fwd")
    return y


def rows_subsample_bwd(dy, T, stride):
    B, T2, C = dy.shape
    dx = torch.empty(B, T, C, dtype=dy.dtype, device=dy.device)
    check(_L().tfasr_rows_subsample_bwd(_p(dy), _p(dx), B, T, C, stride, _dt(dy), _stream()), "rows_subsample_bwd")
    return dx


def se_pool(x, lengths):
    B, T, C = x.shape
    pool = torch.empty(B, C, dtype=torch.float32, device=x.device)
    check(_L().tfasr_se_pool(_p(x), _p(lengths), _p(pool), B, T, C, _dt(x), _stream()), "se_pool")
    return pool


def se_scale_fwd(x, scale):
    B, T, C = x.shape
    y = torch.empty_like(x)
    check(_L().tfasr_se_scale_fwd(_p(x), _p(scale), _p(y), B, T, C, _dt(x), _stream()), "se_scale_fwd")
    return y


def se_scale_bwd_reduce(x, dy):
    B, T, C = x.shape
    ds = torch.empty(B, C, dtype=torch.float32, device=x.device)
    check(_L().tfasr_se_scale_bwd_reduce(_p(x), _p(dy), _p(ds), B, T, C, _dt(x), _stream()), "se_scale_bwd_reduce")
    return ds


def se_bwd_apply(dy, scale, dpool, lengths):
    B, T, C = dy.shape
    dx = torch.empty_like(dy)
    check(_L().tfasr_se_bwd_apply(_p(dy), _p(scale), _p(dpool), _p(lengths), _p(dx), B, T, C, _dt(dy), _stream()), "se_bwd_apply")
    return dx


def add_act_fwd(a, b, act=ACT_NONE):
    y = torch.empty_like(a)
    check(_L().tfasr_add_act_fwd(_p(a), _p(b), _p(y), a.numel(), act, _dt(a), _stream()), "add_act_fwd")
    return y


def add_act_bwd(a, b, dy, act=ACT_NONE):
    d = torch.empty_like(a)
    check(_L().tfasr_add_act_bwd(_p(a), _p(b), _p(dy), _p(d), a.numel(), act, _dt(a), _stream()), "add_act_bwd")
    return d
