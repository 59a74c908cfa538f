"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the reference RNN-T loss + gradient:
    tensorflow_asr/losses/impl/rnnt.py:181-278  compute_rnnt_loss_and_grad_helper
    tensorflow_asr/losses/base_loss.py:28-37    BaseLoss.call (length clamp)
    tensorflow_asr/losses/rnnt_loss.py:50-61    RnntLoss.call (+ Keras sum_over_batch_size mean)

Pinning status: the reference ships NO golden vector for this path (tests/test_rnnt_loss.py has no
assertion).  This restatement is pinned instead by
  (a) tests/golden/rnnt_reference_*.npz — outputs of the reference's OWN rnnt.py source executed in this
      container over oracle/tf_shim (a NumPy stand-in for the tf.* primitives it calls; generator:
      oracle/gen_golden_from_reference.py), and
  (b) brute-force enumeration of all alignments (rnnt_loss_bruteforce below) + finite differences.
TensorFlow itself is not installable here, so (a) pins the reference's Python logic (anti-diagonal
packing, masks, closed-form gradient) but not TF's C++ kernels: "parity pinned to reference source
over a NumPy tf-shim".
"""
import itertools

import numpy as np


def log_softmax(x):
    """tf.nn.log_softmax over the last axis (impl/rnnt.py:211)."""
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))


def _logaddexp(a, b):
    """2-term log-sum-exp with -inf handling (impl/rnnt.py:72-78,126)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        m = np.maximum(a, b)
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), -np.inf, r)


def transition_probs(logits, labels):
    """blank[b,t,u] = lp[b,t,u,0]; truth[b,t,u] = lp[b,t,u,labels[b,u]]  (impl/rnnt.py:94-105)."""
    lp = log_softmax(logits)
    B, T, U1, V = lp.shape
    blank = lp[..., 0]
    idx = np.broadcast_to(labels[:, None, :, None].astype(np.int64), (B, T, U1 - 1, 1))
    truth = np.take_along_axis(lp[:, :, :-1, :], idx, axis=-1)[..., 0]
    return lp, blank, truth


def clamp_lengths(logit_len, label_len):
    """BaseLoss.call: logit_length = max(logit_length, label_length)  (losses/base_loss.py:36)."""
    logit_len = np.asarray(logit_len, np.int32)
    label_len = np.asarray(label_len, np.int32)
    return np.where(logit_len < label_len, label_len, logit_len), label_len


def alpha_beta(blank, truth, label_len, logit_len):
    """alpha over the padded lattice (forward_dp :108-137), beta per sample from the terminal node
    (backward_dp :140-178). Values outside t<Tl,u<=Ul are 0 after masking (:224,:229-231)."""
    B, T, U1 = blank.shape
    dt = blank.dtype
    alpha = np.full((B, T, U1), -np.inf, dt)
    alpha[:, 0, 0] = 0.0
    for n in range(1, T + U1 - 1):
        ts = np.arange(max(0, n - U1 + 1), min(n, T - 1) + 1)
        us = n - ts
        xb = np.full((B, len(ts)), -np.inf, dt)
        xt = np.full((B, len(ts)), -np.inf, dt)
        mb = ts > 0
        xb[:, mb] = alpha[:, ts[mb] - 1, us[mb]] + blank[:, ts[mb] - 1, us[mb]]
        mt = us > 0
        xt[:, mt] = alpha[:, ts[mt], us[mt] - 1] + truth[:, ts[mt], us[mt] - 1]
        alpha[:, ts, us] = _logaddexp(xb, xt)
    beta = np.full((B, T, U1), -np.inf, dt)
    for b in range(B):
        Tl, Ul = int(logit_len[b]), int(label_len[b])
        beta[b, Tl - 1, Ul] = blank[b, Tl - 1, Ul]
        for n in range(Tl + Ul - 2, -1, -1):
            ts = np.arange(max(0, n - Ul), min(n, Tl - 1) + 1)
            us = n - ts
            xb = np.full(len(ts), -np.inf, dt)
            xt = np.full(len(ts), -np.inf, dt)
            mb = ts + 1 < Tl
            xb[mb] = beta[b, ts[mb] + 1, us[mb]] + blank[b, ts[mb], us[mb]]
            mt = us < Ul
            xt[mt] = beta[b, ts[mt], us[mt] + 1] + truth[b, ts[mt], us[mt]]
            beta[b, ts, us] = _logaddexp(xb, xt)
    tmask = np.arange(T)[None, :, None] < np.asarray(logit_len)[:, None, None]
    umask = np.arange(U1)[None, None, :] <= np.asarray(label_len)[:, None, None]
    mask = tmask & umask
    alpha = np.where(mask, alpha, 0.0)
    beta = np.where(mask, beta, 0.0)
    return alpha, beta


def rnnt_loss_and_grad(logits, labels, label_len, logit_len, dtype=np.float64):
    """Returns (loss [B], dloss/dlogits [B,T,U1,V]); lengths are used as given (clamp first)."""
    x = np.asarray(logits).astype(dtype)
    labels = np.asarray(labels)
    B, T, U1, V = x.shape
    lp, blank, truth = transition_probs(x, labels)
    alpha, beta = alpha_beta(blank, truth, label_len, logit_len)
    b00 = beta[:, 0, 0].copy()
    loss = -b00
    tl = np.asarray(logit_len)[:, None, None]
    ul = np.asarray(label_len)[:, None, None]
    tt = np.arange(T)[None, :, None]
    uu = np.arange(U1)[None, None, :]
    # grads wrt blank log-probs (:233-248)
    gb = np.zeros((B, T, U1), dtype)
    mb = (tt[:, :-1] < tl - 1) & (uu <= ul)
    with np.errstate(over="ignore", invalid="ignore"):
        e = alpha[:, :-1, :] + beta[:, 1:, :] - b00[:, None, None] + blank[:, :-1, :]
        gb[:, :-1, :] = np.where(mb, -np.exp(np.where(mb, e, 0.0)), 0.0)
    for b in range(B):
        gb[b, int(logit_len[b]) - 1, int(label_len[b])] += -1.0
    # grads wrt truth log-probs (:251-254)
    mt = (tt < tl) & (uu[:, :, :-1] < ul)
    with np.errstate(over="ignore", invalid="ignore"):
        e = alpha[:, :, :-1] + beta[:, :, 1:] - b00[:, None, None] + truth
        gt = np.where(mt, -np.exp(np.where(mt, e, 0.0)), 0.0)
    # scatter into [B,T,U1,V] then g - softmax * sum(g)   (:256-275)
    grads = np.zeros((B, T, U1, V), dtype)
    grads[..., 0] = gb
    idx = np.broadcast_to(labels[:, None, :, None].astype(np.int64), (B, T, U1 - 1, 1))
    cur = np.take_along_axis(grads[:, :, :-1, :], idx, axis=-1)
    np.put_along_axis(grads[:, :, :-1, :], idx, cur + gt[..., None], axis=-1)
    grads = grads - np.exp(lp) * grads.sum(axis=-1, keepdims=True)
    return loss, grads


def rnnt_loss_keras_mean(logits, labels, label_len, logit_len, dtype=np.float64):
    """RnntLoss.__call__: clamp + per-sample loss + sum_over_batch_size mean (rnnt_loss.py:34,50-61)."""
    tl, ul = clamp_lengths(logit_len, label_len)
    loss, grads = rnnt_loss_and_grad(logits, labels, ul, tl, dtype)
    return loss.mean(), grads / loss.shape[0], loss


def rnnt_loss_bruteforce(logits, labels, Tl, Ul):
    """-log sum over ALL monotonic alignments (single sample, tiny shapes): independent definition."""
    lp = log_softmax(np.asarray(logits, np.float64))
    total = -np.inf
    # an alignment = positions of the Ul label emissions among Tl+Ul steps, path must end with blank at (Tl-1, Ul)
    for emit in itertools.combinations(range(Tl + Ul - 1), Ul):
        t = u = 0
        s = 0.0
        emit = set(emit)
        ok = True
        for step in range(Tl + Ul - 1):
            if step in emit:
                s += lp[t, u, labels[u]]
                u += 1
            else:
                s += lp[t, u, 0]
                t += 1
                if t >= Tl:
                    ok = False
                    break
        if not ok or u != Ul or t != Tl - 1:
            continue
        s += lp[Tl - 1, Ul, 0]
        total = np.logaddexp(total, s)
    return -total
