"""CPU: the RNN-T oracle against (a) the reference-source goldens, (b) brute force, (c) finite differences."""
import glob
import os

import numpy as np
import pytest

from oracle import rnnt_ref


def _goldens(golden_dir):
    fs = sorted(glob.glob(os.path.join(golden_dir, "rnnt_reference_*.npz")))
    assert fs, "golden fixtures missing"
    return fs


def test_oracle_matches_reference_source_goldens(golden_dir):
    for f in _goldens(golden_dir):
        g = np.load(f)
        loss, grads = rnnt_ref.rnnt_loss_and_grad(g["logits"], g["labels"], g["label_len"], g["logit_len"], np.float64)
        np.testing.assert_allclose(loss, g["loss"], rtol=2e-6, atol=1e-5, err_msg=f)
        np.testing.assert_allclose(grads, g["grads"], rtol=2e-5, atol=2e-5, err_msg=f)  # goldens are f32 runs


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_matches_bruteforce(seed):
    rng = np.random.default_rng(seed)
    T, U, V = 4, 3, 4
    logits = rng.standard_normal((1, T, U + 1, V))
    labels = rng.integers(1, V, (1, U)).astype(np.int32)
    for Tl in (1, 3, 4):
        for Ul in (0, 2, 3):
            loss, _ = rnnt_ref.rnnt_loss_and_grad(logits, labels, np.array([Ul]), np.array([Tl]))
            bf = rnnt_ref.rnnt_loss_bruteforce(logits[0], labels[0], Tl, Ul)
            assert abs(loss[0] - bf) < 1e-9


def test_oracle_gradient_finite_difference():
    rng = np.random.default_rng(5)
    B, T, U, V = 2, 4, 2, 5
    logits = rng.standard_normal((B, T, U + 1, V))
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    tl, ul = np.array([4, 3]), np.array([2, 1])
    loss, grads = rnnt_ref.rnnt_loss_and_grad(logits, labels, ul, tl)
    eps = 1e-6
    for _ in range(40):
        idx = tuple(rng.integers(0, s) for s in logits.shape)
        lp = logits.copy(); lp[idx] += eps
        lm = logits.copy(); lm[idx] -= eps
        fd = (rnnt_ref.rnnt_loss_and_grad(lp, labels, ul, tl)[0][idx[0]] - rnnt_ref.rnnt_loss_and_grad(lm, labels, ul, tl)[0][idx[0]]) / (2 * eps)
        assert abs(fd - grads[idx]) < 1e-6
    # alpha-side and beta-side likelihoods agree
    lp_, blank, truth = rnnt_ref.transition_probs(logits, labels)
    alpha, beta = rnnt_ref.alpha_beta(blank, truth, ul, tl)
    for b in range(B):
        a_end = alpha[b, tl[b] - 1, ul[b]] + blank[b, tl[b] - 1, ul[b]]
        assert abs(a_end - beta[b, 0, 0]) < 1e-9


def test_clamp_lengths_like_base_loss():
    tl, ul = rnnt_ref.clamp_lengths([3, 9], [5, 2])
    assert tl.tolist() == [5, 9] and ul.tolist() == [5, 2]
