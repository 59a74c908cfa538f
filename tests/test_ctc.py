"""CTC loss / greedy decode: oracle vs brute force (CPU) and HIP kernels vs oracle (GPU)."""
import numpy as np
import pytest
import torch

from oracle import ctc_ref


def test_oracle_vs_bruteforce():
    rng = np.random.default_rng(0)
    T, V = 5, 3
    logits = rng.standard_normal((1, T, V))
    for labels in ([1], [1, 2], [2, 2], [1, 2, 1]):
        loss, _ = ctc_ref.ctc_loss_and_grad(logits, np.array([labels]), [len(labels)], [T])
        assert abs(loss[0] - ctc_ref.ctc_loss_bruteforce(logits[0], labels)) < 1e-9


def test_oracle_greedy():
    x = np.zeros((1, 6, 3))
    for t, c in enumerate([1, 1, 0, 1, 2, 2]):
        x[0, t, c] = 5
    out, n = ctc_ref.ctc_greedy_decode(x, [6])
    assert out[0, :3].tolist() == [1, 1, 2] and n[0] == 3
    out, n = ctc_ref.ctc_greedy_decode(x, [2])
    assert n[0] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 20, 6, 29), (4, 60, 25, 1000), (2, 9, 4, 5)])
def test_hip_ctc_vs_oracle(dev, dtype, shape):
    from tensorflowasr_amd import kernels as K

    B, T, U, V = shape
    rng = np.random.default_rng(T)
    logits = (rng.standard_normal((B, T, V)) * 2).astype(np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    labels[0, 1] = labels[0, 0]  # a repeated label (needs the blank in between)
    tl = rng.integers(2 * U + 1, T + 1, B).astype(np.int32) if T >= 2 * U + 1 else np.full(B, T, np.int32)
    tl[0] = T
    ul = rng.integers(1, U + 1, B).astype(np.int32)
    ul[0] = U
    xin = torch.from_numpy(logits).to(dtype)
    ref_loss, ref_g = ctc_ref.ctc_loss_and_grad(xin.float().numpy(), labels, ul, tl)
    scale = rng.uniform(0.5, 2, B).astype(np.float32)
    costs, grads = K.ctc_loss_fwd_bwd(xin.to(dev), torch.from_numpy(labels).to(dev), torch.from_numpy(ul).to(dev),
                                      torch.from_numpy(tl).to(dev), grad_scale=torch.from_numpy(scale).to(dev))
    torch.cuda.synchronize()
    np.testing.assert_allclose(costs.cpu().numpy(), ref_loss, rtol=1e-3 if dtype == torch.bfloat16 else 2e-5)  # north_star: CTC loss within 1e-3 relative
    want = ref_g * scale[:, None, None]
    for b in range(B):
        want[b, tl[b]:] = 0.0
    tol = dict(rtol=2e-2, atol=8e-3) if dtype == torch.bfloat16 else dict(rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(grads.float().cpu().numpy(), want, **tol)
    # greedy decode: identical tokens
    toks, n = K.ctc_greedy_decode(xin.to(dev), torch.from_numpy(tl).to(dev))
    rt, rn = ctc_ref.ctc_greedy_decode(xin.float().numpy(), tl)
    np.testing.assert_array_equal(toks.cpu().numpy(), rt)
    np.testing.assert_array_equal(n.cpu().numpy(), rn)


@pytest.mark.gpu
def test_loss_objects_reference_interface(dev):
    """RnntLoss / CtcLoss(y_true, y_pred) with the reference's clamp and sum_over_batch_size reduction."""
    from oracle import rnnt_ref
    from tensorflowasr_amd.losses import CtcLoss, RnntLoss, rnnt_loss
    from tensorflowasr_amd.schemas import TrainLabel, TrainOutput

    rng = np.random.default_rng(1)
    B, T, U, V = 3, 10, 4, 12
    logits = rng.standard_normal((B, T, U + 1, V)).astype(np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    tl, ul = np.array([10, 2, 7], np.int32), np.array([4, 3, 2], np.int32)  # sample 1: logit_len < label_len -> clamped
    y_pred = TrainOutput(torch.from_numpy(logits).to(dev), torch.from_numpy(tl))
    y_true = TrainLabel(torch.from_numpy(labels), torch.from_numpy(ul))
    ref_mean, ref_g, ref_loss = rnnt_ref.rnnt_loss_keras_mean(logits, labels, ul, tl)
    loss = RnntLoss(blank=0)(y_true, y_pred)
    np.testing.assert_allclose(float(loss), ref_mean, rtol=1e-5)
    costs, grads = RnntLoss(blank=0).call_with_grad(y_true, y_pred)
    np.testing.assert_allclose(costs.cpu().numpy(), ref_loss, rtol=1e-5)
    np.testing.assert_allclose(grads.cpu().numpy(), ref_g, rtol=5e-4, atol=2e-6)
    per = rnnt_loss(y_pred.logits, torch.from_numpy(np.maximum(tl, ul)).to(dev), torch.from_numpy(labels).to(dev), torch.from_numpy(ul).to(dev))
    np.testing.assert_allclose(per.cpu().numpy(), ref_loss, rtol=1e-5)
    with pytest.raises(AssertionError):
        RnntLoss(blank=1)
    clog = rng.standard_normal((B, 12, V)).astype(np.float32)
    cl, _ = ctc_ref.ctc_loss_and_grad(clog, labels, ul, [12, 12, 9])
    got = CtcLoss()(TrainLabel(torch.from_numpy(labels), torch.from_numpy(ul)), TrainOutput(torch.from_numpy(clog).to(dev), torch.tensor([12, 12, 9], dtype=torch.int32)))
    np.testing.assert_allclose(float(got), cl.mean(), rtol=2e-5)


def test_ctc_beam_search_host_routine_matches_oracle_and_bruteforce():
    """a25: tf.nn.ctc_beam_search_decoder semantics.  The library routine is host code (like the reference's op), so this runs
    without a GPU: vs the oracle's prefix beam search on random cases, and vs exhaustive enumeration when the beam is wide
    enough to be exact."""
    import ctypes

    from tensorflowasr_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(5)

    def run(x, lens, beam, blank):
        B, T, V = x.shape
        toks, n, lp = np.zeros((B, T), np.int32), np.zeros(B, np.int32), np.zeros(B, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        ln = np.asarray(lens, np.int32)
        st = lib.tfasr_ctc_beam_search_host(x.ctypes.data, ln.ctypes.data, B, T, V, beam, blank, toks.ctypes.data, n.ctypes.data, lp.ctypes.data)
        assert st == 0
        return toks, n, lp

    # exact against enumeration (T=5, V=4: 1024 alignments), both blank conventions
    for blank in (0, 3):
        x = rng.standard_normal((3, 5, 4)) * 2.0
        toks, n, lp = run(x, [5, 5, 4], 64, blank)
        for b, Tb in enumerate([5, 5, 4]):
            lab, p = ctc_ref.ctc_best_labelling_bruteforce(x[b, :Tb], blank)
            assert toks[b, :n[b]].tolist() == lab and abs(lp[b] - p) < 1e-4
            assert (toks[b, n[b]:] == 0).all()
    # narrow beams on longer inputs: same pruning decisions as the oracle
    x = rng.standard_normal((4, 30, 12)) * 3.0
    lens = [30, 17, 25, 1]
    for beam in (1, 4, 10):
        toks, n, lp = run(x, lens, beam, 11)
        for b, Tb in enumerate(lens):
            lab, p = ctc_ref.ctc_beam_search(x[b], Tb, beam, 11)
            assert toks[b, :n[b]].tolist() == lab, (beam, b)
            assert abs(lp[b] - p) < 1e-3
    # small alphabets + narrow beams + many frames: prefixes drop out of the beam and come back while their extensions stayed (ADVICE r02:
    # a label sequence must keep ONE trie node for the whole utterance, or its probability mass splits); 300 random utterances
    x = rng.standard_normal((300, 14, 3)) * 1.5
    for beam in (2, 3):
        toks, n, lp = run(x, [14] * 300, beam, 0)
        for b in range(300):
            lab, p = ctc_ref.ctc_beam_search(x[b], 14, beam, 0)
            assert toks[b, :n[b]].tolist() == lab, (beam, b)
            assert abs(lp[b] - p) < 1e-3
    # argument checking
    assert lib.tfasr_ctc_beam_search_host(None, None, 1, 1, 2, 1, 0, None, None, None) != 0
