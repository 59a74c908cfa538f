"""CPU: pin the layer-level oracle (oracle/conformer_ref.py) to the reference where the reference can be executed
(function bodies over the NumPy tf-shim -> tests/golden/*.npz) and to the reference's own truth tables."""
import os

import numpy as np
import torch

from oracle import conformer_ref as R


def test_streaming_mask_truth_tables_from_reference_tests(golden_dir):
    """tests/test_mask.py:6-55 of the reference (the only numeric goldens it ships)."""
    m = R.compute_streaming_mask(2, 2, 8)
    want = np.array([[1, 1, 0, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 0, 0, 0, 0], [1, 1, 1, 1, 0, 0, 0, 0],
                     [0, 0, 1, 1, 1, 1, 0, 0], [0, 0, 1, 1, 1, 1, 0, 0], [0, 0, 0, 0, 1, 1, 1, 1], [0, 0, 0, 0, 1, 1, 1, 1]], bool)
    assert (m[0] == want).all()
    m = R.compute_streaming_mask(3, 3, 14)
    rows = ["11100000000000"] * 3 + ["11111100000000"] * 3 + ["00011111100000"] * 3 + ["00000011111100"] * 3 + ["00000000011111"] * 2
    want = np.array([[c == "1" for c in r] for r in rows])
    assert (m[0] == want).all()
    g = np.load(os.path.join(golden_dir, "attention_reference.npz"))
    assert (R.compute_streaming_mask(2, 2, 8) == g["mask_2_2_8"]).all()
    assert (R.compute_streaming_mask(3, 3, 14) == g["mask_3_3_14"]).all()
    assert (R.compute_streaming_mask(4, -1, 10) == g["mask_4_m1_10"]).all()


def test_rel_left_shift_matches_reference_body_and_index_identity(golden_dir):
    g = np.load(os.path.join(golden_dir, "attention_reference.npz"))
    for T in (3, 5, 8):
        x = torch.from_numpy(g[f"shift_in_{T}"])
        y = R.rel_left_shift(x)
        np.testing.assert_array_equal(y.numpy(), g[f"shift_out_{T}"])
        # out[i, j] = x[i, T-1-i+j]  (what the HIP softmax kernel implements)
        i = torch.arange(T)[:, None]
        j = torch.arange(T)[None, :]
        np.testing.assert_array_equal(y.numpy(), x[:, :, i, T - 1 - i + j].numpy())


def test_relative_pe_matches_reference_call_body(golden_dir):
    g = np.load(os.path.join(golden_dir, "relpe_reference.npz"))
    for key in ("3_6_8", "2_9_16"):
        B, T, d = (int(v) for v in key.split("_"))
        pe, table = R.relative_position_encoding(T, d, g[f"len_{key}"].tolist())
        np.testing.assert_allclose(pe.numpy(), g[f"pe_{key}"], atol=2e-6)
        # the gather form used on the GPU: pe_b[r] = table[(r + T - len) mod R] for r < 2len-1 else 0
        Rr = 2 * T - 1
        for b, ln in enumerate(g[f"len_{key}"].tolist()):
            for r in range(Rr):
                want = table[(r + T - ln) % Rr] if r < 2 * ln - 1 else torch.zeros(d)
                np.testing.assert_allclose(pe[b, r].numpy(), want.numpy(), atol=1e-7)


def test_logmel_against_direct_dft_and_pure_tone():
    cfg = R.conformer_config("S")
    sr = 16000
    t = np.arange(4000) / sr
    sig = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)[None]
    feat = R.log_mel(sig, cfg)
    assert feat.shape == (1, 25, 80)
    melw = R.mel_weight_matrix()
    assert melw.shape == (257, 80) and (melw[0] == 0).all() and (melw >= 0).all()
    # energy peaks in the mel bin whose triangle covers 1 kHz
    k1k = int(round(1000.0 / (sr / 512)))
    assert feat[0, 5].argmax() == melw[k1k].argmax()
    # O(N^2) DFT of one frame
    x = R.preemphasis(sig, 0.97)[0]
    fr = np.zeros(512)
    fr[:400] = x[5 * 160:5 * 160 + 400] * R.hann_periodic(400)
    n = np.arange(512)
    dft = np.array([np.sum(fr * np.exp(-2j * np.pi * k * n / 512)) for k in range(257)])
    want = np.log((np.abs(dft) ** 2) @ melw.astype(np.float64) + 1e-6)
    np.testing.assert_allclose(feat[0, 5], want, atol=1e-4)
    assert R.get_nframes([4000, 1, 160, 161]).tolist() == [25, 1, 1, 2]


def test_regularized_set_and_param_count():
    cfg = R.conformer_config("S")
    sh = R.param_shapes(cfg)
    n = sum(int(np.prod(v)) for k, v in sh.items() if R.is_trainable(k))
    assert abs(n - 10.3e6) < 0.1e6
    assert R.is_regularized("enc/block0/ff1/d1/w") and not R.is_regularized("enc/block0/ff1/d1/b")
    assert R.is_regularized("enc/block0/ff1/ln/b") and R.is_regularized("enc/sub/bn0/b") and R.is_regularized("enc/block3/conv/bn/g")
    assert not R.is_regularized("enc/u") and not R.is_regularized("pred/lstm/rk") and R.is_regularized("pred/lstm/k")
    assert not R.is_regularized("enc/block3/conv/bn/mm") and R.is_regularized("pred/emb")


def test_schedule_and_adam():
    lr = R.transformer_schedule(1, 144, 10000, 2.0, 0.05 / 144 ** 0.5)
    assert abs(lr - 2.0 * 144 ** -0.5 * 1e-6) < 1e-12
    assert R.transformer_schedule(10 ** 7, 144, 10000, 2.0, 0.05 / 144 ** 0.5) < 0.05 / 12
    p, g = torch.tensor([1.0, -2.0]), torch.tensor([0.5, 0.25])
    p2, m, v = R.adam_step(p, g, torch.zeros(2), torch.zeros(2), 1, 1e-3, 0.9, 0.98, 1e-9, 0.0)
    np.testing.assert_allclose(p2.numpy(), [1.0 - 1e-3, -2.0 - 1e-3], rtol=1e-5)  # first Adam step = lr * sign(g)


def test_greedy_batch_quirk_last_frame():
    """recognize_batch stops when ALL frame_idx >= nframes-1: with all-blank logits nothing is emitted and the loop runs
    nframes-1 iterations."""
    cfg = R.conformer_config("tiny")
    W = R.init_weights(cfg, seed=1)
    W["joint/vocab/b"][0] = 50.0  # force blank
    enc = torch.randn(2, 6, cfg["dmodel"])
    tokens, prev, h, c = R.recognize_batch(enc, [6, 4], W)
    assert tokens.shape == (2, 13) and (tokens == 0).all()
    tok1, *_ = R.recognize_single(enc[:1], [6], W)
    assert tok1.shape == (1, 18) and (tok1 == 0).all()
