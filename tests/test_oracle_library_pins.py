"""The oracle's restatements of what the reference delegates to TensorFlow / Keras C++ kernels ([ext] in oracle/conformer_ref.py) checked
against INDEPENDENT library implementations of the same published definitions (torch's STFT, LSTM, batch norm, AdamW, Hann window;
scipy's DFT).  TensorFlow itself is not installable here, so these are not reference pins - DESIGN section 4 lists them as
"library-checked" - but they rule out a restatement error in the pieces the reference-source goldens cannot reach."""
import numpy as np
import scipy.fft
import torch
import torch.nn.functional as F

from oracle import conformer_ref as R


def test_hann_window_is_torchs_periodic_hann():
    np.testing.assert_allclose(R.hann_periodic(400), torch.hann_window(400, periodic=True, dtype=torch.float64).numpy(), atol=6e-8)  # f32 rounding of the f64 value


def test_stft_power_matches_torch_stft_and_scipy():
    """tf.signal.stft(frame_length=400, frame_step=160, fft_length=512, pad_end=True): frames start at multiples of the hop, the window
    covers the first 400 samples of each 512-point transform, the signal is zero-padded at the end to ceil(N / hop) frames."""
    cfg = R.conformer_config("S")
    rng = np.random.default_rng(0)
    n = 4000
    sig = (rng.standard_normal((2, n)) * 0.1).astype(np.float32)
    x = R.preemphasis(sig, cfg["preemphasis"])
    T0 = -(-n // 160)
    xp = np.pad(x, [[0, 0], [0, (T0 - 1) * 160 + 512 - n]])  # enough zeros for the last 512-sample transform
    win512 = torch.cat([torch.hann_window(400, periodic=True, dtype=torch.float64), torch.zeros(112, dtype=torch.float64)])
    spec = torch.stft(torch.from_numpy(xp).double(), n_fft=512, hop_length=160, win_length=512, window=win512, center=False,
                      return_complex=True)[:, :, :T0]  # [B, 257, T0]
    power = (spec.abs() ** 2).transpose(1, 2).numpy().astype(np.float32)
    melw = R.mel_weight_matrix()
    ref = np.log(power @ melw + np.float32(1e-6))
    np.testing.assert_allclose(R.log_mel(sig, cfg), ref, atol=5e-5, rtol=0)  # f32 window product / mel sum order
    # the first frame once more through scipy's FFT
    f0 = np.zeros(512)
    f0[:400] = x[0, :400].astype(np.float64) * R.hann_periodic(400)
    p0 = np.abs(scipy.fft.rfft(f0)) ** 2
    np.testing.assert_allclose(np.log(p0.astype(np.float32) @ melw + np.float32(1e-6)), R.log_mel(sig, cfg)[0, 0], atol=5e-5, rtol=0)


def test_lstm_matches_torch_lstm():
    """Keras LSTM gate order i, f, c, o with sigmoid recurrent activation = torch.nn.LSTM's i, f, g, o; one bias vector (torch's second is zero)."""
    torch.manual_seed(0)
    B, U, E, P = 3, 7, 10, 12
    lstm = torch.nn.LSTM(E, P, batch_first=True)
    with torch.no_grad():
        lstm.bias_hh_l0.zero_()
    W = {"l/k": lstm.weight_ih_l0.t().detach(), "l/rk": lstm.weight_hh_l0.t().detach(), "l/b": lstm.bias_ih_l0.detach()}
    x = torch.randn(B, U, E)
    y, h, c = R.lstm(x, None, W, "l/")
    yt, (ht, ct) = lstm(x)
    torch.testing.assert_close(y, yt, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(h, ht[0], atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(c, ct[0], atol=1e-6, rtol=1e-5)
    # masked steps carry the state and emit zeros: equal to running each sequence to its own length
    lens = [7, 4, 1]
    ym, hm, cm = R.lstm(x, lens, W, "l/")
    for b, n in enumerate(lens):
        yb, (hb, cb) = lstm(x[b:b + 1, :n])
        torch.testing.assert_close(ym[b, :n], yb[0], atol=1e-6, rtol=1e-5)
        assert float(ym[b, n:].abs().sum()) == 0.0
        torch.testing.assert_close(hm[b], hb[0, 0], atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(cm[b], cb[0, 0], atol=1e-6, rtol=1e-5)


def test_batch_norm_train_matches_torch():
    torch.manual_seed(1)
    x = torch.randn(4, 9, 6) * 2 + 0.5
    g, b = torch.randn(6), torch.randn(6)
    y, mean, var = R.batch_norm_train(x, g, b)
    rm, rv = torch.zeros(6), torch.ones(6)
    yt = F.batch_norm(x.reshape(-1, 6), rm, rv, g, b, training=True, momentum=1.0, eps=1e-3)
    torch.testing.assert_close(y.reshape(-1, 6), yt, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(mean, rm, atol=1e-6, rtol=1e-5)  # momentum 1: the running mean IS the batch mean
    # (torch keeps the UNBIASED variance in its running estimate, keras the biased one the oracle returns)
    n = x.numel() // 6
    torch.testing.assert_close(var * n / (n - 1), rv, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(R.batch_norm_infer(x, g, b, mean, var), y, atol=1e-5, rtol=1e-5)


def test_layer_norm_and_depthwise_conv_definitions():
    torch.manual_seed(2)
    x = torch.randn(2, 5, 8)
    g, b = torch.randn(8), torch.randn(8)
    mu, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    torch.testing.assert_close(R.layer_norm(x, g, b), (x - mu) / torch.sqrt(var + 1e-3) * g + b, atol=1e-5, rtol=1e-5)
    # causal depthwise convolution written out: y[t, c] = sum_k w[k, c] x[t - (K - 1) + k, c]
    w, bias = torch.randn(3, 8), torch.randn(8)
    y = R.depthwise_conv1d_causal(x, w, bias)
    ref = torch.zeros_like(x)
    for t in range(5):
        for k in range(3):
            s = t - 2 + k
            if s >= 0:
                ref[:, t] += w[k] * x[:, s]
    torch.testing.assert_close(y, ref + bias, atol=1e-5, rtol=1e-5)


def test_adam_step_matches_torch_adamw():
    """keras Adam(weight_decay) = decoupled decay, then the bias-corrected step; torch.optim.AdamW is the same update up to where epsilon enters
    (keras: sqrt(v) + eps under a folded step size; torch: sqrt(v / bc2) + eps) - invisible at eps = 1e-9 with gradients of order 1."""
    torch.manual_seed(3)
    p0 = torch.randn(50)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    p, m, v = p0.clone(), torch.zeros(50), torch.zeros(50)
    for step in range(1, 6):
        gr = torch.randn(50)
        pt.grad = gr.clone()
        opt.step()
        p, m, v = R.adam_step(p, gr, m, v, step, 1e-3)
        torch.testing.assert_close(p, pt.detach(), atol=1e-7, rtol=1e-6)


def test_conv2d_causal_is_left_padded_valid_convolution():
    torch.manual_seed(4)
    x = torch.randn(1, 7, 6, 2)
    w, b = torch.randn(3, 3, 2, 4), torch.randn(4)
    y = R.conv2d_causal_s2(x, w, b)
    assert y.shape == (1, 4, 3, 4)
    ref = torch.zeros(1, 4, 3, 4)
    for to in range(4):
        for fo in range(3):
            for kh in range(3):
                for kw in range(3):
                    ti, fi = 2 * to + kh - 2, 2 * fo + kw - 2
                    if ti >= 0 and fi >= 0:
                        ref[0, to, fo] += x[0, ti, fi] @ w[kh, kw]
    torch.testing.assert_close(y, ref + b, atol=1e-5, rtol=1e-5)
