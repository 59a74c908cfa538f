"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement (NumPy + torch-CPU, fp32 with optional fp64) of the reference Conformer-Transducer
forward pass, greedy decoding and optimizer arithmetic.  Each function cites the reference lines it
follows; `[ext]` marks Keras/TF behaviour restated from the documented defaults because
TensorFlow/Keras are not installable in this container.

Pinning status: the WIRING of every function below - layer order, residual factors, dropout sites, norm positions, the attention
mask's construction, BatchNorm statistics over every frame - is pinned to the reference's own classes: tests/golden/wiring_conformer*.npz
hold per-stage outputs of `tensorflow_asr.models.transducer.conformer.Conformer` constructed and run from /root/reference over
oracle/tf_shim + oracle/keras_shim (oracle/gen_wiring_from_reference.py; checked by tests/test_reference_wiring.py).
`compute_streaming_mask` is also pinned by the reference's own truth tables (tests/test_mask.py:6-55); `rel_left_shift`,
`compute_sinusoid_position_encoding`, the relative-PE roll/mask, `_compute_attention`, the greedy loops, the joint / call_next bodies, the
schedule, the accumulator and the RNN-T / CTC losses by the function-level goldens of oracle/gen_golden_from_reference.py.  What the
Keras / TF LIBRARY kernels compute (Dense/LN/BN/LSTM/conv, tf.signal.stft, the mel matrix) is [ext]: restated from the documented
defaults, twice and independently (here with torch, in oracle/keras_shim.py with NumPy - the wiring goldens make the two agree to
2e-5) and library-checked (tests/test_oracle_library_pins.py); no TensorFlow exists here to execute them.

Weights live in a flat dict name -> torch tensor with the Keras layouts of SURVEY.md A.2.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# config
# ----------------------------------------------------------------------------------------------
def conformer_config(size="S", vocab_size=1000):
    """Conformer-S = examples/models/transducer/conformer/small.yml.j2:1-69; M derived from arXiv:2005.08100 Table 1
    through the same kwargs surface (models/transducer/conformer.py:23-79); `tiny` is a test-only shrink."""
    base = dict(sample_rate=16000, frame_ms=25, stride_ms=10, nfft=512, num_feature_bins=80, preemphasis=0.97,
                epsilon=1e-6, kernel_size=31, num_blocks=16, num_heads=4, ffm_scale=4, ffm_residual=0.5,
                vocab_size=vocab_size, blank=0, l2=1e-6, dropout=0.1)
    if size == "S":
        base.update(dmodel=144, head_size=36, filters=144, embed_dim=320, rnn_units=320, joint_dim=320)
    elif size == "M":
        base.update(dmodel=256, head_size=64, filters=256, embed_dim=640, rnn_units=640, joint_dim=640)
    elif size == "tiny":
        base.update(dmodel=32, head_size=8, filters=32, embed_dim=24, rnn_units=24, joint_dim=40, num_blocks=2,
                    kernel_size=7, vocab_size=min(vocab_size, 29))
    else:
        raise ValueError(size)
    return base


# ----------------------------------------------------------------------------------------------
# frontend  (models/layers/feature_extraction.py)
# ----------------------------------------------------------------------------------------------
def hann_periodic(n):
    """tf.signal.hann_window(n, periodic=True) [ext]."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


def hertz_to_mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_weight_matrix(num_mel_bins=80, num_spectrogram_bins=257, sample_rate=16000, lower=0.0, upper=8000.0):
    """tf.signal.linear_to_mel_weight_matrix [ext]: HTK mel, DC row zero, un-normalised triangles, float32 math."""
    f32 = np.float32
    nyquist = f32(sample_rate) / f32(2.0)
    linear = np.linspace(f32(0.0), nyquist, num_spectrogram_bins, dtype=f32)[1:]
    spec_mel = hertz_to_mel(linear.astype(f32)).astype(f32)[:, None]
    edges = np.linspace(f32(hertz_to_mel(f32(lower))), f32(hertz_to_mel(f32(upper))), num_mel_bins + 2, dtype=f32)
    lo, ce, hi = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (spec_mel - lo) / (ce - lo)
    upper_slopes = (hi - spec_mel) / (hi - ce)
    w = np.maximum(f32(0.0), np.minimum(lower_slopes, upper_slopes)).astype(f32)
    return np.pad(w, [[1, 0], [0, 0]])


def mel_bands(melw):
    """first/last non-zero spectrogram row per mel bin (the HIP kernel only visits these rows)."""
    band = np.zeros((melw.shape[1], 2), np.int32)
    for m in range(melw.shape[1]):
        nz = np.nonzero(melw[:, m])[0]
        band[m] = (nz[0], nz[-1]) if len(nz) else (0, -1)
    return band


def preemphasis(signal, coef=0.97):
    """feature_extraction.py:170-175."""
    if not coef or coef <= 0.0:
        return signal
    return np.concatenate([signal[:, :1], signal[:, 1:] - np.float32(coef) * signal[:, :-1]], axis=-1)


def get_nframes(nsamples, frame_step=160):
    """feature_extraction.py:305-313 (pad_end=True): ceil(n / step)."""
    return -(-np.asarray(nsamples) // frame_step)


def log_mel(signal, cfg):
    """FeatureExtraction.call without augmentation (feature_extraction.py:255-298): [B,N] -> [B,T0,F]."""
    sr = cfg["sample_rate"]
    frame_len = int(round(sr * cfg["frame_ms"] / 1000.0))
    step = int(round(sr * cfg["stride_ms"] / 1000.0))
    nfft = cfg["nfft"]
    x = preemphasis(np.asarray(signal, np.float32), cfg["preemphasis"])
    B, N = x.shape
    T0 = -(-N // step)
    need = (T0 - 1) * step + frame_len
    xp = np.pad(x, [[0, 0], [0, max(0, need - N)]])
    idx = np.arange(T0)[:, None] * step + np.arange(frame_len)[None, :]
    frames = xp[:, idx] * hann_periodic(frame_len)[None, None, :]
    spec = np.fft.rfft(frames.astype(np.float64), n=nfft, axis=-1)
    power = np.square(np.abs(spec)).astype(np.float32)
    melw = mel_weight_matrix(cfg["num_feature_bins"], nfft // 2 + 1, sr, 0.0, 8000.0)
    mel = power @ melw
    return np.log(mel + np.float32(cfg["epsilon"])).astype(np.float32)


def specaugment_apply(feat, fmask, tmask, mask_value=0.0):
    """Mask application of FreqMasking/TimeMasking.augment (specaugment.py:78-86,128-136); masks [B,n,2]=(start,width)."""
    out = np.array(feat, copy=True)
    B, T, Fb = out.shape[:3]
    for b in range(B):
        for f0, fw in (fmask[b] if fmask is not None else []):
            out[b, :, f0:f0 + fw] = mask_value
        for t0, tw in (tmask[b] if tmask is not None else []):
            out[b, t0:t0 + tw] = mask_value
    return out


def specaugment_draw(rng, lengths, nfreq_bins=80, num_freq_masks=1, freq_mask_factor=27, num_time_masks=10,
                     p_upperbound=0.05, prob=1.0):
    """Random draws of specaugment.py:72-77,122-131 in the reference's order: per utterance (tf.map_fn, augmentation.py:78-90)
    all frequency masks then all time masks (sorted keys, augmentation.py:95); per mask `prob`, width, start are ALWAYS drawn and
    multiplied by do_apply afterwards; TimeMasking ignores mask_factor (width bound = floor(len*p)).  `rng` plays
    tf.random.uniform: .uniform() for floats, .integers(lo, hi) for ints (pinned by tests/golden/specaugment_reference.npz)."""
    B = len(lengths)
    fmask = np.zeros((B, num_freq_masks, 2), np.int32)
    tmask = np.zeros((B, num_time_masks, 2), np.int32)
    for b in range(B):
        ln = int(lengths[b])
        for k in range(num_freq_masks):
            do = 1 if rng.uniform() <= prob else 0
            f = do * min(int(rng.integers(0, freq_mask_factor)), nfreq_bins)
            f0 = do * int(rng.integers(0, max(1, nfreq_bins - f)))
            fmask[b, k] = (f0, f)
        Tb = int(math.floor(np.float32(ln) * np.float32(p_upperbound)))
        for k in range(num_time_masks):
            do = 1 if rng.uniform() <= prob else 0
            t = do * min(int(rng.integers(0, max(1, Tb))), ln)
            t0 = do * int(rng.integers(0, max(1, ln - t)))
            tmask[b, k] = (t0, t)
    return fmask, tmask


# ----------------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------------
def swish(x):
    return x * torch.sigmoid(x)


def layer_norm(x, g, b, eps=1e-3):
    """keras LayerNormalization(epsilon=1e-3) [ext]."""
    return F.layer_norm(x, (x.shape[-1],), g, b, eps)


def batch_norm_train(x, g, b, eps=1e-3):
    """keras BatchNormalization training mode (biased batch variance over all but the channel axis) [ext]."""
    dims = tuple(range(x.dim() - 1))
    mean = x.mean(dims)
    var = x.var(dims, unbiased=False)
    return (x - mean) * torch.rsqrt(var + eps) * g + b, mean, var


def batch_norm_infer(x, g, b, mm, mv, eps=1e-3):
    return (x - mm) * torch.rsqrt(mv + eps) * g + b


def conv2d_causal_s2(x, w, b):
    """Conv2D(padding='causal', strides 2): left-pad (k-1) in time AND frequency, then VALID (convolution.py:25-37,132-144).
    x [B,T,F,Cin], w [kh,kw,Cin,Cout]."""
    xp = F.pad(x.permute(0, 3, 1, 2), (w.shape[1] - 1, 0, w.shape[0] - 1, 0))
    y = F.conv2d(xp, w.permute(3, 2, 0, 1), b, stride=2)
    return y.permute(0, 2, 3, 1)


def conv_len(length, stride=2):
    """math_util.conv_output_length(padding='causal'): ceil(L / stride) (math_util.py:282-305)."""
    return (length + stride - 1) // stride


def depthwise_conv1d_causal(x, w, b):
    """DepthwiseConv1D(padding='causal') (convolution.py:159-228): x [B,T,C], w [K,C]."""
    K, C = w.shape
    xp = F.pad(x.transpose(1, 2), (K - 1, 0))
    y = F.conv1d(xp, w.t().unsqueeze(1), b, groups=C)
    return y.transpose(1, 2)


def compute_sinusoid_position_encoding(position, dmodel, interleave=True):
    """positional_encoding.py:31-52 (without the batch repeat)."""
    position = torch.as_tensor(position, dtype=torch.float32)
    min_freq = torch.tensor(1.0 / 10000.0, dtype=torch.float32)
    if interleave:
        ts = torch.pow(min_freq, (2 * (torch.arange(0, dmodel, dtype=torch.float32) // 2)) / float(dmodel))
        ang = position[:, None] * ts[None, :]
        cos_mask = (torch.arange(dmodel) % 2).to(torch.float32)
        return torch.sin(ang) * (1 - cos_mask) + torch.cos(ang) * cos_mask
    ts = torch.pow(min_freq, torch.arange(0, dmodel, 2, dtype=torch.float32) / float(dmodel))
    ang = position[:, None] * ts[None, :]
    return torch.cat([torch.sin(ang), torch.cos(ang)], -1)


def relative_position_encoding(T, dmodel, lengths, interleave=True):
    """RelativeSinusoidalPositionalEncoding.call, causal=False, memory 0 (positional_encoding.py:114-174):
    positions [T-1..1] ++ [0..-(T-1)], per-sample roll(-(T-len)) and zero beyond 2*len-1.  Returns [B, 2T-1, d]."""
    pos = torch.cat([torch.arange(T - 1, 0, -1, dtype=torch.float32), torch.arange(0, -T, -1, dtype=torch.float32)])
    pe = compute_sinusoid_position_encoding(pos, dmodel, interleave)  # [2T-1, d]
    R = 2 * T - 1
    out = []
    for ln in lengths:
        ln = int(ln)
        rolled = torch.roll(pe, shifts=-(T - ln), dims=0)
        mask = (torch.arange(R) < (2 * ln - 1)).to(pe.dtype)[:, None]
        out.append(rolled * mask)
    return torch.stack(out, 0), pe


def rel_left_shift(x):
    """multihead_attention.py:27-77 (causal=False): [B,N,T,R] -> [B,N,T,R-T+1]... returns the padded/reshaped/sliced tensor."""
    b, n, t, r = x.shape
    x = F.pad(x, (0, 1))
    x = x.reshape(b, n, -1)
    x = F.pad(x, (0, r - t))
    x = x.reshape(b, n, 1 + t, r)
    return x[:, :, :t, (t - 1):]


def compute_streaming_mask(chunk_size, history_size, q_len, v_len=None):
    """multihead_attention.py:104-143; returns bool [1, T, S]."""
    v_len = q_len if v_len is None else v_len
    hist = v_len if history_size < 0 else history_size
    rows = []
    for q in range(q_len):
        index = (q // chunk_size) * chunk_size
        start = max(0, index - hist)
        end = min(v_len, index + chunk_size)
        rows.append([(start <= j < end) for j in range(v_len)])
    return np.asarray(rows, bool)[None]


def rel_mhsa(x, pe, W, pfx, H, dh, lengths, u, v, use_mask=True, chunk_size=None, history_size=None):
    """MultiHeadRelativeAttention.call/_compute_attention (multihead_attention.py:543-667); x is the LN output [B,T,d],
    pe [B,2T-1,d]; kernels q/k/v/encoding [d,H,dh] (+bias [H,dh]), out [H,dh,d] (+bias [d]).
    Auto mask = padded QUERY rows only."""
    B, T, d = x.shape
    q = torch.einsum("btd,dhe->bthe", x, W[pfx + "q/w"]) + W[pfx + "q/b"]
    k = torch.einsum("btd,dhe->bthe", x, W[pfx + "k/w"]) + W[pfx + "k/b"]
    vv = torch.einsum("btd,dhe->bthe", x, W[pfx + "v/w"]) + W[pfx + "v/b"]
    p = torch.einsum("brd,dhe->brhe", pe, W[pfx + "pos/w"]) + W[pfx + "pos/b"]
    ctx, _ = rel_attention_core(q, k, vv, p, u, v, dh, lengths, use_mask, chunk_size, history_size)
    return torch.einsum("bthe,hed->btd", ctx, W[pfx + "o/w"]) + W[pfx + "o/b"]


def rel_attention_core(q, k, vv, p, u, v, dh, lengths, use_mask=True, chunk_size=None, history_size=None):
    """MultiHeadRelativeAttention._compute_attention (multihead_attention.py:543-582) on projected tensors q/k/vv [B,T,H,dh],
    p [B,2T-1,H,dh]: content + shifted positional scores, keras auto mask (padded QUERY rows, -1e9 fill: general.py:30-41),
    softmax, weighted values.  Pinned by tests/golden/attention_core_reference.npz (the reference's own function body).
    Returns (context [B,T,H,dh], probabilities [B,H,T,T])."""
    T = q.shape[1]
    scale = 1.0 / math.sqrt(dh)
    cq = (q + u) * scale
    pq = (q + v) * scale
    content = torch.einsum("bshe,bthe->bhts", k, cq)
    positional = torch.einsum("brhe,bthe->bhtr", p, pq)
    positional = rel_left_shift(positional)
    positional = positional[..., positional.shape[-1] - content.shape[-1]:]
    scores = content + positional
    mask = None
    if use_mask and lengths is not None:
        mask = (torch.arange(T)[None, :] < torch.as_tensor(lengths)[:, None])[:, None, :, None].expand(-1, 1, T, T)  # [B,1,T,S]
    if chunk_size is not None and history_size is not None:  # _compute_attention_mask (multihead_attention.py:331-345)
        sm = torch.from_numpy(compute_streaming_mask(int(chunk_size), int(history_size), T))[:, None]  # [1,1,T,S]
        mask = sm if mask is None else (mask & sm)
    if mask is not None:
        scores = torch.where(mask, scores, torch.full_like(scores, -1e9))  # math_util.masked_fill / general.py:30-41
    probs = torch.softmax(scores, dim=-1)
    return torch.einsum("bhts,bshe->bthe", probs, vv), probs


def _no_drop(site, y):
    return y


def ff_module(x, W, pfx, factor=0.5, drop=_no_drop, site=None):
    """FFModule.call (conformer.py:101-109): x + factor * do2(Dense(do1(swish(Dense(LN(x)))))).  `drop(site, y)` injects the
    keras Dropout masks of a training step (default: dropout off); sites `site`, `site + 1` (conformer.py:80,88)."""
    y = layer_norm(x, W[pfx + "ln/g"], W[pfx + "ln/b"])
    y = drop(site, swish(y @ W[pfx + "d1/w"] + W[pfx + "d1/b"]))
    y = drop(None if site is None else site + 1, y @ W[pfx + "d2/w"] + W[pfx + "d2/b"])
    return x + factor * y


def mhsa_module(x, pe, W, pfx, H, dh, lengths, u, v, use_mask=True, chunk_size=None, history_size=None, drop=_no_drop, site=None):
    """MHSAModule.call (conformer.py:209-239); Dropout after the attention (conformer.py:193,234)."""
    y = layer_norm(x, W[pfx + "ln/g"], W[pfx + "ln/b"])
    y = drop(site, rel_mhsa(y, pe, W, pfx, H, dh, lengths, u, v, use_mask, chunk_size, history_size))
    return x + y


def conv_module(x, W, pfx, training=True, stats=None, dw_norm="batch", drop=_no_drop, site=None):
    """ConvModule.call (conformer.py:366-377): LN -> pw(2d) -> GLU -> causal depthwise K -> BN -> swish -> pw(d) -> +res."""
    y = layer_norm(x, W[pfx + "ln/g"], W[pfx + "ln/b"])
    y = y @ W[pfx + "pw1/w"] + W[pfx + "pw1/b"]
    a, b = y.chunk(2, dim=-1)
    y = a * torch.sigmoid(b)  # activations/glu.py:25-28
    y = depthwise_conv1d_causal(y, W[pfx + "dw/w"], W[pfx + "dw/b"])
    if dw_norm == "layer":  # encoder_convm_dw_norm_type: layer -> keras LayerNormalization (conformer.py:334-340), stored in bn/g, bn/b
        y = layer_norm(y, W[pfx + "bn/g"], W[pfx + "bn/b"])
    elif training:
        y, mean, var = batch_norm_train(y, W[pfx + "bn/g"], W[pfx + "bn/b"])
        if stats is not None:
            stats[pfx + "bn"] = (mean.detach(), var.detach())
    else:
        y = batch_norm_infer(y, W[pfx + "bn/g"], W[pfx + "bn/b"], W[pfx + "bn/mm"], W[pfx + "bn/mv"])
    y = swish(y)
    y = drop(site, y @ W[pfx + "pw2/w"] + W[pfx + "pw2/b"])  # Dropout after pw_conv_2 (conformer.py:353,374)
    return x + y


def conformer_block(x, pe, W, pfx, cfg, lengths, u, v, training=True, use_mask=True, stats=None, drop=_no_drop, site=None):
    """ConformerBlock.call (conformer.py:504-535).  Dropout sites of the block (product numbering): site .. site + 5."""
    H, dh = cfg["num_heads"], cfg["head_size"]
    st = (lambda k: None) if site is None else (lambda k: site + k)
    x = ff_module(x, W, pfx + "ff1/", cfg["ffm_residual"], drop, st(0))
    x = mhsa_module(x, pe, W, pfx + "mhsa/", H, dh, lengths, u, v, use_mask, cfg.get("chunk_size"), cfg.get("history_size"), drop, st(2))
    x = conv_module(x, W, pfx + "conv/", training, stats, cfg.get("convm_dw_norm", "batch"), drop, st(3))
    x = ff_module(x, W, pfx + "ff2/", cfg["ffm_residual"], drop, st(4))
    return layer_norm(x, W[pfx + "ln/g"], W[pfx + "ln/b"])


def subsampling(feat, lengths, W, training=True, stats=None, norm="batch"):
    """Conv2dSubsampling.call (subsampling.py:218-230): feat [B,T0,F,1] -> [B,T',F'*C].  `norm`: the yml's `norms` entry -
    "batch" (small.yml.j2:31) or "layer" (small-streaming.yml.j2: a keras LayerNormalization over the channel axis of the
    [B,T,F,C] conv output, subsampling.py:198-207; its gamma / beta live in the bn{i}/g, bn{i}/b slots)."""
    x = feat
    ln = torch.as_tensor(lengths)
    for i in range(2):
        x = conv2d_causal_s2(x, W[f"enc/sub/conv{i}/w"], W[f"enc/sub/conv{i}/b"])
        if norm == "layer":
            x = layer_norm(x, W[f"enc/sub/bn{i}/g"], W[f"enc/sub/bn{i}/b"])
        elif training:
            x, mean, var = batch_norm_train(x, W[f"enc/sub/bn{i}/g"], W[f"enc/sub/bn{i}/b"])
            if stats is not None:
                stats[f"enc/sub/bn{i}"] = (mean.detach(), var.detach())
        else:
            x = batch_norm_infer(x, W[f"enc/sub/bn{i}/g"], W[f"enc/sub/bn{i}/b"], W[f"enc/sub/bn{i}/mm"], W[f"enc/sub/bn{i}/mv"])
        x = swish(x)
        ln = conv_len(ln)
    B, T, Fq, C = x.shape
    return x.reshape(B, T, Fq * C), ln  # math_util.merge_two_last_dims (math_util.py:130-132)


def encoder(feat, lengths, W, cfg, training=True, use_mask=True, stats=None, drop=_no_drop):
    """ConformerEncoder.call (conformer.py:672-701).  `drop(site, y)`: the step's Dropout masks (site 0 = after the linear layer,
    conformer.py:594,682; block i: 16 + 8 i + {0..5}); default = dropout off.  (The relative encoding's own Dropout has rate 0:
    conformer.py:604-610.)"""
    x, ln = subsampling(feat, lengths, W, training, stats, cfg.get("sub_norm", "batch"))
    x = drop(0, x @ W["enc/linear/w"] + W["enc/linear/b"])
    B, T, d = x.shape
    pe, _ = relative_position_encoding(T, d, ln.tolist(), interleave=True)
    pe = pe.to(x.dtype)
    for i in range(cfg["num_blocks"]):
        # shared encoder-level biases (encoders/conformer.py:647-663) or the layer's own pair (multihead_attention.py:522-538)
        u, v = (W[f"enc/block{i}/mhsa/u"], W[f"enc/block{i}/mhsa/v"]) if cfg.get("mhsam_use_attention_bias") else (W["enc/u"], W["enc/v"])
        x = conformer_block(x, pe, W, f"enc/block{i}/", cfg, ln, u, v, training, use_mask, stats, drop, 16 + 8 * i)
    return x, ln


def lstm(x, lengths, W, pfx, h0=None, c0=None):
    """keras LSTM(return_sequences, zero_output_for_mask=True) [ext]: gates i,f,c,o; masked steps carry the state and
    emit zeros (base_transducer.py:71-85).  x [B,U,E] -> (y [B,U,P], h, c)."""
    B, U, _ = x.shape
    P = W[pfx + "rk"].shape[0]
    h = torch.zeros(B, P, dtype=x.dtype) if h0 is None else h0
    c = torch.zeros(B, P, dtype=x.dtype) if c0 is None else c0
    xg = x @ W[pfx + "k"] + W[pfx + "b"]
    ys = []
    for t in range(U):
        z = xg[:, t] + h @ W[pfx + "rk"]
        i, f, g, o = z.chunk(4, dim=-1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        cn = f * c + i * g
        hn = o * torch.tanh(cn)
        if lengths is not None:
            m = (t < torch.as_tensor(lengths)).to(x.dtype)[:, None]
            ys.append(hn * m)
            h = hn * m + h * (1 - m)
            c = cn * m + c * (1 - m)
        else:
            ys.append(hn)
            h, c = hn, cn
    return torch.stack(ys, 1), h, c


def prediction_net(tokens, lengths, W):
    """TransducerPrediction.call (base_transducer.py:123-132): Embedding -> LSTM -> LN."""
    e = W["pred/emb"][tokens.long()]
    y, _, _ = lstm(e, lengths, W, "pred/lstm/")
    return layer_norm(y, W["pred/ln/g"], W["pred/ln/b"])


def joint_net(enc, pred, W):
    """TransducerJoint.call (base_transducer.py:280-293)."""
    e = enc @ W["joint/enc/w"] + W["joint/enc/b"]
    p = pred @ W["joint/pred/w"] + W["joint/pred/b"]
    h = torch.tanh(e[:, :, None, :] + p[:, None, :, :])
    return h @ W["joint/vocab/w"] + W["joint/vocab/b"]


def transducer_forward(features, feat_len, predictions, pred_len, W, cfg, training=True, use_mask=True, stats=None, drop=_no_drop):
    """Transducer.call after the frontend (base_transducer.py:427-435): features [B,T0,F] -> logits [B,T',U1,V]."""
    enc, ln = encoder(features[..., None], feat_len, W, cfg, training, use_mask, stats, drop)
    pred = prediction_net(predictions, pred_len, W)
    return joint_net(enc, pred, W), ln


def ctc_forward(features, feat_len, W, cfg, training=True, use_mask=True, stats=None):
    """CtcModel.call after the frontend (models/ctc/base_ctc.py:74-81): encoder -> ConformerDecoder Dense -> logits [B,T',V]."""
    enc, ln = encoder(features[..., None], feat_len, W, cfg, training, use_mask, stats)
    return enc @ W["dec/logits/w"] + W["dec/logits/b"], ln


def l2_regularization(W, l2=1e-6):
    """Sum of keras l2 regularizers: kernels, LN/BN gamma+beta, embedding (conformer.py:63-68 etc.; biases and the
    shared u/v attention biases use bias_regularizer=None)."""
    tot = 0.0
    for k, w in W.items():
        if is_regularized(k):
            tot = tot + l2 * (w.double() ** 2).sum()
    return tot


def is_regularized(name):
    """Which variables carry the keras l2(1e-6) regulariser: every kernel (Dense/EinsumDense/Conv/depthwise/LSTM input
    kernel/Embedding) and the gamma AND beta of every LayerNorm/BatchNorm (gamma_regularizer=beta_regularizer=
    kernel_regularizer, e.g. conformer.py:59-68,327-333); biases, the LSTM recurrent kernel (recurrent_regularizer=None)
    and the shared u/v biases (bias_regularizer=None) do not."""
    parts = name.split("/")
    leaf, parent = parts[-1], parts[-2] if len(parts) > 1 else ""
    if leaf in ("mm", "mv") or name in ("enc/u", "enc/v") or (parent == "mhsa" and leaf in ("u", "v")):
        return False
    if leaf == "w" or name in ("pred/emb", "pred/lstm/k"):
        return True
    if parent == "ln" or parent.startswith("bn"):
        return leaf in ("g", "b")
    return False


def is_trainable(name):
    return not (name.endswith("/mm") or name.endswith("/mv"))


# ----------------------------------------------------------------------------------------------
# parameters (Keras default initialisers) in a deterministic order
# ----------------------------------------------------------------------------------------------
def param_shapes(cfg):
    d, H, dh, C = cfg["dmodel"], cfg["num_heads"], cfg["head_size"], cfg["filters"]
    K, V, E, P, J = cfg["kernel_size"], cfg["vocab_size"], cfg["embed_dim"], cfg["rnn_units"], cfg["joint_dim"]
    Fb = cfg["num_feature_bins"]
    F2 = conv_len(conv_len(Fb))
    s = {}
    s["enc/sub/conv0/w"] = (3, 3, 1, C); s["enc/sub/conv0/b"] = (C,)
    s["enc/sub/bn0/g"] = (C,); s["enc/sub/bn0/b"] = (C,); s["enc/sub/bn0/mm"] = (C,); s["enc/sub/bn0/mv"] = (C,)
    s["enc/sub/conv1/w"] = (3, 3, C, C); s["enc/sub/conv1/b"] = (C,)
    s["enc/sub/bn1/g"] = (C,); s["enc/sub/bn1/b"] = (C,); s["enc/sub/bn1/mm"] = (C,); s["enc/sub/bn1/mv"] = (C,)
    s["enc/linear/w"] = (F2 * C, d); s["enc/linear/b"] = (d,)
    if not cfg.get("mhsam_use_attention_bias"):
        s["enc/u"] = (H, dh); s["enc/v"] = (H, dh)
    for i in range(cfg["num_blocks"]):
        p = f"enc/block{i}/"
        for ff in ("ff1/", "ff2/"):
            s[p + ff + "ln/g"] = (d,); s[p + ff + "ln/b"] = (d,)
            s[p + ff + "d1/w"] = (d, cfg["ffm_scale"] * d); s[p + ff + "d1/b"] = (cfg["ffm_scale"] * d,)
            s[p + ff + "d2/w"] = (cfg["ffm_scale"] * d, d); s[p + ff + "d2/b"] = (d,)
        m = p + "mhsa/"
        s[m + "ln/g"] = (d,); s[m + "ln/b"] = (d,)
        for nm in ("q", "k", "v", "pos"):
            s[m + nm + "/w"] = (d, H, dh); s[m + nm + "/b"] = (H, dh)
        s[m + "o/w"] = (H, dh, d); s[m + "o/b"] = (d,)
        if cfg.get("mhsam_use_attention_bias"):
            s[m + "u"] = (H, dh); s[m + "v"] = (H, dh)
        c = p + "conv/"
        s[c + "ln/g"] = (d,); s[c + "ln/b"] = (d,)
        s[c + "pw1/w"] = (d, 2 * d); s[c + "pw1/b"] = (2 * d,)
        s[c + "dw/w"] = (K, d); s[c + "dw/b"] = (d,)
        s[c + "bn/g"] = (d,); s[c + "bn/b"] = (d,); s[c + "bn/mm"] = (d,); s[c + "bn/mv"] = (d,)
        s[c + "pw2/w"] = (d, d); s[c + "pw2/b"] = (d,)
        s[p + "ln/g"] = (d,); s[p + "ln/b"] = (d,)
    if cfg.get("head") == "ctc":  # ConformerDecoder: Dense(vocab_size) named "logits" (models/ctc/conformer.py:21-47)
        s["dec/logits/w"] = (d, V); s["dec/logits/b"] = (V,)
        return s
    s["pred/emb"] = (V, E)
    s["pred/lstm/k"] = (E, 4 * P); s["pred/lstm/rk"] = (P, 4 * P); s["pred/lstm/b"] = (4 * P,)
    s["pred/ln/g"] = (P,); s["pred/ln/b"] = (P,)
    s["joint/enc/w"] = (d, J); s["joint/enc/b"] = (J,)
    s["joint/pred/w"] = (P, J); s["joint/pred/b"] = (J,)
    s["joint/vocab/w"] = (J, V); s["joint/vocab/b"] = (V,)
    return s


def _fans(name, shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    if name.endswith("conv0/w") or name.endswith("conv1/w"):
        rf = shape[0] * shape[1]
        return shape[2] * rf, shape[3] * rf
    if name.endswith("o/w"):  # [H, dh, d]: keras computes fans on the raw shape: receptive = H, in = dh, out = d
        return shape[0] * shape[1], shape[0] * shape[2]
    if len(shape) == 3:  # [d, H, dh]: receptive = d, fan_in = H*d?  keras: in = shape[-2]*prod(shape[:-2]), out = shape[-1]*prod(shape[:-2])
        rf = shape[0]
        return shape[1] * rf, shape[2] * rf
    return shape[0], shape[-1]


def init_weights(cfg, seed=3, scale_bias=0.0, dtype=torch.float32):
    """Keras default initialisers [ext]: glorot_uniform kernels, zeros biases, ones gamma, uniform(-.05,.05) embedding,
    orthogonal LSTM recurrent kernel, unit forget bias, moving_var ones.  `scale_bias` > 0 draws small random biases /
    u / v / gamma perturbations so parity tests exercise every parameter (default 0 = the Keras values)."""
    rng = np.random.default_rng(seed)
    W = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("/g") or name.endswith("/mv"):
            w = np.ones(shape, np.float32)
            if scale_bias and name.endswith("/g"):
                w = w + rng.uniform(-scale_bias, scale_bias, shape).astype(np.float32)
        elif name == "pred/emb":
            w = rng.uniform(-0.05, 0.05, shape).astype(np.float32)
        elif name == "pred/lstm/rk":
            P = shape[0]
            a = rng.standard_normal((4 * P, P))
            q, r = np.linalg.qr(a)
            q = q * np.sign(np.diag(r))
            w = q.T.astype(np.float32)  # [P, 4P]
        elif name == "pred/lstm/b":
            P = shape[0] // 4
            w = np.zeros(shape, np.float32)
            w[P:2 * P] = 1.0
        elif name.endswith("/w") or name.endswith("/k"):
            if name.endswith("dw/w"):
                fi, fo = shape[0] * shape[1], shape[0]  # keras depthwise kernel [K, C, 1]: rfs=K, in=C*K, out=K
            else:
                fi, fo = _fans(name, shape)
            lim = math.sqrt(6.0 / (fi + fo))
            w = rng.uniform(-lim, lim, shape).astype(np.float32)
        else:  # biases, beta, moving_mean, u, v
            w = np.zeros(shape, np.float32)
            if scale_bias and not (name.endswith("/mm")):
                w = rng.uniform(-scale_bias, scale_bias, shape).astype(np.float32)
        W[name] = torch.from_numpy(w).to(dtype)
    return W


# ----------------------------------------------------------------------------------------------
# optimizer arithmetic
# ----------------------------------------------------------------------------------------------
def transformer_schedule(step, dmodel, warmup_steps=10000, scale=2.0, max_lr=None, min_lr=None):
    """TransformerSchedule.__call__ (optimizers/schedules.py:28-37), float32 arithmetic."""
    f32 = np.float32
    step = f32(step)
    if step <= 0:  # keras' `iterations` is 0 at the first update: tf gives min(inf, 0) = 0 (then max_lr / min_lr as usual)
        lr = f32(0.0)
    else:
        lr = f32(dmodel) ** f32(-0.5) * min(step ** f32(-0.5), step * f32(warmup_steps) ** f32(-1.5))
    lr = f32(scale) * lr
    if max_lr is not None:
        lr = min(f32(max_lr), lr)
    if min_lr is not None:
        lr = max(f32(min_lr), lr)
    return float(lr)


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.98, eps=1e-9, weight_decay=1e-6):
    """keras.optimizers.Adam.update_step [ext] with decoupled weight_decay applied first (small.yml.j2:84-87)."""
    p = p - p * weight_decay * lr
    m = m + (g - m) * (1 - beta1)
    v = v + (g * g - v) * (1 - beta2)
    alpha = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    p = p - alpha * m / (torch.sqrt(v) + eps)
    return p, m, v


def ga_gradients(micro_grads):
    """GradientAccumulator semantics (optimizers/accumulation.py:54-70): apply step uses (g_last + acc) / ga_steps."""
    return sum(micro_grads) / len(micro_grads)


# ----------------------------------------------------------------------------------------------
# greedy decoding (base_transducer.py:437-712)
# ----------------------------------------------------------------------------------------------
def _call_next(enc_frame, prev_tok, h, c, W):
    """Transducer.call_next (base_transducer.py:437-464): one prediction-net step + joint + log_softmax."""
    e = W["pred/emb"][prev_tok.long()]  # [B,1,E]
    y, hn, cn = lstm(e, None, W, "pred/lstm/", h, c)
    y = layer_norm(y, W["pred/ln/g"], W["pred/ln/b"])
    logits = joint_net(enc_frame, y, W)  # [B,1,1,V]
    return torch.log_softmax(logits, -1), hn, cn


def recognize_batch(encoded, encoded_length, W, blank=0, max_iters=None):
    """Transducer.recognize_batch (base_transducer.py:496-575) on already-encoded frames. Returns tokens [B, 2T+1]."""
    B, T, _ = encoded.shape
    nframes = torch.as_tensor(encoded_length).long().view(B, 1)
    P = W["pred/lstm/rk"].shape[0]
    frame_idx = torch.zeros(B, 1, dtype=torch.long)
    prev_tok = torch.full((B, 1), blank, dtype=torch.long)
    h = torch.zeros(B, P, dtype=encoded.dtype)
    c = torch.zeros(B, P, dtype=encoded.dtype)
    max_tokens = T * 2 + 1
    tokens = torch.full((B, max_tokens), blank, dtype=torch.long)
    tok_idx = torch.ones(B, 1, dtype=torch.long)
    iters = 0
    while True:
        if bool((frame_idx >= nframes - 1).all()) or bool((tok_idx >= max_tokens - 1).all()):
            break
        iters += 1
        # the reference's tf.while_loop has no bound: a sample that keeps emitting non-blank symbols at
        # tok_idx == max_tokens-1 never advances its frame, so the loop can spin forever.  The product caps the trip count
        # at T + max_tokens + 2 (every useful iteration advances a frame or appends a token); mirror that here.
        if iters > (max_iters if max_iters is not None else T + max_tokens + 2):
            break
        fi = torch.minimum(frame_idx, nframes - 1)
        cur = encoded[torch.arange(B), fi[:, 0]][:, None, :]
        lsm, hn, cn = _call_next(cur, prev_tok, h, c, W)
        cur_tok = lsm.argmax(-1).view(B, 1)
        eq_blank = cur_tok == blank
        eq_blank = eq_blank | (tok_idx >= max_tokens)
        eq_blank = eq_blank | (frame_idx > nframes)
        upd_tok = torch.where(eq_blank, torch.full_like(cur_tok, blank), cur_tok).view(B)
        upd_idx = torch.where(eq_blank, torch.zeros_like(tok_idx), torch.clamp(tok_idx + 1, max=max_tokens - 1))
        tokens[torch.arange(B), upd_idx[:, 0]] = upd_tok
        tok_idx = torch.where(eq_blank, tok_idx, torch.clamp(tok_idx + 1, max=max_tokens - 1))
        frame_idx = torch.where(eq_blank, frame_idx + 1, frame_idx)
        prev_tok = torch.where(eq_blank, prev_tok, cur_tok)
        h = torch.where(eq_blank, h, hn)
        c = torch.where(eq_blank, c, cn)
    return tokens, prev_tok, h, c


def recognize_single(encoded, encoded_length, W, blank=0, max_tokens_per_frame=3):
    """Transducer.recognize_single (base_transducer.py:577-712), batch size 1. Returns tokens [1, nframes*3]."""
    nframes = int(encoded_length[0])
    P = W["pred/lstm/rk"].shape[0]
    frame = 0
    prev_tok = torch.full((1, 1), blank, dtype=torch.long)
    h = torch.zeros(1, P, dtype=encoded.dtype)
    c = torch.zeros(1, P, dtype=encoded.dtype)
    token_index = -1
    tokens = [0] * (nframes * max_tokens_per_frame)
    per_frame = [0] * nframes
    while frame < nframes:
        cur = encoded[:, frame:frame + 1]
        lsm, hn, cn = _call_next(cur, prev_tok, h, c, W)
        cur_tok = int(lsm.argmax(-1).view(-1)[0])
        is_blank = cur_tok == blank
        if not is_blank:
            per_frame[frame] += 1
        nf = per_frame[frame]
        if is_blank or nf >= max_tokens_per_frame:
            frame_next = frame + 1
        else:
            frame_next = frame
        if not is_blank:
            token_index += 1
            prev_tok = torch.full((1, 1), cur_tok, dtype=torch.long)
            h, c = hn, cn
        if token_index >= 0:
            tokens[token_index] = int(prev_tok.view(-1)[0])
        frame = frame_next
    return torch.tensor(tokens, dtype=torch.long)[None, :], prev_tok, h, c
