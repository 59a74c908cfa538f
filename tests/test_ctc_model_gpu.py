"""Conformer-CTC (models/ctc/conformer.py:21-143, base_ctc.py:74-149) on the HIP path vs the oracle: logits, CTC loss (mean over the
batch), every gradient, greedy and beam-search decoding; per-layer attention biases (examples/models/ctc/conformer/small.yml.j2:45)."""
import numpy as np
import pytest
import torch

from oracle import conformer_ref as R
from oracle import ctc_ref
from tensorflowasr_amd import configs
from tensorflowasr_amd.ctc_model import ConformerCTC
from tensorflowasr_amd.schemas import PredictInput, TrainData, TrainInput, TrainLabel

pytestmark = pytest.mark.gpu


def _setup(dev, dtype, lens, ulens, N=4000, U=4, seed=0):
    cfg = configs.conformer_tiny(head="ctc", mhsam_use_attention_bias=True)
    ocfg = R.conformer_config("tiny")
    ocfg.update(head="ctc", mhsam_use_attention_bias=True)
    model = ConformerCTC(cfg, dev, dtype=dtype, seed=seed)
    W = R.init_weights(ocfg, seed=seed + 1, scale_bias=0.1)
    model.ps.import_keras(W)
    rng = np.random.default_rng(seed)
    B = len(lens)
    sig = np.clip(rng.standard_normal((B, N)) * 0.1, -1, 1).astype(np.float32)
    for b, n in enumerate(lens):
        sig[b, n:] = 0.0
    labels = rng.integers(1, cfg.vocab_size, (B, U)).astype(np.int32)
    for b, u in enumerate(ulens):
        labels[b, u:] = 0
    data = TrainData(TrainInput(torch.from_numpy(sig), torch.tensor(lens, dtype=torch.int32), torch.zeros(B, 1, dtype=torch.int32),
                                torch.ones(B, dtype=torch.int32)),
                     TrainLabel(torch.from_numpy(labels), torch.tensor(ulens, dtype=torch.int32)))
    return cfg, ocfg, model, W, data, sig, labels


def _oracle(ocfg, W, sig, lens, labels, ulens):
    Wg = {k: v.clone().requires_grad_(R.is_trainable(k)) for k, v in W.items()}
    feat = R.log_mel(sig, ocfg)
    logits, elen = R.ctc_forward(torch.from_numpy(feat), R.get_nframes(lens), Wg, ocfg, training=True)
    tl = np.maximum(elen.numpy(), np.asarray(ulens))
    loss, g = ctc_ref.ctc_loss_and_grad(logits.detach().numpy(), labels, ulens, tl, blank=0, dtype=torch.float64)
    logits.backward(torch.from_numpy(g / len(lens)).to(logits.dtype))
    return logits.detach(), elen, loss, {k: v.grad for k, v in Wg.items() if v.requires_grad}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ctc_conformer_step_matches_oracle(dev, dtype):
    lens, ulens = [4000, 2700, 3300], [4, 2, 3]
    cfg, ocfg, model, W, data, sig, labels = _setup(dev, dtype, lens, ulens)
    ref_logits, elen, ref_loss, ref_grads = _oracle(ocfg, W, sig, lens, labels, ulens)
    assert "enc/block1/mhsa/u" in ref_grads and "dec/logits/w" in ref_grads
    logits, my_elen, _ = model._forward_ctc(data.inputs, True, None, (None, None))  # training-mode statistics, SpecAugment off
    assert my_elen == elen.tolist()
    tol = 2e-3 if dtype == torch.float32 else 6e-2
    np.testing.assert_allclose(logits.float().cpu().numpy(), ref_logits.numpy(), rtol=tol, atol=tol)
    model.zero_grad()
    costs = model.loss_and_backward(data, True, (None, None)).float().cpu().numpy()
    torch.cuda.synchronize()
    np.testing.assert_allclose(costs, ref_loss, rtol=1e-3 if dtype == torch.float32 else 3e-2)
    mine = model.ps.export_keras(model.ps.grad)
    num = sum(float(((mine[k] - g) ** 2).sum()) for k, g in ref_grads.items())
    den = sum(float((g ** 2).sum()) for g in ref_grads.values())
    assert (num / den) ** 0.5 < (2e-3 if dtype == torch.float32 else 0.15)
    # the per-layer attention biases receive their own gradients
    for k in ("enc/block0/mhsa/u", "enc/block1/mhsa/v"):
        a, b = mine[k].numpy(), ref_grads[k].numpy()
        np.testing.assert_allclose(a, b, rtol=0, atol=(2e-3 if dtype == torch.float32 else 0.1) * float(np.abs(b).max()))


def test_ctc_conformer_trains_and_decodes_like_the_oracle(dev):
    lens, ulens = [4000, 4000, 2500], [3, 2, 2]
    cfg, ocfg, model, W, data, sig, labels = _setup(dev, torch.float32, lens, ulens, seed=2)
    model.optimizer["schedule"] = 3e-3
    hist = [float(model.train_step(data, masks=(None, None))["loss"].mean()) for _ in range(30)]
    assert hist[-1] < 0.6 * hist[0], hist
    # decode with the trained weights: HIP greedy / beam vs the oracle's decoders on the oracle's own logits
    Wt = model.ps.export_keras()
    feat = R.log_mel(sig, ocfg)
    with torch.no_grad():
        logits, elen = R.ctc_forward(torch.from_numpy(feat), R.get_nframes(lens), Wt, ocfg, training=False)
    inp = PredictInput(torch.from_numpy(sig), torch.tensor(lens, dtype=torch.int32))
    got = model.recognize(inp).tokens.cpu().numpy()
    want, wl = ctc_ref.ctc_greedy_decode(logits.numpy(), elen.numpy(), blank=0)
    width = max(int(wl.max()), 1)
    np.testing.assert_array_equal(got, want[:, :width])
    beam = model.recognize_beam(inp, beam_width=4).tokens.cpu().numpy()
    V = logits.shape[-1]
    for b in range(len(lens)):
        seq = ctc_ref.ctc_beam_search(logits[b].numpy(), int(elen[b]), 4, blank=V - 1)
        seq = list(seq[0]) if isinstance(seq, tuple) else list(seq)
        assert list(beam[b][:len(seq)]) == seq and not beam[b][len(seq):].any()
    # bf16 model runs the same API
    m16 = ConformerCTC(cfg, dev, dtype=torch.bfloat16, seed=0)
    m16.ps.import_keras({k: v for k, v in Wt.items()})
    t16 = m16.recognize(inp).tokens.cpu().numpy()
    assert t16.shape[0] == len(lens)
