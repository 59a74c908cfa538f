"""ContextNet encoder: forward and every parameter gradient of the HIP path against the torch-CPU
oracle (oracle/contextnet_ref.py) with identical weights, ragged lengths, strided + residual + squeeze-excite blocks; then the
whole transducer step (shared prediction / joint / loss kernels) trains."""
import numpy as np
import pytest
import torch

from oracle import contextnet_ref as R
from tensorflowasr_amd import configs, params
from tensorflowasr_amd.contextnet import ContextNetTransducer
from tensorflowasr_amd.schemas import TrainData, TrainInput, TrainLabel

pytestmark = pytest.mark.gpu


def _load_encoder_weights(model, W):
    for name, w in W.items():
        model.ps.p(name).copy_(w.to(model.device))
    model.ps.refresh_shadow()


@pytest.mark.parametrize("lens", [[57, 57, 57], [57, 31, 44], [57, 1, 9]])
def test_encoder_forward_backward_matches_oracle(dev, lens):
    cfg = configs.contextnet_tiny()
    model = ContextNetTransducer(cfg, dev, dtype=torch.float32, seed=3)
    blocks = params.contextnet_modules(cfg)
    W = R.init_weights(params.param_specs(cfg), seed=11)
    _load_encoder_weights(model, W)
    g = torch.Generator().manual_seed(5)
    B, T0, F = len(lens), max(lens), cfg.num_feature_bins
    feats = torch.randn(B, T0, F, generator=g)
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    ref, ref_len = R.encoder_forward(feats, lens, Wg, blocks)
    ctx = {}
    out, T, elen, _ = model.encoder_fwd(feats.to(dev), lens, True, ctx)
    assert elen == ref_len and T == ref.shape[1] and cfg.time_reduction_factor == 4
    np.testing.assert_allclose(out.view(B, T, -1).cpu().numpy(), ref.detach().numpy(), rtol=2e-3, atol=2e-3)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    model.zero_grad()
    model.encoder_bwd(dy.reshape(B * T, -1).to(dev), ctx)
    torch.cuda.synchronize()
    assert not ctx or set(ctx) == {"enc"}  # every saved activation was consumed
    gmax = max(float(v.grad.abs().max()) for v in Wg.values())
    worst = []
    for k, v in Wg.items():
        mine = model.ps.g(k).cpu().numpy()
        err = np.abs(mine - v.grad.numpy()).max() / max(float(v.grad.abs().max()), 1e-3 * gmax)
        worst.append((err, k))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-2, worst[:6]


def test_transducer_step_trains_and_bf16_tracks_f32(dev):
    cfg = configs.contextnet_tiny()
    rng = np.random.default_rng(0)
    lens, ulens, N, U = [4000, 2900, 3600], [6, 3, 5], 4000, 6
    sig = np.clip(rng.standard_normal((3, N)) * 0.1, -1, 1).astype(np.float32)
    labels = rng.integers(1, cfg.vocab_size, (3, U)).astype(np.int32)
    for b, u in enumerate(ulens):
        labels[b, u:] = 0
    preds = np.concatenate([np.zeros((3, 1), np.int32), labels], 1)
    data = TrainData(TrainInput(torch.from_numpy(sig), torch.tensor(lens, dtype=torch.int32), torch.from_numpy(preds),
                                torch.tensor([u + 1 for u in ulens], dtype=torch.int32)),
                     TrainLabel(torch.from_numpy(labels), torch.tensor(ulens, dtype=torch.int32)))
    costs = {}
    for dtype in (torch.float32, torch.bfloat16):
        model = ContextNetTransducer(cfg, dev, dtype=dtype, seed=1)
        model.optimizer["schedule"] = 3e-3
        first = model.loss_and_backward(data, True, (None, None), want_backward=False).float().cpu().numpy()
        hist = [float(model.train_step(data)["loss"].float().mean()) for _ in range(16)]
        assert np.isfinite(hist).all() and hist[-1] < 0.85 * hist[0] and hist[8] < hist[0], hist
        costs[dtype] = first
    np.testing.assert_allclose(costs[torch.bfloat16], costs[torch.float32], rtol=5e-2)


def test_contextnet_recognize_runs_without_prediction_layernorm(dev):
    """contextnet/small.yml.j2 sets prediction_layer_norm: False: the greedy search must skip pred/ln."""
    from tensorflowasr_amd.schemas import PredictInput

    cfg = configs.contextnet_tiny()
    assert not cfg.prediction_layer_norm
    model = ContextNetTransducer(cfg, dev, dtype=torch.float32, seed=1)
    rng = np.random.default_rng(0)
    sig = np.clip(rng.standard_normal((2, 4000)) * 0.1, -1, 1).astype(np.float32)
    out = model.recognize(PredictInput(torch.from_numpy(sig), torch.tensor([4000, 3000], dtype=torch.int32)))
    T = -(-(-(-4000 // 160)) // cfg.time_reduction_factor)
    assert out.tokens.shape == (2, 2 * T + 1)
    one = model.recognize(PredictInput(torch.from_numpy(sig[:1]), torch.tensor([4000], dtype=torch.int32)))
    assert one.tokens.shape[0] == 1


def test_batched_depthwise_weight_gradients_match_the_per_layer_launches(dev, monkeypatch):
    """bf16: the depthwise weight gradients of all layers go out batched by shape after the encoder's backward (one launch pair per shape,
    tfasr_dwconv_bwd_weight_many); `model.dw_batch = False` restores one launch pair per layer.  Same gradients (partial sums in another order)."""
    cfg = configs.contextnet_tiny()
    model = ContextNetTransducer(cfg, dev, dtype=torch.bfloat16, seed=3)
    g = torch.Generator().manual_seed(5)
    lens = [57, 31, 44]
    B, T0, F = len(lens), max(lens), cfg.num_feature_bins
    feats = torch.randn(B, T0, F, generator=g).to(dev).to(torch.bfloat16)
    grads = {}
    for mode in ("0", "1"):
        model.dw_batch = mode == "1"
        ctx = {}
        out, T, elen, _ = model.encoder_fwd(feats, lens, True, ctx)
        dy = torch.randn(out.shape, generator=torch.Generator().manual_seed(7)).to(dev).to(torch.bfloat16)
        model.zero_grad()
        model.encoder_bwd(dy, ctx)
        torch.cuda.synchronize()
        grads[mode] = model.ps.grad.clone()
    g0, g1 = grads["0"], grads["1"]
    assert bool(torch.isfinite(g1).all()) and float(g0.abs().max()) > 0
    assert float((g1 - g0).abs().max()) <= 2e-3 * float(g0.abs().max())
    names = [m[0] + "/dw" for blk in params.contextnet_modules(cfg) for m in blk["convs"]]
    assert any(float(model.ps.g(n).abs().max()) > 0 for n in names)
