"""GPU parity of the whole Conformer-Transducer step (forward, RNN-T loss, every gradient, Adam) vs the torch-CPU oracle
with identical weights and inputs (dropout 0, SpecAugment masks injected: SURVEY.md §7 'hard parts')."""
import numpy as np
import pytest
import torch

from oracle import conformer_ref as R
from oracle import rnnt_ref
from tensorflowasr_amd import configs
from tensorflowasr_amd.conformer import ConformerTransducer
from tensorflowasr_amd.schemas import TrainData, TrainInput, TrainLabel

pytestmark = pytest.mark.gpu


def _setup(dev, dtype, lens, ulens, seed=0, N=4000, U=6, **over):
    cfg = configs.conformer_tiny(**over)
    ocfg = R.conformer_config("tiny")
    ocfg.update({k: v for k, v in over.items() if k in ("chunk_size", "history_size", "convm_dw_norm")})
    model = ConformerTransducer(cfg, dev, dtype=dtype, seed=seed)
    model.use_pred_stream = True
    W = R.init_weights(ocfg, seed=seed + 1, scale_bias=0.1)
    gen = torch.Generator().manual_seed(seed + 7)
    for k in W:  # make BN / LN affine parameters non-trivial
        if k.endswith("/g"):
            W[k] = W[k] + 0.1 * torch.randn(W[k].shape, generator=gen)
    model.ps.import_keras(W)
    rng = np.random.default_rng(seed)
    B = len(lens)
    sig = np.clip(rng.standard_normal((B, N)) * 0.1, -1, 1).astype(np.float32)
    for b, n in enumerate(lens):
        sig[b, n:] = 0.0
    labels = rng.integers(1, cfg.vocab_size, (B, U)).astype(np.int32)
    for b, u in enumerate(ulens):
        labels[b, u:] = 0
    preds = np.concatenate([np.zeros((B, 1), np.int32), labels], 1)  # blank-prepended (tokenizers.py:165-167)
    data = TrainData(
        TrainInput(torch.from_numpy(sig), torch.tensor(lens, dtype=torch.int32), torch.from_numpy(preds),
                   torch.tensor([u + 1 for u in ulens], dtype=torch.int32)),
        TrainLabel(torch.from_numpy(labels), torch.tensor(ulens, dtype=torch.int32)))
    return cfg, ocfg, model, W, data, sig, labels, preds


def _oracle_step(ocfg, W, sig, lens, preds, ulens, labels, masks=None, use_mask=True, drop=None):
    Wg = {k: v.clone().requires_grad_(R.is_trainable(k)) for k, v in W.items()}
    feat = R.log_mel(sig, ocfg)
    flen = R.get_nframes(lens)
    if masks is not None:
        feat = R.specaugment_apply(feat, masks[0], masks[1])
    stats = {}
    logits, elen = R.transducer_forward(torch.from_numpy(feat), flen, torch.from_numpy(preds), torch.tensor([u + 1 for u in ulens]),
                                        Wg, ocfg, training=True, use_mask=use_mask, stats=stats, **({"drop": drop} if drop else {}))
    tl, ul = rnnt_ref.clamp_lengths(elen.numpy(), np.asarray(ulens))
    loss, g = rnnt_ref.rnnt_loss_and_grad(logits.detach().numpy(), labels, ul, np.minimum(tl, logits.shape[1]), np.float32)
    B = len(lens)
    logits.backward(torch.from_numpy(g / B).to(logits.dtype))
    grads = {k: v.grad for k, v in Wg.items() if v.requires_grad}
    return logits.detach(), elen, loss, grads, stats


@pytest.mark.parametrize("lens,ulens", [([4000, 4000], [6, 6]), ([4000, 2500, 3100], [6, 3, 5]),
                                        # edge cases: an EMPTY transcript, a one-frame utterance (161 samples -> 2 -> 1 -> 1 frames)
                                        # whose label is longer than its encoder output (logit_length is raised to it,
                                        # losses/base_loss.py:36), a single-label utterance
                                        ([4000, 161, 1600, 3000], [6, 2, 0, 1])])
def test_f32_step_matches_oracle(dev, lens, ulens):
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.float32, lens, ulens)
    masks = model.draw_specaugment([int(n) for n in R.get_nframes(lens)])
    ref_logits, elen, ref_loss, ref_grads, stats = _oracle_step(ocfg, W, sig, lens, preds, ulens, labels,
                                                                 (masks[0].numpy(), masks[1].numpy()))
    # forward only (training-mode statistics) -----------------------------------------------------
    logits, my_elen, _ = model._forward(data.inputs, True, None, masks)
    assert my_elen == elen.tolist()
    np.testing.assert_allclose(logits.cpu().numpy(), ref_logits.numpy(), rtol=2e-3, atol=2e-3)
    # full step -------------------------------------------------------------------------------------
    # dense-lattice path (what Transducer.call materialises) and packed-lattice path give the same step
    model.zero_grad()
    costs_dense = model.loss_and_backward(data, True, masks, packed=False)
    g_dense = model.ps.grad.clone()
    model.zero_grad()
    costs = model.loss_and_backward(data, True, masks)
    torch.cuda.synchronize()
    np.testing.assert_allclose(costs.cpu().numpy(), ref_loss, rtol=1e-3)  # BASELINE.json: loss within 1e-3 relative
    np.testing.assert_allclose(costs_dense.cpu().numpy(), costs.cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(g_dense.cpu().numpy(), model.ps.grad.cpu().numpy(), rtol=1e-3, atol=1e-5 * float(g_dense.abs().max()))
    mine = model.ps.export_keras(model.ps.grad)
    worst = []
    gmax = max(float(g.abs().max()) for g in ref_grads.values())
    for k, g in ref_grads.items():
        a, b = mine[k].numpy().reshape(-1), g.numpy().reshape(-1)
        # biases feeding a BatchNorm / the k and pos biases have an analytically ZERO gradient (pure rounding noise in
        # both implementations), so errors are measured against max(|g|, 1e-3 * largest gradient in the model)
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-3 * gmax)
        worst.append((err, k))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-2, worst[:8]
    # moving statistics updated like keras (momentum .99)
    mm = model.ps.state["enc/block0/conv/bn/mm"].cpu()
    # three training-mode forwards ran above: moving = mean * (1 - 0.99^3)
    np.testing.assert_allclose(mm.numpy(), ((1 - 0.99 ** 3) * stats["enc/block0/conv/bn"][0]).numpy(), rtol=2e-2, atol=1e-5)
    # optimizer --------------------------------------------------------------------------------------
    before = model.ps.export_keras()
    assert model.step == 0 and model.learning_rate(0) == 0.0  # keras: schedule(iterations = 0) at the first update
    model.step = 7                                              # a later update: lr = schedule(7), bias correction with 8
    lr = model.apply_gradients()
    assert lr == model.learning_rate(7) > 0
    after = model.ps.export_keras()
    k = "enc/block1/ff2/d1/w"
    g = mine[k] + 2 * cfg.l2 * before[k]
    p_ref, _, _ = R.adam_step(before[k], g, torch.zeros_like(g), torch.zeros_like(g), 8, lr, 0.9, 0.98, 1e-9, 1e-6)
    np.testing.assert_allclose(after[k].numpy(), p_ref.numpy(), rtol=1e-5, atol=1e-7)
    k = "enc/block1/ff2/d1/b"  # not regularised
    p_ref, _, _ = R.adam_step(before[k], mine[k], torch.zeros_like(mine[k]), torch.zeros_like(mine[k]), 8, lr, 0.9, 0.98, 1e-9, 1e-6)
    np.testing.assert_allclose(after[k].numpy(), p_ref.numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("chunk,hist", [(2, 4), (3, -1)])
def test_streaming_conformer_step_matches_oracle(dev, dtype, chunk, hist):
    """small-streaming.yml.j2's encoder: chunked attention mask (multihead_attention.py:104-143,331-345, pinned to the reference's
    truth tables) ANDed with the padded-query auto mask + LayerNormalization after the depthwise conv (conformer.py:334-340):
    logits, loss and every gradient of the train step against the oracle, ragged lengths; native executor and host path."""
    lens, ulens = [4000, 2500, 3100], [6, 3, 5]
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, dtype, lens, ulens, chunk_size=chunk, history_size=hist, convm_dw_norm="layer")
    assert model._fused_attention() == (dtype == torch.bfloat16)  # bf16: streaming mask inside the fused kernels (heads stored padded to 64)
    ref_logits, elen, ref_loss, ref_grads, _ = _oracle_step(ocfg, W, sig, lens, preds, ulens, labels, None)
    # the mask matters: the full-context oracle gives different logits
    ocfg_full = dict(ocfg, chunk_size=None, history_size=None)
    full_logits = _oracle_step(ocfg_full, W, sig, lens, preds, ulens, labels, None)[0]
    assert float((full_logits - ref_logits).abs().max()) > 1e-3
    tol = 2e-3 if dtype == torch.float32 else 6e-2
    out = {}
    for native in (True, False):
        model.native_blocks = native
        logits, _, _ = model._forward(data.inputs, True, None, (None, None))
        np.testing.assert_allclose(logits.float().cpu().numpy(), ref_logits.numpy(), rtol=tol, atol=tol)
        model.zero_grad()
        costs = model.loss_and_backward(data, True, (None, None)).float().cpu().numpy()
        torch.cuda.synchronize()
        np.testing.assert_allclose(costs, ref_loss, rtol=1e-3 if dtype == torch.float32 else 3e-2)
        mine = model.ps.export_keras(model.ps.grad)
        num = sum(float(((mine[k] - g) ** 2).sum()) for k, g in ref_grads.items())
        den = sum(float((g ** 2).sum()) for g in ref_grads.values())
        assert (num / den) ** 0.5 < (2e-3 if dtype == torch.float32 else 0.15), (native, (num / den) ** 0.5)
        out[native] = model.ps.grad.clone()
    g0, g1 = out[False].cpu().numpy(), out[True].cpu().numpy()
    t2 = 1e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(g1, g0, rtol=t2, atol=t2 * float(np.abs(g0).max()))


def test_bf16_step_close_to_oracle(dev):
    lens, ulens = [4000, 3300], [6, 4]
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, lens, ulens)
    ref_logits, elen, ref_loss, ref_grads, _ = _oracle_step(ocfg, W, sig, lens, preds, ulens, labels, None)
    model.zero_grad()
    costs = model.loss_and_backward(data, True, (None, None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(costs.cpu().numpy(), ref_loss, rtol=3e-2)
    mine = model.ps.export_keras(model.ps.grad)
    num = sum(float(((mine[k] - g) ** 2).sum()) for k, g in ref_grads.items())
    den = sum(float((g ** 2).sum()) for g in ref_grads.values())
    assert (num / den) ** 0.5 < 0.15


def test_train_steps_reduce_loss_and_ga(dev):
    lens, ulens = [4000, 4000], [6, 5]
    cfg, ocfg, model, W, data, *_ = _setup(dev, torch.float32, lens, ulens)
    model.optimizer["schedule"] = 2e-3
    losses = []
    for _ in range(8):
        losses.append(float(model.train_step(data, masks=(None, None))["loss"].mean()))
    assert losses[-1] < losses[0]
    # gradient accumulation: 2 identical micro-batches == 1 step on the same batch (accumulation.py:64-70)
    m1 = _setup(dev, torch.float32, lens, ulens)[2]
    m2 = _setup(dev, torch.float32, lens, ulens)[2]
    for m in (m1, m2):
        m.optimizer["schedule"] = 1e-3
        m.optimizer["eps"] = 1e-4  # keeps the update of (analytically) zero-gradient variables out of the rounding noise
    m2.ga_steps = 2
    m1.train_step(data, masks=(None, None))
    m2.train_step(data, masks=(None, None))
    assert m2.step == 0
    m2.train_step(data, masks=(None, None))
    assert m2.step == 1
    a, b = m1.ps.flat.cpu().numpy(), m2.ps.flat.cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=0, atol=5e-5)


@pytest.mark.parametrize("bias", [1.2, 1.6])
@pytest.mark.parametrize("lens", [[4000, 4000, 4000], [4000, 2000, 3000], [4000]])
def test_greedy_decode_tokens_bit_exact(dev, lens, bias):
    """BASELINE north_star: bit-exact token indices for greedy decode vs the reference semantics (f32 path), including the
    batch-loop quirks (last frame of the slowest sample never decoded, tokens from column 2) and the bs=1 variant."""
    ulens = [3] * len(lens)
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.float32, lens, ulens, seed=3)
    W["joint/vocab/b"] = W["joint/vocab/b"].clone()
    W["joint/vocab/b"][0] += bias  # blank/non-blank mix: some samples saturate their token buffer, others emit a few
    model.ps.import_keras(W)
    # oracle: inference-mode encoder (moving BN statistics) + the reference's greedy loops
    feat = R.log_mel(sig, ocfg)
    enc_ref, elen = R.encoder(torch.from_numpy(feat)[..., None], R.get_nframes(lens), W, ocfg, training=False)
    from tensorflowasr_amd.schemas import PredictInput
    out = model.recognize(PredictInput(torch.from_numpy(sig), torch.tensor(lens, dtype=torch.int32)))
    enc_mine, my_elen = model.encode(torch.from_numpy(sig), torch.tensor(lens, dtype=torch.int32))
    np.testing.assert_allclose(enc_mine.cpu().numpy(), enc_ref.numpy(), rtol=2e-3, atol=2e-3)
    if len(lens) == 1:
        tok_ref, prev_ref, h_ref, c_ref = R.recognize_single(enc_ref, elen.tolist(), W)
    else:
        tok_ref, prev_ref, h_ref, c_ref = R.recognize_batch(enc_ref, elen.tolist(), W)
    assert out.tokens.shape == tok_ref.shape
    np.testing.assert_array_equal(out.tokens.cpu().numpy(), tok_ref.numpy())
    np.testing.assert_array_equal(out.next_tokens.cpu().numpy().reshape(-1), prev_ref.numpy().reshape(-1))
    np.testing.assert_allclose(out.next_decoder_states[:, 0, 0].cpu().numpy(), h_ref.numpy(), rtol=1e-3, atol=1e-4)


def test_dropout_kernel_statistics_and_gemm_consistency(dev):
    from tensorflowasr_amd import kernels as K

    x = torch.ones(1000, 256, device=dev)
    y = K.dropout(x, 0.1, 1234)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.005
    np.testing.assert_allclose(y[y != 0].cpu().numpy(), 1.0 / 0.9, rtol=1e-6)
    y2 = K.dropout(x, 0.1, 1235)
    assert (y != y2).any()
    # the GEMM epilogue draws the SAME mask as the standalone kernel for the same (seed, element index)
    for dtype in (torch.float32, torch.bfloat16):
        A = torch.eye(256, device=dev, dtype=dtype)
        out = K.matmul(torch.ones(1000, 256, device=dev, dtype=dtype), A, drop_p=0.1, drop_seed=1234)
        np.testing.assert_array_equal((out.float() != 0).cpu().numpy(), (y != 0).cpu().numpy())


def test_dropout_mask_is_statistically_independent(dev):
    """Round 5 replaced the per-pair murmur3 finaliser by two rounds of 24-bit multiply + xor-shift (csrc/common.h drop_mix, full-rate vector
    instructions).  What a dropout mask has to be: keep rate 1 - p, row / column keep rates with the spread of independent draws (no lattice
    along the row stride), no correlation at small lags or at the row stride, and masks of neighbouring seeds (consecutive dropout sites,
    consecutive steps, the rank bits) agreeing no more often than independent ones (0.9^2 + 0.1^2 = 0.82)."""
    from tensorflowasr_amd import kernels as K

    for rows, C in ((4096, 1024), (19264, 256)):
        ones = torch.ones(rows, C, device=dev)
        keep = (K.dropout(ones, 0.1, 8192 * 5 + 16) != 0).float()
        kr = keep.mean().item()
        assert abs(kr - 0.9) < 4 * (0.09 / (rows * C)) ** 0.5 + 1e-4
        col, row = keep.mean(0).std().item(), keep.mean(1).std().item()
        assert 0.8 < col / (0.09 / rows) ** 0.5 < 1.25, col
        assert 0.8 < row / (0.09 / C) ** 0.5 < 1.25, row
        a = (keep - kr).reshape(-1)
        for lag in (1, 2, 3, 4, C // 2, C, 2 * C):
            ac = float((a[:-lag] * a[lag:]).mean() / 0.09)
            assert abs(ac) < 5.0 / (rows * C) ** 0.5 + 2e-3, (lag, ac)
    base = K.dropout(torch.ones(2048, 1024, device=dev), 0.1, 8192 * 7 + 18) != 0
    for seed in (8192 * 7 + 19, 8192 * 7 + 20, 8192 * 8 + 18, 8192 * 7 + 18 + (1 << 27), 8192 * 7 + 18 + (1 << 40)):
        agree = ((K.dropout(torch.ones(2048, 1024, device=dev), 0.1, seed) != 0) == base).float().mean().item()
        assert abs(agree - 0.82) < 3e-3, (seed, agree)


def test_dropout_backward_consistent_finite_difference(dev):
    """With dropout ON the regenerated backward masks must match the forward ones: directional finite difference of the
    mean loss along a random parameter direction vs <grad, direction> (f32 path, fixed mask epoch)."""
    lens, ulens = [4000, 4000], [6, 5]
    cfg, ocfg, model, W, data, *_ = _setup(dev, torch.float32, lens, ulens)
    model.cfg.dropout = 0.2
    model.use_pred_stream = False

    def loss_at(flat):
        model.ps.flat.copy_(flat)
        model.ps.refresh_shadow()
        model._drop_epoch = 41  # _forward bumps it to 42 every time -> identical masks
        for k, v in model.ps.state.items():  # undo the moving-average side effect
            v.copy_(state0[k])
        return float(model.loss_and_backward(data, True, (None, None), want_backward=False).double().mean())

    state0 = {k: v.clone() for k, v in model.ps.state.items()}
    base = model.ps.flat.clone()
    model._drop_epoch = 41
    model.zero_grad()
    model.loss_and_backward(data, True, (None, None))
    g = model.ps.grad.clone()
    gen = torch.Generator(device="cpu").manual_seed(0)
    direction = torch.randn(base.numel(), generator=gen).to(dev)
    direction *= (base != 0).float()  # stay inside real parameters (alignment padding is zero)
    eps = 2e-3
    fd = (loss_at(base + eps * direction) - loss_at(base - eps * direction)) / (2 * eps)
    an = float((g.double() * direction.double()).sum())
    assert abs(fd - an) < 5e-2 * max(1.0, abs(an)), (fd, an)


@pytest.mark.parametrize("dtype,head", [(torch.float32, 8), (torch.bfloat16, 8), (torch.bfloat16, 64)])
def test_native_block_executor_matches_per_kernel_host_path(dev, dtype, head):
    """csrc/block.hip queues the same kernels as the Python per-kernel path: identical loss / gradients (up to the f32
    atomics of the split-K weight gradients), with dropout ON (same counter-based masks) and ragged lengths."""
    lens, ulens = [4000, 2700, 3300], [6, 3, 5]
    cfg = configs.conformer_tiny(dropout=0.1, head_size=head, dmodel=32 if head == 8 else 64, filters=32)
    rng = np.random.default_rng(3)
    B, N, U = len(lens), 4000, 6
    sig = np.clip(rng.standard_normal((B, N)) * 0.1, -1, 1).astype(np.float32)
    labels = rng.integers(1, cfg.vocab_size, (B, U)).astype(np.int32)
    for b, u in enumerate(ulens):
        labels[b, u:] = 0
    preds = np.concatenate([np.zeros((B, 1), np.int32), labels], 1)
    data = TrainData(
        TrainInput(torch.from_numpy(sig), torch.tensor(lens, dtype=torch.int32), torch.from_numpy(preds),
                   torch.tensor([u + 1 for u in ulens], dtype=torch.int32)),
        TrainLabel(torch.from_numpy(labels), torch.tensor(ulens, dtype=torch.int32)))
    out = {}
    for native in (False, True):
        model = ConformerTransducer(cfg, dev, dtype=dtype, seed=5)
        model.native_blocks = native
        masks = model.draw_specaugment([int(n) for n in R.get_nframes(lens)])
        model.zero_grad()
        costs = model.loss_and_backward(data, True, masks)
        torch.cuda.synchronize()
        out[native] = (costs.cpu().numpy(), model.ps.grad.cpu().numpy(), model.ps.state["enc/block1/conv/bn/mm"].cpu().numpy())
        # eval-mode forward through the same executor
        model.native_blocks = native
        logits, _, _ = model._forward(data.inputs, False, None)
        out[native] += (logits.float().cpu().numpy(),)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=tol)
    g0, g1 = out[False][1], out[True][1]
    np.testing.assert_allclose(g1, g0, rtol=tol, atol=tol * float(np.abs(g0).max()))
    # moving mean of a zero-mean activation: absolute scale ~1e-5, f32 atomics order differs between runs
    np.testing.assert_allclose(out[True][2], out[False][2], rtol=1e-5 if dtype == torch.float32 else 1e-2, atol=1e-7 if dtype == torch.float32 else 2e-5)
    np.testing.assert_allclose(out[True][3], out[False][3], rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_step_with_dropout_and_specaugment_on_matches_oracle(dev, dtype):
    """The whole train step with Dropout (rate 0.1) AND SpecAugment ON, as `bench.py` runs it: the product's
    masks are a pure function of (seed, element index), so the masks of this very step are regenerated with tfasr_dropout on a tensor
    of ones and injected into the oracle at the reference's Dropout sites (conformer.py:80-88,193,353,594,682); the SpecAugment draws
    are the product's own.  Logits, loss and every gradient against the oracle; native block executor and per-kernel host path."""
    from tensorflowasr_amd import kernels as K

    lens, ulens = [4000, 2500, 3100], [6, 3, 5]
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, dtype, lens, ulens, dropout=0.1)
    masks = model.draw_specaugment([int(n) for n in R.get_nframes(lens)])
    p = 0.1
    cache = {}

    def drop_for(epoch):
        def drop(site, y):
            if site is None:
                return y
            key = (epoch, site, tuple(y.shape))
            if key not in cache:
                ones = torch.ones(y.shape, dtype=torch.float32, device=dev)  # (f32: the exact 1 / (1 - p) the epilogues multiply by)
                cache[key] = K.dropout(ones.view(-1), p, (epoch * 8192 + site) & 0x7FFFFFFFFFFF).float().cpu().view(y.shape)
            m = cache[key]
            assert 0.8 < float((m > 0).float().mean()) < 0.97 and abs(float(m.max()) - 1.0 / (1.0 - p)) < 1e-2
            return y * m.to(y.dtype)
        return drop

    tol = 2e-3 if dtype == torch.float32 else 6e-2
    for native in (True, False):
        model.native_blocks = native
        epoch = model._drop_epoch + 1  # the forward pass below draws its masks under this epoch
        ref_logits, elen, ref_loss, ref_grads, _ = _oracle_step(ocfg, W, sig, lens, preds, ulens, labels, (masks[0].numpy(), masks[1].numpy()),
                                                                drop=drop_for(epoch))
        # the masks matter: without them the oracle's logits differ
        plain = _oracle_step(ocfg, W, sig, lens, preds, ulens, labels, (masks[0].numpy(), masks[1].numpy()))[0]
        assert float((plain - ref_logits).abs().max()) > 1e-2
        model.zero_grad()
        costs = model.loss_and_backward(data, True, masks).float().cpu().numpy()
        torch.cuda.synchronize()
        assert model._drop_epoch == epoch
        np.testing.assert_allclose(costs, ref_loss, rtol=1e-3 if dtype == torch.float32 else 3e-2)
        mine = model.ps.export_keras(model.ps.grad)
        num = sum(float((mine[k].double() - ref_grads[k].double()).pow(2).sum()) for k in ref_grads if k in mine)
        den = sum(float(ref_grads[k].double().pow(2).sum()) for k in ref_grads if k in mine)
        err = (num / den) ** 0.5
        assert err < (2e-3 if dtype == torch.float32 else 0.15), err
