"""Row g1 of VERDICT r01: parity of the path that bench.py TIMES — bf16 storage, fused relative attention (head 64), native
block executor, packed lattice — against the f32 torch-CPU oracle (oracle/conformer_ref.py + oracle/rnnt_ref.py) at the
dimensions BASELINE.json quotes:

  * Conformer-S (examples/models/transducer/conformer/small.yml.j2:26-66: d=144, dh=36, 16 blocks, k=31, E=P=J=320, V=1000),
    10 s utterances, U in [32, 64] (BASELINE configs[1]; SURVEY.md section 8(d) cfg2 seeds).  The CPU oracle bounds the batch
    (8 utterances, ~35 s on 8 cores) — the batch is shrunk, the dimensions are not.
  * Conformer-M dimensions (d=256, dh=64, k=31, E=P=J=640), ragged LibriSpeech-shaped lengths, T' = 462 and T' = 743
    (the two padded lengths of bench.py's batches), fewer blocks (the oracle's cost is linear in the block count and every
    block runs the same kernels).

BASELINE.json's tolerance: RNN-T loss within 1e-3 relative of the reference CPU (f32) path; greedy tokens bit-exact.
What is achieved is asserted below and tabulated in DESIGN.md section 4.
"""
import numpy as np
import pytest
import torch

from oracle import conformer_ref as R
from oracle import rnnt_ref
from tensorflowasr_amd import configs
from tensorflowasr_amd.conformer import ConformerTransducer
from tensorflowasr_amd.schemas import PredictInput, TrainData, TrainInput, TrainLabel

pytestmark = pytest.mark.gpu


def _make(dev, size, dtype, nsamp, ulens, U, blocks=None, seed=0, scale_bias=0.1):
    """Same weights on both sides (oracle initialiser -> import_keras), BASELINE-shaped synthetic batch."""
    over = {} if blocks is None else dict(num_blocks=blocks)
    cfg = (configs.conformer_s if size == "S" else configs.conformer_m)(dropout=0.0, **over)
    ocfg = R.conformer_config(size, cfg.vocab_size)
    if blocks is not None:
        ocfg["num_blocks"] = blocks
    model = ConformerTransducer(cfg, dev, dtype=dtype, seed=seed)
    W = R.init_weights(ocfg, seed=3, scale_bias=scale_bias)
    model.ps.import_keras(W)
    B, N = len(nsamp), int(max(nsamp))
    sig = np.clip(np.random.default_rng(0).standard_normal((B, N)).astype(np.float32) * 0.1, -1, 1)
    for b, n in enumerate(nsamp):
        sig[b, n:] = 0.0
    labels = np.random.default_rng(1).integers(1, cfg.vocab_size, (B, U)).astype(np.int32)
    for b, u in enumerate(ulens):
        labels[b, u:] = 0
    preds = np.concatenate([np.zeros((B, 1), np.int32), labels], 1)
    data = TrainData(
        TrainInput(torch.from_numpy(sig), torch.tensor(nsamp, dtype=torch.int32), torch.from_numpy(preds),
                   torch.tensor([u + 1 for u in ulens], dtype=torch.int32)),
        TrainLabel(torch.from_numpy(labels), torch.tensor(ulens, dtype=torch.int32)))
    return cfg, ocfg, model, W, data, sig, labels, preds


def _oracle(ocfg, W, sig, nsamp, preds, ulens, labels, want_grads=True):
    torch.set_num_threads(min(16, torch.get_num_threads() or 8))
    Wg = {k: v.clone().requires_grad_(want_grads and R.is_trainable(k)) for k, v in W.items()}
    feat = R.log_mel(sig, ocfg)
    with torch.set_grad_enabled(want_grads):
        logits, elen = R.transducer_forward(torch.from_numpy(feat), R.get_nframes(nsamp), torch.from_numpy(preds),
                                            torch.tensor([u + 1 for u in ulens]), Wg, ocfg, training=True)
    tl, ul = rnnt_ref.clamp_lengths(elen.numpy(), np.asarray(ulens))
    loss, g = rnnt_ref.rnnt_loss_and_grad(logits.detach().numpy(), labels, ul, np.minimum(tl, logits.shape[1]), np.float32)
    grads = None
    if want_grads:
        logits.backward(torch.from_numpy(g / len(nsamp)).to(logits.dtype))
        grads = {k: v.grad for k, v in Wg.items() if v.requires_grad}
    return loss, grads, elen


def _grad_rel_l2(model, ref_grads):
    mine = model.ps.export_keras(model.ps.grad)
    num = sum(float(((mine[k].double() - g.double()) ** 2).sum()) for k, g in ref_grads.items())
    den = sum(float((g.double() ** 2).sum()) for g in ref_grads.values())
    return (num / den) ** 0.5


def _report(name, costs, ref_loss, extra=""):
    rel = np.abs(costs - ref_loss) / np.abs(ref_loss)
    print(f"\n[g1] {name}: loss rel err max {rel.max():.3e} mean {rel.mean():.3e} (loss ~{np.mean(ref_loss):.1f}) {extra}")
    return rel


# --------------------------------------------------------------------------------------------- BASELINE configs[1]
@pytest.mark.timeout(900)
def test_conformer_s_16_blocks_10s_loss_and_gradients_vs_oracle(dev):
    """Conformer-S exactly as small.yml.j2 builds it (16 blocks), 8 x 10 s, labels in [32, 64]: the bf16 step (what
    `bench.py --model S` times) and the f32 step against the f32 CPU oracle."""
    B, U = 8, 64
    nsamp = [160000] * B
    ulens = [int(u) for u in np.random.default_rng(2).integers(32, 65, B)]
    out = {}
    ref_loss = ref_grads = None
    for dtype in (torch.bfloat16, torch.float32):
        cfg, ocfg, model, W, data, sig, labels, preds = _make(dev, "S", dtype, nsamp, ulens, U)
        if ref_loss is None:
            ref_loss, ref_grads, _ = _oracle(ocfg, W, sig, nsamp, preds, ulens, labels)
        assert model.native_blocks
        model.zero_grad()
        costs = model.loss_and_backward(data, True, (None, None)).float().cpu().numpy()
        torch.cuda.synchronize()
        rel = _report(f"Conformer-S 16 blocks B={B} T'=250 {str(dtype)[6:]}", costs, ref_loss)
        gl2 = _grad_rel_l2(model, ref_grads)
        print(f"[g1]   gradient relative L2 error over all variables: {gl2:.3e}")
        out[dtype] = (rel.max(), gl2)
        del model
        torch.cuda.empty_cache()
    assert out[torch.float32][0] < 1e-4 and out[torch.float32][1] < 2e-3, out
    assert out[torch.bfloat16][0] < 1e-3, out  # BASELINE.json: RNN-T loss within 1e-3 relative
    assert out[torch.bfloat16][1] < 2e-2, out


# --------------------------------------------------------------------------------------------- BASELINE configs[2] dimensions
@pytest.mark.timeout(900)
@pytest.mark.parametrize("case", ["T462_B4", "T743_B2", "T150_B3_L16", "T462_B2_L16", "T743_B1_L16"])
def test_conformer_m_dims_ragged_bf16_fused_path_vs_oracle(dev, case):
    """d=256, dh=64, k=31, J=640 with ragged lengths at the padded lengths of bench.py's two batches: fused attention +
    native executor + packed lattice (bf16) against the oracle, which applies the reference's padded-query mask and the
    per-sample relative-PE roll (positional_encoding.py:152-172)."""
    if case == "T462_B4":
        nsamp, ulens, U, blocks = [295600, 201000, 131072, 20800], [68, 46, 30, 5], 68, 3
    elif case == "T743_B2":
        nsamp, ulens, U, blocks = [475200, 160000], [40, 37], 40, 2
    elif case == "T743_B1_L16":  # ALL 16 blocks at bench.py's OTHER padded length (T' = 743), one utterance (the oracle's cost sets B)
        nsamp, ulens, U, blocks = [475200], [40], 40, 16
    elif case == "T462_B2_L16":  # ALL 16 blocks at one of bench.py's padded lengths (T' = 462), two ragged utterances
        nsamp, ulens, U, blocks = [295600, 131072], [68, 30], 68, 16
    else:  # ALL 16 blocks of the model bench.py times (depth: bf16 drift through the whole encoder), shorter utterances, ragged incl. padded blocks
        nsamp, ulens, U, blocks = [96000, 70000, 30000], [22, 16, 7], 22, 16
    cfg, ocfg, model, W, data, sig, labels, preds = _make(dev, "M", torch.bfloat16, nsamp, ulens, U, blocks=blocks)
    assert model._fused_attention() and model.native_blocks
    ref_loss, ref_grads, elen = _oracle(ocfg, W, sig, nsamp, preds, ulens, labels)
    assert int(elen.max()) == {"T462_B4": 462, "T743_B2": 743, "T150_B3_L16": 150, "T462_B2_L16": 462, "T743_B1_L16": 743}[case]
    model.zero_grad()
    costs = model.loss_and_backward(data, True, (None, None)).float().cpu().numpy()
    torch.cuda.synchronize()
    rel = _report(f"Conformer-M dims {blocks} blocks {case} bf16 fused/native/packed", costs, ref_loss)
    gl2 = _grad_rel_l2(model, ref_grads)
    print(f"[g1]   gradient relative L2 error over all variables: {gl2:.3e}")
    assert rel.max() < 1e-3, rel
    assert gl2 < 2e-2, gl2
    # the f32 parity mode of the same model (different kernels: exact-f32 MFMA GEMMs, unfused attention) on the same input
    cfg, ocfg, model32, W, data, *_ = _make(dev, "M", torch.float32, nsamp, ulens, U, blocks=blocks)
    model32.zero_grad()
    c32 = model32.loss_and_backward(data, True, (None, None)).float().cpu().numpy()
    torch.cuda.synchronize()
    rel32 = _report(f"Conformer-M dims {blocks} blocks {case} f32", c32, ref_loss)
    g32 = _grad_rel_l2(model32, ref_grads)
    print(f"[g1]   f32 gradient relative L2 error: {g32:.3e}")
    assert rel32.max() < 1e-4 and g32 < 2e-3


# --------------------------------------------------------------------------------------------- attention at the bench length
@pytest.mark.timeout(600)
def test_mhsa_module_T743_B8_fused_vs_oracle(dev):
    """MHSAModule (conformer.py:209-239) at T' = 743, B = 8, H = 4, dh = 64, ragged: LayerNorm -> fused qkv projection ->
    relattn_fused_fwd/bwd -> output projection, forward output and the gradients w.r.t. the input and every parameter of
    the module against the oracle's rel_mhsa under torch autograd."""
    B, T, d, H, dh = 8, 743, 256, 4, 64
    lens = [743, 743, 700, 601, 462, 333, 100, 17]
    cfg = configs.conformer_m(dropout=0.0, num_blocks=1)
    ocfg = R.conformer_config("M")
    ocfg["num_blocks"] = 1
    model = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=1)
    W = R.init_weights(ocfg, seed=5, scale_bias=0.2)
    model.ps.import_keras(W)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, d, generator=g)
    dy = torch.randn(B, T, d, generator=g) * 0.1
    pfx = "enc/block0/mhsa/"
    # oracle
    xr = x.clone().requires_grad_(True)
    names = [pfx + n for n in ("ln/g", "ln/b", "q/w", "q/b", "k/w", "k/b", "v/w", "v/b", "pos/w", "pos/b", "o/w", "o/b")] + ["enc/u", "enc/v"]
    Wg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in W.items()}
    pe, _ = R.relative_position_encoding(T, d, lens)
    yr = R.mhsa_module(xr, pe, Wg, pfx, H, dh, lens, Wg["enc/u"], Wg["enc/v"], use_mask=True)
    yr.backward(dy)
    # HIP path (per-kernel host path of the same kernels the native executor queues)
    ctx = {}
    xd = x.to(dev).to(torch.bfloat16).view(B * T, d)
    elen_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
    model.zero_grad()
    y = model._mhsa_fwd(xd, pfx, B, T, elen_dev, ctx, 0, False)
    dx = model._mhsa_bwd(dy.to(dev).to(torch.bfloat16).view(B * T, d), pfx, B, T, elen_dev, ctx)
    torch.cuda.synchronize()
    yv, dxv = y.float().cpu().view(B, T, d), dx.float().cpu().view(B, T, d)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())

    e_y, e_dx = rel(yv, yr.detach()), rel(dxv, xr.grad)
    mine = model.ps.export_keras(model.ps.grad)
    # the key bias and the positional bias shift every score of a row by the same amount: their gradient is analytically
    # ZERO (rounding noise on both sides), so errors are measured against max(|g|, 1e-3 x the largest gradient of the module)
    gmax = max(float(Wg[k].grad.double().norm()) for k in names if Wg[k].grad is not None)
    errs = {k: float((mine[k].double() - Wg[k].grad.double()).norm() / max(float(Wg[k].grad.double().norm()), 1e-3 * gmax))
            for k in names if Wg[k].grad is not None}
    print(f"\n[g1] MHSA T'=743 B=8 bf16 fused: y rel L2 {e_y:.3e}, dx rel L2 {e_dx:.3e}, worst parameter gradient "
          f"{max(errs, key=errs.get)} {max(errs.values()):.3e}")
    assert e_y < 1e-2 and e_dx < 2e-2, (e_y, e_dx)
    assert max(errs.values()) < 3e-2, errs


@pytest.mark.timeout(600)
@pytest.mark.parametrize("chunk,hist", [(None, None), (16, 64), (8, -1), (5, 3)])
def test_mhsa_module_S_heads36_fused_and_streaming_vs_oracle(dev, chunk, hist):
    """Row g2: the head size the reference ships (36, small.yml.j2:39) on the fused LDS-staged attention kernels (heads stored
    zero-padded to 64) and the streaming mask of small-streaming.yml.j2:26,38-39 (chunk 16 / history 64; also unlimited history and a
    window narrower than a key block, which exercises the skipped key blocks and rows whose first processed block is fully masked)
    inside those kernels.  MHSAModule (conformer.py:209-239) at Conformer-S dims, T' = 250, B = 8, ragged: forward output, input
    gradient and every parameter gradient against the oracle's rel_mhsa (compute_streaming_mask AND auto mask) under autograd."""
    B, T, d, H, dh = 8, 250, 144, 4, 36
    lens = [250, 250, 249, 200, 129, 64, 33, 7]
    cfg = configs.conformer_s(dropout=0.0, num_blocks=1, chunk_size=chunk, history_size=hist)
    ocfg = R.conformer_config("S")
    ocfg["num_blocks"] = 1
    model = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=1)
    assert model.ps.head_phys == 64 and model._fused_attention()
    W = R.init_weights(ocfg, seed=5, scale_bias=0.2)
    model.ps.import_keras(W)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, d, generator=g)
    dy = torch.randn(B, T, d, generator=g) * 0.1
    pfx = "enc/block0/mhsa/"
    xr = x.clone().requires_grad_(True)
    names = [pfx + n for n in ("ln/g", "ln/b", "q/w", "q/b", "k/w", "k/b", "v/w", "v/b", "pos/w", "pos/b", "o/w", "o/b")] + ["enc/u", "enc/v"]
    Wg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in W.items()}
    pe, _ = R.relative_position_encoding(T, d, lens)
    yr = R.mhsa_module(xr, pe, Wg, pfx, H, dh, lens, Wg["enc/u"], Wg["enc/v"], use_mask=True, chunk_size=chunk, history_size=hist)
    yr.backward(dy)
    if chunk:  # the mask matters
        with torch.no_grad():
            yfull = R.mhsa_module(x, pe, W, pfx, H, dh, lens, W["enc/u"], W["enc/v"], use_mask=True)
        assert float((yfull - yr.detach()).abs().max()) > 1e-2
    elen_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
    xd = x.to(dev).to(torch.bfloat16).view(B * T, d)
    dyd = dy.to(dev).to(torch.bfloat16).view(B * T, d)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())

    ctx = {}
    model.zero_grad()
    y = model._mhsa_fwd(xd, pfx, B, T, elen_dev, ctx, 0, False)
    dx = model._mhsa_bwd(dyd, pfx, B, T, elen_dev, ctx)
    torch.cuda.synchronize()
    e_y, e_dx = rel(y.float().cpu().view(B, T, d), yr.detach()), rel(dx.float().cpu().view(B, T, d), xr.grad)
    mine = model.ps.export_keras(model.ps.grad)
    gmax = max(float(Wg[k].grad.double().norm()) for k in names if Wg[k].grad is not None)
    errs = {k: float((mine[k].double() - Wg[k].grad.double()).norm() / max(float(Wg[k].grad.double().norm()), 1e-3 * gmax))
            for k in names if Wg[k].grad is not None}
    print(f"\n[g2] MHSA S dims T'=250 B=8 bf16 fused (heads 36 -> 64), chunk {chunk} history {hist}: y rel L2 {e_y:.3e}, dx rel L2 {e_dx:.3e}, "
          f"worst parameter gradient {max(errs, key=errs.get)} {max(errs.values()):.3e}")
    assert np.isfinite(y.float().cpu().numpy()).all() and np.isfinite(dx.float().cpu().numpy()).all()
    assert e_y < 1e-2 and e_dx < 2e-2, (e_y, e_dx)
    assert max(errs.values()) < 3e-2, errs
    # the native block executor queues the same kernels: one whole block, forward + backward, equals the per-kernel host path
    outs = {}
    for native in (True, False):
        model.native_blocks = native
        model.zero_grad()
        cx = {}
        fn = model._block_fwd_native if native else model._block_fwd
        yb = fn(xd, 0, B, T, elen_dev, True, cx)
        dxb = model._block_bwd_native(dyd, 0, cx) if native else model._block_bwd(dyd, 0, B, T, elen_dev, cx)
        torch.cuda.synchronize()
        outs[native] = (yb.float().cpu(), dxb.float().cpu(), model.ps.grad.clone().cpu())
    assert rel(outs[True][0], outs[False][0]) < 2e-3 and rel(outs[True][1], outs[False][1]) < 5e-3
    assert rel(outs[True][2], outs[False][2]) < 5e-3


# --------------------------------------------------------------------------------------------- greedy tokens, bf16 vs f32 oracle
@pytest.mark.timeout(600)
@pytest.mark.parametrize("sharpen,blank_bias", [(6.0, 5.0), (10.0, 10.0)])
def test_conformer_s_greedy_tokens_bf16_and_f32_vs_f32_oracle(dev, sharpen, blank_bias):
    """Greedy search (base_transducer.py:496-712) on a seeded Conformer-S (16 blocks, random weights; the vocabulary
    projection is sharpened and the blank biased so that some utterances emit nothing, some a handful of tokens and some
    saturate their token buffer): the f32 model's tokens are bit-exact against the f32 oracle, and so are the bf16-trained
    model's in its default decode mode (f32 inference twin = what `bench.py --mode decode` times).  With precision="bf16" (the
    training kernels) the search arithmetic is still f32: tokens are bit-exact against the reference search applied to its own
    (bf16) encoder output; what differs from the all-f32 oracle is attributable to the encoder's bf16 rounding alone and is
    reported (agreement, margin at the first divergence)."""
    B = 4
    nsamp = [64000, 64000, 48000, 30000]
    ulens = [3] * B
    cfg, ocfg, model32, W, data, sig, labels, preds = _make(dev, "S", torch.float32, nsamp, ulens, 3, scale_bias=0.0)
    W = dict(W)
    W["joint/vocab/w"] = W["joint/vocab/w"] * sharpen
    W["joint/vocab/b"] = W["joint/vocab/b"].clone()
    W["joint/vocab/b"][0] += blank_bias
    model32.ps.import_keras(W)
    feat = R.log_mel(sig, ocfg)
    with torch.no_grad():
        enc_ref, elen = R.encoder(torch.from_numpy(feat)[..., None], R.get_nframes(nsamp), W, ocfg, training=False)
        tok_ref, _, _, _ = R.recognize_batch(enc_ref, elen.tolist(), W)
    inp = PredictInput(torch.from_numpy(sig), torch.tensor(nsamp, dtype=torch.int32))
    out32 = model32.recognize(inp)
    np.testing.assert_array_equal(out32.tokens.cpu().numpy(), tok_ref.numpy())
    per_utt = [int((tok_ref[b] != 0).sum()) for b in range(B)]
    assert sum(per_utt) > 10 and min(per_utt) < 10, per_utt
    model16 = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
    model16.ps.import_keras(W)
    # THE BENCHMARKED MODE (`bench.py --mode decode`, the default of recognize()): a bf16-trained model decodes on its f32 master
    # weights with the exact-f32 kernels (inference_twin) -> bit-exact token indices against the f32 oracle, batch and single variant
    assert model16.decode_precision == "f32"
    np.testing.assert_array_equal(model16.recognize(inp).tokens.cpu().numpy(), tok_ref.numpy())
    # the twin reads the live master buffers (no copy): a weight update is seen without a rebuild
    assert model16.inference_twin().ps.flat.data_ptr() == model16.ps.flat.data_ptr()
    out16 = model16.recognize(inp, precision="bf16")
    t16, tr = out16.tokens.cpu().numpy(), tok_ref.numpy()
    # (a) the search itself is exact: the reference loop applied to the bf16 model's OWN encoder output gives its tokens
    enc16, elen16 = model16.encode(torch.from_numpy(sig), torch.tensor(nsamp, dtype=torch.int32), precision="bf16")
    assert list(elen16) == elen.tolist()
    with torch.no_grad():
        tok_ref16, _, _, _ = R.recognize_batch(enc16.float().cpu(), elen.tolist(), W)
    np.testing.assert_array_equal(t16, tok_ref16.numpy())
    # (b) against the all-f32 oracle a bf16 ENCODER cannot promise identical arg-max decisions: report the agreement and
    # the f32 top-2 margin at each utterance's first disagreement (near-ties of the random-weight model)
    enc_err = float((enc16.float().cpu() - enc_ref).norm() / enc_ref.norm())
    agree = float((t16 == tr).mean())
    first_div = [int(np.argmax(t16[b] != tr[b])) if (t16[b] != tr[b]).any() else -1 for b in range(B)]
    print(f"\n[g1] greedy S x{sharpen:g} blank+{blank_bias:g}: f32 bit-exact (tokens per utterance {per_utt}); bf16: search bit-exact "
          f"on its own encoder output; encoder rel L2 error {enc_err:.3e}; token agreement with the f32 oracle {agree:.4f}, first "
          f"divergent column per utterance {first_div}")
    assert enc_err < 3e-2
    if any(d >= 0 for d in first_div):
        with torch.no_grad():
            margins = _oracle_margins(enc_ref, elen.tolist(), W)
        for b, col in enumerate(first_div):
            if col < 0:
                continue
            # column `col` holds emission number col - 1 (tokens start at column 2, base_transducer.py:518-520,545-552): the
            # decision that flipped lies among those taken after col - 2 emissions
            window = [m for m, ne in margins[b] if ne == max(col - 2, 0)] or [m for m, _ in margins[b]]
            print(f"[g1]   utterance {b}: smallest f32 top-2 log-prob margin around the first divergence {min(window):.4f} "
                  f"(logits sharpened x{sharpen:g})")
    # single-utterance variant (recognize_single: <= 3 symbols per frame)
    inp1 = PredictInput(torch.from_numpy(sig[:1]), torch.tensor(nsamp[:1], dtype=torch.int32))
    with torch.no_grad():
        tok1, _, _, _ = R.recognize_single(enc_ref[:1], elen.tolist()[:1], W)
    np.testing.assert_array_equal(model32.recognize(inp1).tokens.cpu().numpy(), tok1.numpy())
    np.testing.assert_array_equal(model16.recognize(inp1).tokens.cpu().numpy(), tok1.numpy())


@pytest.mark.timeout(600)
@pytest.mark.parametrize("sharpen,blank_bias", [(1.0, 2.5), (4.0, 12.0)])
def test_conformer_m_dims_greedy_tokens_bf16_and_f32_vs_f32_oracle(dev, sharpen, blank_bias):
    """The decode line `bench.py --mode decode --model M` quotes is the token-exact mode of a Conformer-M: the same bit-exactness claim as
    the Conformer-S test above at the M dimensions (d 256, 4 heads of 64, kernel 31, prediction 640, joint 640; 4 of the 16 blocks - the
    oracle's cost sets the depth), ragged batch incl. an utterance shorter than the others' padding: f32 model and the bf16-trained model's
    default mode (f32 inference twin) against the f32 oracle, batch variant and the batch-size-1 variant (recognize_single)."""
    B = 4
    nsamp = [64000, 52000, 40000, 24000]
    ulens = [3] * B
    cfg, ocfg, model32, W, data, sig, labels, preds = _make(dev, "M", torch.float32, nsamp, ulens, 3, blocks=4, scale_bias=0.0)
    W = dict(W)
    W["joint/vocab/w"] = W["joint/vocab/w"] * sharpen
    W["joint/vocab/b"] = W["joint/vocab/b"].clone()
    W["joint/vocab/b"][0] += blank_bias
    model32.ps.import_keras(W)
    feat = R.log_mel(sig, ocfg)
    with torch.no_grad():
        enc_ref, elen = R.encoder(torch.from_numpy(feat)[..., None], R.get_nframes(nsamp), W, ocfg, training=False)
        tok_ref, _, _, _ = R.recognize_batch(enc_ref, elen.tolist(), W)
    inp = PredictInput(torch.from_numpy(sig), torch.tensor(nsamp, dtype=torch.int32))
    np.testing.assert_array_equal(model32.recognize(inp).tokens.cpu().numpy(), tok_ref.numpy())
    per_utt = [int((tok_ref[b] != 0).sum()) for b in range(B)]
    print(f"\n[g1] greedy M dims (4 blocks) x{sharpen:g} blank+{blank_bias:g}: tokens per utterance {per_utt} (buffer {tok_ref.shape[1]})")
    # (1.0, 2.5): three rows saturate their token buffer, one emits a handful; (4.0, 12.0): a few tokens on two rows, none on the others
    assert sum(per_utt) > 4 and min(per_utt) < 10, per_utt
    model16 = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
    model16.ps.import_keras(W)
    assert model16.decode_precision == "f32"
    np.testing.assert_array_equal(model16.recognize(inp).tokens.cpu().numpy(), tok_ref.numpy())
    # batch size 1 -> recognize_single (<= 3 symbols per frame; NOT equivalent to the batch loop in the reference: both reproduced)
    one = PredictInput(torch.from_numpy(sig[1:2, :nsamp[1]].copy()), torch.tensor(nsamp[1:2], dtype=torch.int32))
    feat1 = R.log_mel(sig[1:2, :nsamp[1]], ocfg)
    with torch.no_grad():
        enc1, elen1 = R.encoder(torch.from_numpy(feat1)[..., None], R.get_nframes(nsamp[1:2]), W, ocfg, training=False)
        tok1, _, _, _ = R.recognize_single(enc1, elen1.tolist(), W)
    np.testing.assert_array_equal(model16.recognize(one).tokens.cpu().numpy(), tok1.numpy())


def _oracle_margins(enc, elen, W):
    """(top-2 log-probability margin, emissions so far) of every decision of the f32 greedy search, per utterance."""
    out = []
    for b in range(enc.shape[0]):
        T = int(elen[b])
        P = W["pred/lstm/rk"].shape[0]
        h = torch.zeros(1, P)
        c = torch.zeros(1, P)
        prev = torch.zeros(1, 1, dtype=torch.long)
        t, ms, n, ne = 0, [], 0, 0
        while n < 3 * T + 2:
            lsm, hn, cn = R._call_next(enc[b:b + 1, min(t, T - 1):min(t, T - 1) + 1], prev, h, c, W)
            top = torch.topk(lsm.view(-1), 2)
            ms.append((float(top.values[0] - top.values[1]), ne))
            k = int(top.indices[0])
            if k == 0:
                t += 1
            else:
                prev = torch.full((1, 1), k, dtype=torch.long)
                h, c = hn, cn
                ne += 1
            n += 1
        out.append(ms)
    return out
