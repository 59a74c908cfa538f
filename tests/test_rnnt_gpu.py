"""GPU parity: HIP RNN-T loss (through the C ABI) vs the oracle, reference-source goldens, and size-independent properties."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import rnnt_ref
from tensorflowasr_amd import kernels

pytestmark = pytest.mark.gpu


def _run(dev, logits, labels, ul, tl, dtype=torch.float32, grad_scale=None, inplace=False):
    lg = torch.from_numpy(logits).to(dev).to(dtype).contiguous()
    gs = None if grad_scale is None else torch.from_numpy(grad_scale.astype(np.float32)).to(dev)
    costs, grads = kernels.rnnt_loss_fwd_bwd(
        lg, torch.from_numpy(labels.astype(np.int32)).to(dev), torch.from_numpy(np.asarray(ul, np.int32)).to(dev),
        torch.from_numpy(np.asarray(tl, np.int32)).to(dev), grad_scale=gs, grads=lg if inplace else None)
    torch.cuda.synchronize()
    return costs.cpu().numpy(), grads.float().cpu().numpy()


def test_goldens_from_reference_source(dev, golden_dir):
    fs = sorted(glob.glob(os.path.join(golden_dir, "rnnt_reference_*.npz")))
    assert fs
    for f in fs:
        g = np.load(f)
        loss, grads = _run(dev, g["logits"], g["labels"], g["label_len"], g["logit_len"])
        # north_star tolerance: RNN-T loss within 1e-3 relative; f32 path is far tighter
        np.testing.assert_allclose(loss, g["loss"], rtol=1e-5, atol=1e-5, err_msg=f)
        np.testing.assert_allclose(grads, g["grads"], rtol=1e-4, atol=2e-5, err_msg=f)


@pytest.mark.parametrize("shape", [(2, 5, 3, 4), (3, 33, 17, 40), (4, 50, 20, 1000), (2, 70, 66, 129), (1, 9, 1, 7)])
def test_vs_oracle_ragged(dev, shape):
    B, T, U, V = shape
    rng = np.random.default_rng(B * 1000 + T)
    logits = (rng.standard_normal((B, T, U + 1, V)) * 1.5).astype(np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    tl = rng.integers(max(1, T // 2), T + 1, B).astype(np.int32)
    ul = rng.integers(0, U + 1, B).astype(np.int32)
    tl[0], ul[0] = T, U
    tl, ul = rnnt_ref.clamp_lengths(np.minimum(tl, T), ul)
    tl = np.minimum(tl, T)
    scale = rng.uniform(0.5, 2.0, B)
    ref_loss, ref_g = rnnt_ref.rnnt_loss_and_grad(logits, labels, ul, tl)
    loss, grads = _run(dev, logits, labels, ul, tl, grad_scale=scale)
    np.testing.assert_allclose(loss, ref_loss, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(grads, ref_g * scale[:, None, None, None], rtol=5e-4, atol=2e-5)  # __expf in the grad pass
    # in-place variant gives the same gradient
    _, g2 = _run(dev, logits, labels, ul, tl, grad_scale=scale, inplace=True)
    np.testing.assert_array_equal(g2, grads)


def test_bf16_within_north_star_tolerance(dev):
    B, T, U, V = 4, 40, 12, 256
    rng = np.random.default_rng(7)
    logits = rng.standard_normal((B, T, U + 1, V)).astype(np.float32)
    lb = torch.from_numpy(logits).to(torch.bfloat16).float().numpy()  # oracle sees the same rounded inputs
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    tl = np.array([40, 33, 40, 21], np.int32)
    ul = np.array([12, 5, 0, 12], np.int32)
    ref_loss, ref_g = rnnt_ref.rnnt_loss_and_grad(lb, labels, ul, tl)
    loss, grads = _run(dev, logits, labels, ul, tl, dtype=torch.bfloat16)
    np.testing.assert_allclose(loss, ref_loss, rtol=1e-3)  # BASELINE.json: within 1e-3 relative
    np.testing.assert_allclose(grads, ref_g, rtol=2e-2, atol=4e-3)  # bf16 storage of the gradient


def test_full_size_properties(dev):
    """BASELINE cfg2 lattice (B=32,T=250,U1=65,V=1000): size-independent invariants instead of an O(N) oracle pass.
    (i) sum_v grad = 0 on every node, (ii) grad == 0 outside the valid lattice, (iii) blank-terminal gradient,
    (iv) sum over the lattice of grad wrt blank along any full path cut = -1 per t (flow conservation),
    (v) a subset of utterances checked against the oracle."""
    B, T, U, V = 32, 250, 64, 1000
    g = torch.Generator(device="cpu").manual_seed(3)
    logits = torch.randn(B, T, U + 1, V, generator=g)
    labels = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32)
    ul = torch.randint(32, 65, (B,), generator=g, dtype=torch.int32)
    tl = torch.full((B,), T, dtype=torch.int32)
    tl[1] = 200
    lg = logits.to(dev)
    costs, grads = kernels.rnnt_loss_fwd_bwd(lg, labels.to(dev), ul.to(dev), tl.to(dev))
    torch.cuda.synchronize()
    assert torch.isfinite(costs).all()
    s = grads.sum(-1)
    assert s.abs().max().item() < 1e-4
    # outside lattice
    tt = torch.arange(T, device=dev)[None, :, None]
    uu = torch.arange(U + 1, device=dev)[None, None, :]
    outside = (tt >= tl.to(dev)[:, None, None]) | (uu > ul.to(dev)[:, None, None])
    assert grads[outside].abs().max().item() == 0.0
    # flow conservation: for each t < Tl-1 the total blank-transition mass leaving row t is exactly 1
    gb = torch.zeros(B, T, U + 1, device=dev)
    p = torch.softmax(lg, -1)
    # dL/dlogit_blank = gb - p0*(gb+gt)  and  sum_v grad = 0 => recover gb+gt from any non-blank/non-label v is noisy;
    # use oracle on 2 utterances instead for exact values:
    for b in (0, 1):
        rl, rg = rnnt_ref.rnnt_loss_and_grad(logits[b:b + 1].numpy(), labels[b:b + 1].numpy(), ul[b:b + 1].numpy(),
                                             tl[b:b + 1].numpy(), np.float32)
        np.testing.assert_allclose(costs[b].item(), rl[0], rtol=1e-5)
        np.testing.assert_allclose(grads[b].cpu().numpy(), rg[0], rtol=1e-3, atol=2e-5)
    del p, gb


def test_reference_smoke_shape(dev):
    """The reference's own smoke shape tests/test_rnnt_loss.py:6-10 (B=1,T=743,U=200,V=1000), labels = arange(U)."""
    B, T, U, V = 1, 743, 200, 1000
    g = torch.Generator(device="cpu").manual_seed(0)
    logits = torch.randn(B, T, U + 1, V, generator=g)
    labels = torch.arange(U, dtype=torch.int32)[None, :].contiguous()
    ul = torch.tensor([U], dtype=torch.int32)
    tl = torch.tensor([T], dtype=torch.int32)
    costs, grads = kernels.rnnt_loss_fwd_bwd(logits.to(dev), labels.to(dev), ul.to(dev), tl.to(dev))
    rl, rg = rnnt_ref.rnnt_loss_and_grad(logits.numpy(), labels.numpy(), ul.numpy(), tl.numpy(), np.float64)
    np.testing.assert_allclose(costs.cpu().numpy(), rl, rtol=1e-5)
    np.testing.assert_allclose(grads.cpu().numpy(), rg, rtol=1e-2, atol=1e-4)  # 943-diagonal f32 lattice (like the f32 TF reference) + __expf


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_packed_lattice_equals_dense(dev, dtype):
    """tfasr_rnnt_loss_packed on the valid nodes only == the dense entry (padded nodes have zero gradient)."""
    B, T, U, V = 5, 23, 9, 64
    rng = np.random.default_rng(9)
    logits = rng.standard_normal((B, T, U + 1, V)).astype(np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    tl = np.array([23, 11, 17, 1, 20], np.int32)
    ul = np.array([9, 3, 0, 1, 9], np.int32)
    scale = rng.uniform(0.5, 2, B).astype(np.float32)
    dense = torch.from_numpy(logits).to(dev).to(dtype)
    costs_d, g_d = kernels.rnnt_loss_fwd_bwd(dense, torch.from_numpy(labels).to(dev), torch.from_numpy(ul).to(dev),
                                             torch.from_numpy(tl).to(dev), grad_scale=torch.from_numpy(scale).to(dev))
    off = np.zeros(B + 1, np.int64)
    off[1:] = np.cumsum(tl.astype(np.int64) * (ul + 1))
    packed = torch.cat([dense[b, :tl[b], :ul[b] + 1].reshape(-1, V) for b in range(B)]).contiguous()
    costs_p, g_p = kernels.rnnt_loss_packed(packed, torch.from_numpy(labels).to(dev), torch.from_numpy(ul).to(dev), torch.from_numpy(tl).to(dev),
                                            torch.from_numpy(off).to(dev), int(off[-1]), T, grad_scale=torch.from_numpy(scale).to(dev))
    torch.cuda.synchronize()
    np.testing.assert_allclose(costs_p.cpu().numpy(), costs_d.cpu().numpy(), rtol=1e-6)
    want = torch.cat([g_d[b, :tl[b], :ul[b] + 1].reshape(-1, V) for b in range(B)])
    np.testing.assert_array_equal(g_p.float().cpu().numpy(), want.float().cpu().numpy())
    ref_loss, _ = rnnt_ref.rnnt_loss_and_grad(dense.float().cpu().numpy(), labels, ul, tl)
    np.testing.assert_allclose(costs_p.cpu().numpy(), ref_loss, rtol=1e-3 if dtype == torch.bfloat16 else 1e-5)


def test_joint_projection_statistics_256_row_tiles(dev):
    """The same statistics from gemm_big.h's epilogue (taken in the MFMA C layout with DPP row reductions): a joint of bench size
    (150 000 packed rows x 1000 classes -> 2344 tiles of 256 x 256) against torch on the f32 product."""
    from tensorflowasr_amd import kernels as K

    g = torch.Generator().manual_seed(5)
    total, V, J = 150001, 1000, 320
    h = (torch.randn(total, J, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    W = (torch.randn(J, V, generator=g) * 0.3).to(dev).to(torch.bfloat16)
    bias = torch.randn(V, generator=g).to(dev)
    row_label = torch.randint(-1, V, (total,), generator=g, dtype=torch.int32).to(dev)
    parts = -(-V // 128) * 2
    lse_part = torch.full((total, parts, 2), float("nan"), dtype=torch.float32, device=dev)
    pick = torch.full((total, 2), float("nan"), dtype=torch.float32, device=dev)
    logits = torch.empty(total, V, dtype=torch.bfloat16, device=dev)
    K.gemm(h, W, logits, total, V, J, J, V, V, bias=bias, lse=(lse_part, row_label, pick))
    torch.cuda.synchronize()
    x32 = h.float() @ W.float() + bias
    assert (logits.float() - x32).abs().max().item() < 0.25
    assert not torch.isnan(lse_part).any()
    m, s = lse_part[..., 0], lse_part[..., 1]
    mx = m.max(dim=1).values
    lse = mx + torch.log((s * torch.exp(m - mx[:, None])).sum(1))
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(x32, 1).cpu().numpy(), rtol=1e-5, atol=3e-5)
    np.testing.assert_allclose(pick[:, 0].cpu().numpy(), x32[:, 0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    has = row_label >= 0
    want = x32[has, row_label[has].long()]
    np.testing.assert_allclose(pick[has, 1].cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("V", [1000, 256, 29 * 8])
def test_joint_projection_epilogue_statistics_feed_the_loss(dev, V):
    """tfasr_gemm_args.lse_part / pick: the vocabulary projection's epilogue emits max / sum-exp per 64-column slice and the blank /
    label logits from its f32 accumulators; tfasr_rnnt_loss_packed_stats then skips the first pass over the logits.  Statistics vs
    torch on the f32 product, loss vs the plain route and vs the oracle (packed ragged lattice, V not a multiple of 128)."""
    from oracle import rnnt_ref
    from tensorflowasr_amd import kernels as K

    rng = np.random.default_rng(V)
    B, T, U, J = 3, 37, 9, 64
    tl, ul = np.array([37, 20, 31], np.int32), np.array([9, 4, 0], np.int32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    off = np.zeros(B + 1, np.int64)
    off[1:] = np.cumsum(tl * (ul + 1))
    total = int(off[-1])
    h = torch.from_numpy(rng.standard_normal((total, J)).astype(np.float32)).to(dev).to(torch.bfloat16)
    W = torch.from_numpy((rng.standard_normal((J, V)) * 0.3).astype(np.float32)).to(dev).to(torch.bfloat16)
    bias = torch.from_numpy(rng.standard_normal(V).astype(np.float32)).to(dev)
    lab_d, ul_d, tl_d, off_d = (torch.from_numpy(x).to(dev) for x in (labels, ul, tl, off))
    parts = -(-V // 128) * 2
    lse_part = torch.full((total, parts, 2), float("nan"), dtype=torch.float32, device=dev)
    pick = torch.full((total, 2), float("nan"), dtype=torch.float32, device=dev)
    row_label = K.rnnt_row_labels(lab_d, ul_d, tl_d, off_d, total, T, V)
    logits = torch.empty(total, V, dtype=torch.bfloat16, device=dev)
    K.gemm(h, W, logits, total, V, J, J, V, V, bias=bias, lse=(lse_part, row_label, pick))
    plain = K.matmul(h, W, bias=bias)
    torch.cuda.synchronize()
    assert torch.equal(plain, logits)  # the stored logits are those of the plain epilogue
    x32 = h.float() @ W.float() + bias  # what the accumulators hold (f32 sum of bf16 products)
    m, s = lse_part[..., 0], lse_part[..., 1]
    mx = m.max(dim=1).values
    lse = mx + torch.log((s * torch.exp(m - mx[:, None])).sum(1))
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(x32, 1).cpu().numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(pick[:, 0].cpu().numpy(), x32[:, 0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    rl = row_label.cpu().numpy()
    has = rl >= 0
    assert has.sum() == int((tl * ul).sum())
    np.testing.assert_allclose(pick[:, 1].cpu().numpy()[has], x32.cpu().numpy()[np.nonzero(has)[0], rl[has]], rtol=1e-5, atol=1e-5)
    # loss: fused statistics vs plain route vs oracle (on the f32 product)
    c_stats, g_stats = K.rnnt_loss_packed(logits.clone(), lab_d, ul_d, tl_d, off_d, total, T, stats=(lse_part, pick))
    c_plain, g_plain = K.rnnt_loss_packed(logits.clone(), lab_d, ul_d, tl_d, off_d, total, T)
    torch.cuda.synchronize()
    dense = np.zeros((B, T, U + 1, V), np.float32)
    x_np = x32.cpu().numpy()
    for b in range(B):
        n = tl[b] * (ul[b] + 1)
        dense[b, :tl[b], :ul[b] + 1] = x_np[off[b]:off[b] + n].reshape(tl[b], ul[b] + 1, V)
    ref_loss, _ = rnnt_ref.rnnt_loss_and_grad(dense, labels, ul, tl, np.float32)
    np.testing.assert_allclose(c_stats.cpu().numpy(), ref_loss, rtol=2e-4)       # f32 statistics: closer to the oracle than ...
    np.testing.assert_allclose(c_plain.cpu().numpy(), ref_loss, rtol=3e-3)       # ... statistics of the bf16-rounded logits
    np.testing.assert_allclose(g_stats.float().cpu().numpy(), g_plain.float().cpu().numpy(), rtol=5e-2, atol=2e-3)


def test_joint_recompute_variant_without_materialised_logits(dev):
    """SURVEY section 7 step 8 / 8(d) "recompute" variant of the joint + loss: the vocabulary projection emits ONLY the log-softmax
    statistics (D = NULL), tfasr_rnnt_loss_packed_coef turns the lattice into per-row coefficients, and a second projection re-computes
    the logit tile and stores the loss gradient from its epilogue (tfasr_gemm_args.rgrad_coef).  Against the materialised route (same
    costs bit for bit - both read the statistics of the f32 accumulators - gradients equal up to the bf16 rounding of the stored
    logits) and against the f64 oracle on the f32 product."""
    from tensorflowasr_amd import kernels as K

    rng = np.random.default_rng(3)
    B, T, U, V, J = 6, 150, 60, 1000, 128
    tl = np.array([150, 120, 150, 75, 140, 150], np.int32)
    ul = np.array([60, 31, 0, 60, 45, 59], np.int32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    off = np.zeros(B + 1, np.int64)
    off[1:] = np.cumsum(tl.astype(np.int64) * (ul + 1))
    total = int(off[-1])
    assert total >= 32768  # enough 256 x 256 tiles for the 256-row kernel (2 per CU)
    g = torch.Generator().manual_seed(1)
    h = torch.tanh(torch.randn(total, J, generator=g)).to(dev).to(torch.bfloat16)
    W = (torch.randn(J, V, generator=g) * 0.25).to(dev).to(torch.bfloat16)
    bias = (torch.randn(V, generator=g) * 0.5).to(dev)
    scale = torch.from_numpy(rng.uniform(0.5, 2, B).astype(np.float32)).to(dev)
    d = lambda a: torch.from_numpy(a).to(dev)
    lab_d, ul_d, tl_d, off_d = d(labels), d(ul), d(tl), d(off)
    row_label = K.rnnt_row_labels(lab_d, ul_d, tl_d, off_d, total, T, V)
    parts = -(-V // 128) * 2

    def stats_buffers():
        return torch.empty(total, parts, 2, dtype=torch.float32, device=dev), torch.empty(total, 2, dtype=torch.float32, device=dev)

    # materialised route
    part_a, pick_a = stats_buffers()
    logits = torch.empty(total, V, dtype=torch.bfloat16, device=dev)
    K.gemm(h, W, logits, total, V, J, J, V, V, bias=bias, lse=(part_a, row_label, pick_a))
    costs_a, g_a = K.rnnt_loss_packed(logits, lab_d, ul_d, tl_d, off_d, total, T, grad_scale=scale, stats=(part_a, pick_a))
    # recompute route: no logits tensor at all
    part_b, pick_b = stats_buffers()
    K.gemm(h, W, None, total, V, J, J, V, V, bias=bias, lse=(part_b, row_label, pick_b))
    costs_b, coef = K.rnnt_loss_packed_coef(lab_d, ul_d, tl_d, off_d, total, T, V, (part_b, pick_b), grad_scale=scale)
    g_b = torch.full((total, V), float("nan"), dtype=torch.bfloat16, device=dev)
    K.gemm(h, W, g_b, total, V, J, J, V, V, bias=bias, rgrad=(coef, row_label))
    torch.cuda.synchronize()
    has = row_label >= 0  # (pick[:, 1] is untouched where a row has no label)
    assert torch.equal(part_a, part_b) and torch.equal(pick_a[:, 0], pick_b[:, 0]) and torch.equal(pick_a[has, 1], pick_b[has, 1])
    np.testing.assert_array_equal(costs_a.cpu().numpy(), costs_b.cpu().numpy())
    ga, gb = g_a.float().cpu().numpy(), g_b.float().cpu().numpy()
    assert np.isfinite(gb).all()
    # oracle on the f32 product of the bf16 operands (dense lattice rebuilt from the packed rows)
    x = (h.float() @ W.float() + bias).cpu().numpy()
    dense = np.zeros((B, T, U + 1, V), np.float32)
    for b in range(B):
        dense[b, :tl[b], :ul[b] + 1] = x[off[b]:off[b + 1]].reshape(tl[b], ul[b] + 1, V)
    ref_loss, ref_g = rnnt_ref.rnnt_loss_and_grad(dense, labels, ul, tl)
    ref_gp = np.concatenate([ref_g[b, :tl[b], :ul[b] + 1].reshape(-1, V) for b in range(B)]) * np.repeat(scale.cpu().numpy(), np.diff(off))[:, None]
    np.testing.assert_allclose(costs_b.cpu().numpy(), ref_loss, rtol=2e-4)
    den = np.linalg.norm(ref_gp)
    ea, eb = np.linalg.norm(ga - ref_gp) / den, np.linalg.norm(gb - ref_gp) / den
    print(f"\n[joint recompute] gradient rel. L2 error vs the f64 oracle: materialised {ea:.3e}, recompute {eb:.3e}")
    assert eb < 5e-3 and eb <= ea * 1.05  # the recompute route sees unrounded logits: at least as accurate


def test_joint_recompute_variant_in_the_train_step(dev, monkeypatch):
    """TFASR_JOINT_RECOMPUTE=1 on a model whose lattice is large enough for the 256-row kernel: same losses, same gradients (bf16 noise)."""
    from tensorflowasr_amd import configs
    from tensorflowasr_amd.conformer import ConformerTransducer
    from tensorflowasr_amd.schemas import TrainData, TrainInput, TrainLabel

    cfg = configs.conformer_tiny(vocab_size=1000, joint_dim=128, dropout=0.0, time_masking={}, freq_masking={})
    rng = np.random.default_rng(0)
    B, N, U = 8, 64000, 50
    sig = (rng.standard_normal((B, N)) * 0.1).astype(np.float32)
    labels = rng.integers(1, 1000, (B, U)).astype(np.int32)
    preds = np.concatenate([np.zeros((B, 1), np.int32), labels], 1)
    data = TrainData(TrainInput(torch.from_numpy(sig), torch.full((B,), N, dtype=torch.int32), torch.from_numpy(preds), torch.full((B,), U + 1, dtype=torch.int32)),
                     TrainLabel(torch.from_numpy(labels), torch.full((B,), U, dtype=torch.int32)))
    out = {}
    for flag in (False, True):
        m = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
        m.joint_recompute = flag
        m.zero_grad()
        costs = m.loss_and_backward(data, True, (None, None)).float().cpu().numpy()
        torch.cuda.synchronize()
        out[flag] = (costs, m.ps.grad.clone().cpu())
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-3)  # (two forward passes: BatchNorm statistics are atomic sums, bf16 activations)
    a, b = out[True][1].double(), out[False][1].double()
    assert not torch.equal(a, b) and float((a - b).norm() / b.norm()) < 2e-2
