"""GPU: the launches taken out of the Conformer blocks' dependent chain (tfasr_block_io.pext_pre / defer_pos_grad / ln_part_ext,
conformer.py `block_hoist`) change nothing: the positional tables computed ahead on the auxiliary stream are the SAME product (logits
bitwise equal), the deferred positional-projection gradients and the one-launch LayerNorm fold give the per-block launches' gradients
(f32 sums in another order; the LayerNorm folds add the same partial sums in the same order).  The whole step is also held against the oracle in both modes by
tests/test_model_gpu.py / test_parity_baseline_gpu.py (the default is hoisting ON)."""
import os
import sys

import numpy as np
import pytest
import torch

from tensorflowasr_amd.conformer import SingleProcess

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_model_gpu import _setup  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lens,ulens", [([4000, 2500, 3100], [6, 3, 5])])
def test_hoisted_block_launches_same_step(dev, lens, ulens):
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, lens, ulens)
    assert model._fused_attention() and model.native_blocks and isinstance(model.dp, SingleProcess)
    out = {}
    for hoist in (False, True):
        model.block_hoist = hoist
        logits, _, _ = model._forward(data.inputs, True, None, (None, None))
        assert ("pext" in model._hoisted) == hoist
        model.zero_grad()
        costs = model.loss_and_backward(data, True, (None, None))
        torch.cuda.synchronize()
        out[hoist] = (logits.float().cpu().numpy(), costs.float().cpu().numpy(), {k: v.float().cpu().numpy() for k, v in model.ps.export_keras(model.ps.grad).items()})
    np.testing.assert_array_equal(out[True][0], out[False][0])  # same projection kernel, same arguments: bitwise
    np.testing.assert_array_equal(out[True][1], out[False][1])
    g0, g1 = out[False][2], out[True][2]
    gmax = max(float(np.abs(v).max()) for v in g0.values())
    for k in g0:
        a, b = g0[k], g1[k]
        scale = float(np.abs(a).max()) + 1e-5 * gmax  # (a bias in front of a BatchNorm has a zero gradient: noise of 1e-8)
        # LayerNorm gamma / beta: the same partial sums folded in the same order (upstream BatchNorm atomics reorder run to run: not bitwise);
        # positional projection bias: f32 column sums in another order; everything else: split-K atomics
        tol = 1e-4 if k.endswith(("/ln/g", "/ln/b", "/pos/b")) else 2e-3
        if k.endswith(("sub/conv0/b", "sub/conv1/b", "conv/dw/b")):
            # a bias in front of a BatchNorm: its exact gradient is ZERO, what is stored is the cancellation residue of bf16 summands three
            # orders of magnitude larger.  One summand rounding the other way (the BatchNorm backward's f32 atomics reorder run to run)
            # moves an element by a whole bf16 ulp of the summands (2^-12 seen once in 14 runs of this test; tools/r05/t34.sh) - bounded here
            # against the residue's own size, not against 2e-3 of it
            tol = 0.5
        np.testing.assert_allclose(b, a, rtol=0, atol=tol * scale, err_msg=k)
    assert any("/pos/" in k for k in g0) and any(k.endswith("/ln/g") for k in g0)


def test_hoists_survive_a_process_group_and_buckets_are_final_when_released(dev):
    """A data-parallel rank runs the SAME hoisted step as one GPU: the variables whose gradients are finished after the
    last block (LayerNorm gamma / beta, positional projection, depthwise kernel) live in one region of the flat buffer behind the last block
    (ParamStore.defer_lo / defer_hi), so (i) the hoists stay on with a process group, (ii) every released bucket is FINAL at the moment it is
    released (a copy queued on the compute stream at release time equals the end-of-step gradient: an all-reduce started there sums finished
    values), (iii) the buckets are disjoint, the blocks' buckets and the deferred region tile the encoder blocks' part of the buffer, and
    (iv) the gradients are those of the single-process step."""
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, [4000, 3300], [6, 4])
    ps = model.ps
    assert ps.defer_lo is not None and ps.defer_lo < ps.defer_hi <= ps.offsets["pred/emb"]

    class OneRank:  # duck-typed hook object (tensorflowasr_amd.dp.DataParallel is one): not SingleProcess = a process group
        world, rank = 1, 0

        def __init__(self):
            self.released = []

        def allreduce_stats_(self, t):
            return t

        def set_reduce(self, on):
            pass

        def grads_ready(self, lo, hi):
            self.released.append((lo, hi, ps.grad[lo:hi].clone()))  # stream-ordered behind everything queued so far

        def finish_grads(self):
            pass

    calls = []
    orig = model._deferred_block_grads
    model._deferred_block_grads = lambda hb, T: (calls.append(1), orig(hb, T))
    model.zero_grad()
    model.loss_and_backward(data, True, (None, None))
    torch.cuda.synchronize()
    assert calls == [1] and "pext" in model._hoisted  # one GPU: deferred
    ref = ps.grad.clone()
    hook = OneRank()
    model.dp = hook
    model.zero_grad()
    model.loss_and_backward(data, True, (None, None))
    torch.cuda.synchronize()
    assert calls == [1, 1] and "pext" in model._hoisted  # a process group: the same hoisted step
    spans = sorted((lo, hi) for lo, hi, _ in hook.released)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "buckets overlap"
    blk_lo = ps.offsets["enc/block0/ff1/d1/w"]
    enc = [sp for sp in spans if sp[0] >= blk_lo and sp[1] <= ps.defer_hi]
    assert enc[0][0] == blk_lo and enc[-1] == (ps.defer_lo, ps.defer_hi) and all(a[1] == b[0] for a, b in zip(enc, enc[1:])), enc
    assert len(enc) == cfg.num_blocks + 1
    for lo, hi, snap in hook.released:
        assert torch.equal(snap, ps.grad[lo:hi]), f"bucket [{lo}, {hi}) changed after it was released"
    gmax = float(ref.abs().max())
    np.testing.assert_allclose(ps.grad.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-3 * gmax)  # (split-K / BatchNorm atomics reorder run to run)


def _grads_after_two_steps(model, data):
    """gradients of the SECOND of two consecutive backward passes (the side stream's two scratch arenas and event slots are reused)"""
    out = None
    for _ in range(2):
        model.zero_grad()
        costs = model.loss_and_backward(data, True, (None, None))
        torch.cuda.synchronize()
        out = (costs.float().cpu().numpy(), model.ps.grad.clone())
    return out


def test_side_stream_weight_gradients_and_deferred_launches_same_gradients(dev, monkeypatch):
    """the grouped weight gradients on the low-priority side stream (wgrad_slot arenas reused across two steps) against the
    in-line launches, and the deferred small gradients beside the subsampling's backward (TFASR_DEFER_SIDE=1, default) against the chain
    (TFASR_DEFER_SIDE=0): same costs bitwise, same gradients up to the order of the f32 atomics."""
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, [4000, 2500, 3100], [6, 3, 5])
    res = {}
    for name, ws, defer in (("side", True, "1"), ("inline", False, "1"), ("side_chain", True, "0")):
        model.wgrad_stream = ws
        monkeypatch.setenv("TFASR_DEFER_SIDE", defer)
        res[name] = _grads_after_two_steps(model, data)
    gmax = float(res["inline"][1].abs().max())
    for name in ("side", "side_chain"):
        np.testing.assert_array_equal(res[name][0], res["inline"][0])
        np.testing.assert_allclose(res[name][1].cpu().numpy(), res["inline"][1].cpu().numpy(), rtol=0, atol=2e-3 * gmax, err_msg=name)


def test_batchnorm_statistics_inside_the_depthwise_conv_same_step(dev):
    """ConvModule BatchNorm statistics accumulated by the depthwise conv kernel over 8 copies (`_bn_stats_copies`, default) against the
    separate tfasr_bn_stats launch (1): same sums in another order - costs and gradients agree to the f32-atomics tolerance, and so do the
    moving statistics the forward updates (conformer.py:305-333)."""
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, [4000, 2500, 3100], [6, 3, 5])
    res, mov = {}, {}
    for cp in (1, 8):
        model._bn_stats_copies = cp
        snap = {k: v.clone() for k, v in model.ps.state.items()} if hasattr(model.ps, "state") else None
        res[cp] = _grads_after_two_steps(model, data)
        mov[cp] = snap
    gmax = float(res[1][1].abs().max())
    np.testing.assert_allclose(res[8][0], res[1][0], rtol=2e-3)
    np.testing.assert_allclose(res[8][1].cpu().numpy(), res[1][1].cpu().numpy(), rtol=0, atol=4e-3 * gmax)


def test_front_end_on_the_prediction_stream_same_step(dev):
    """`prefetched_inputs` (what bench.py sets: log-mel + SpecAugment on the prediction network's stream ahead of the previous
    step's tail) against the in-line front end, over two consecutive steps with an optimizer update in between (the second step's front end
    really runs beside the first one's tail): same features path -> same costs bitwise, same parameters after both updates."""
    outs = {}
    for pre in (False, True):
        cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, [4000, 2500, 3100], [6, 3, 5])
        model.prefetched_inputs = pre
        costs = []
        for _ in range(2):
            out = model.train_step(data)
            torch.cuda.synchronize()
            costs.append(out["loss"].float().cpu().numpy().tolist())
        outs[pre] = (costs, model.ps.flat.clone())
    assert outs[True][0] == outs[False][0], outs
    pmax = float(outs[False][1].abs().max())
    np.testing.assert_allclose(outs[True][1].cpu().numpy(), outs[False][1].cpu().numpy(), rtol=0, atol=1e-5 * pmax)
