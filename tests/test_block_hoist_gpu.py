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
        np.testing.assert_allclose(b, a, rtol=0, atol=tol * scale, err_msg=k)
    assert any("/pos/" in k for k in g0) and any(k.endswith("/ln/g") for k in g0)


def test_hoist_stays_per_block_with_a_process_group(dev):
    """A data-parallel group releases a block's gradient bucket right behind the block, and the hoists' auxiliary stream together with the
    weight-gradient stream was measured 1.7x slower under a process group: by default nothing is hoisted there."""
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, [4000, 3300], [6, 4])

    class OneRank:  # duck-typed hook object (tensorflowasr_amd.dp.DataParallel is one): not SingleProcess = a process group
        world, rank = 1, 0

        def allreduce_stats_(self, t):
            return t

        def set_reduce(self, on):
            pass

        def grads_ready(self, lo, hi):
            pass

        def finish_grads(self):
            pass

    calls = []
    orig = model._deferred_block_grads
    model._deferred_block_grads = lambda hb, T: (calls.append(1), orig(hb, T))
    model.zero_grad()
    model.loss_and_backward(data, True, (None, None))
    torch.cuda.synchronize()
    assert calls == [1] and "pext" in model._hoisted  # one GPU: deferred
    ref = model.ps.grad.clone()
    model.dp = OneRank()
    model.zero_grad()
    model.loss_and_backward(data, True, (None, None))
    torch.cuda.synchronize()
    assert calls == [1] and "pext" not in model._hoisted  # a process group: the measured round-3 step, nothing hoisted, weight gradients in line
    g = model.ps.grad
    assert bool(torch.isfinite(g).all())
    assert float((g - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_weight_gradient_stream_same_step(dev):
    """The blocks' grouped weight gradients beside the next block's backward on the executor's second stream (tfasr_block_io.wgrad_slot,
    the default) against the in-line launches: same gradients (split-K atomics reorder run to run), ragged batch, two steps in a row
    (the second step reuses the arenas the first step's groups read)."""
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, [4000, 2500, 3100], [6, 3, 5])
    out = {}
    for on in (False, True, True):
        model.wgrad_stream = on
        model.zero_grad()
        costs = model.loss_and_backward(data, True, (None, None))
        torch.cuda.synchronize()
        out[on] = (costs.float().cpu().numpy(), model.ps.grad.clone())
    np.testing.assert_array_equal(out[True][0], out[False][0])
    g0, g1 = out[False][1], out[True][1]
    assert bool(torch.isfinite(g1).all())
    assert float((g1 - g0).abs().max()) <= 2e-3 * float(g0.abs().max())
