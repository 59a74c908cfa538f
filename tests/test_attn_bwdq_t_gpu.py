"""The query-side attention backward in transposed orientation (relattn_fused_bwd_qT_kernel, csrc/attn_fused.hip) against the
row-oriented kernel it replaced on the default route (TFASR_ATTN_BWDQ_T=0 selects the old one; the switch is read per call):
same per-pair arithmetic, so every gradient of a train step must agree up to the bf16 rounding of differently ordered f32 sums.
Several key blocks (T' = 250 > 64), ragged lengths with one fully padded query block, full context and the streaming mask.
(The step itself is pinned to the oracle by tests/test_model_gpu.py and tests/test_parity_baseline_gpu.py with the new kernel.)"""
import os

import numpy as np
import pytest
import torch

from test_model_gpu import _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["1", "2", "3"])  # pipelined (default) / lean (three workgroups per CU) / first single-buffered version
@pytest.mark.parametrize("stream", [None, (8, 24), (16, -1)])
def test_transposed_query_backward_matches_row_oriented_kernel(dev, stream, mode):
    N = 160000  # 1000 frames -> T' = 250: four key blocks, the last one ragged
    lens, ulens = [160000, 70000, 121000], [6, 3, 5]  # sample 1: T' = 110, its last query block is entirely padding
    over = {} if stream is None else dict(chunk_size=stream[0], history_size=stream[1], convm_dw_norm="layer")
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, lens, ulens, N=N, **over)
    assert model._fused_attention()
    model.native_blocks = True
    out = {}
    old = os.environ.get("TFASR_ATTN_BWDQ_T")
    try:
        for tag, flag in (("old", "0"), ("old2", "0"), ("new", mode)):
            os.environ["TFASR_ATTN_BWDQ_T"] = flag
            model.zero_grad()
            costs = model.loss_and_backward(data, True, (None, None)).float().cpu().numpy()
            torch.cuda.synchronize()
            out[tag] = (costs, model.ps.export_keras(model.ps.grad))
    finally:
        if old is None:
            os.environ.pop("TFASR_ATTN_BWDQ_T", None)
        else:
            os.environ["TFASR_ATTN_BWDQ_T"] = old
    # (the bf16 forward is not bitwise repeatable - f32 atomics in the batch-norm statistics - so two runs of the SAME kernel give the
    # noise floor the comparison is read against)
    np.testing.assert_allclose(out["new"][0], out["old"][0], rtol=2e-3)

    def rel(a, b):
        num = sum(float(((a[k].double() - b[k].double()) ** 2).sum()) for k in a)
        den = sum(float((a[k].double() ** 2).sum()) for k in a)
        return (num / den) ** 0.5

    g0, g0b, g1 = out["old"][1], out["old2"][1], out["new"][1]
    noise, diff = rel(g0, g0b), rel(g0, g1)
    worst = max(((float((g0[k].double() - g1[k].double()).norm()) / (float(g0[k].double().norm()) + 1e-12), k) for k in g0))
    print(f"all gradients, relative L2: old vs old again {noise:.2e}, old vs new {diff:.2e}; worst tensor {worst}")
    assert diff < max(3.0 * noise, 2e-3), (noise, diff)
    att = [k for k in g0 if "mhsa" in k]
    assert att and all(float(g1[k].abs().max()) > 0 for k in att), att
    big = max(float(g0[k].double().norm()) for k in att)
    for k in att:  # the tensors the kernel feeds directly: per tensor, against that tensor's own noise
        a, b, c = g0[k].double(), g0b[k].double(), g1[k].double()
        n0, n1 = float((a - b).norm()), float((a - c).norm())
        if float(a.norm()) < 1e-2 * big:
            continue  # (the key bias: its gradient is zero in exact arithmetic - softmax rows sum to one - and pure rounding here)
        assert n1 <= max(4.0 * n0, 1e-2 * float(a.norm())), (k, n0, n1, float(a.norm()))


@pytest.mark.parametrize("stream", [None, (8, 24), (16, -1)])
def test_key_per_lane_backward_matches_row_oriented_kernel(dev, stream):
    """relattn_fused_bwd_kT_kernel (a lane owns one key; TFASR_ATTN_BWDK_T=1, read per call) against relattn_fused_bwd_k_kernel: same
    per-pair arithmetic, dK / dV from registers instead of through the per-wave A images."""
    N = 160000
    lens, ulens = [160000, 70000, 121000], [6, 3, 5]
    over = {} if stream is None else dict(chunk_size=stream[0], history_size=stream[1], convm_dw_norm="layer")
    cfg, ocfg, model, W, data, sig, labels, preds = _setup(dev, torch.bfloat16, lens, ulens, N=N, **over)
    assert model._fused_attention()
    model.native_blocks = True
    out = {}
    old = os.environ.get("TFASR_ATTN_BWDK_T")
    try:
        for tag, flag in (("old", "0"), ("old2", "0"), ("new", "1")):
            os.environ["TFASR_ATTN_BWDK_T"] = flag
            model.zero_grad()
            costs = model.loss_and_backward(data, True, (None, None)).float().cpu().numpy()
            torch.cuda.synchronize()
            out[tag] = (costs, model.ps.export_keras(model.ps.grad))
    finally:
        if old is None:
            os.environ.pop("TFASR_ATTN_BWDK_T", None)
        else:
            os.environ["TFASR_ATTN_BWDK_T"] = old

    def rel(a, b):
        num = sum(float(((a[k].double() - b[k].double()) ** 2).sum()) for k in a)
        den = sum(float((a[k].double() ** 2).sum()) for k in a)
        return (num / den) ** 0.5

    g0, g0b, g1 = out["old"][1], out["old2"][1], out["new"][1]
    noise, diff = rel(g0, g0b), rel(g0, g1)
    print(f"all gradients, relative L2: old vs old again {noise:.2e}, old vs new {diff:.2e}")
    assert diff < max(3.0 * noise, 2e-3), (noise, diff)
    att = [k for k in g0 if "mhsa" in k]
    big = max(float(g0[k].double().norm()) for k in att)
    for k in att:
        a, b, c = g0[k].double(), g0b[k].double(), g1[k].double()
        n0, n1 = float((a - b).norm()), float((a - c).norm())
        if float(a.norm()) < 1e-2 * big:
            continue
        assert n1 <= max(4.0 * n0, 1e-2 * float(a.norm())), (k, n0, n1, float(a.norm()))
