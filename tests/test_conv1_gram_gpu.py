"""conv1 + BatchNorm0 through the Gram matrix of the 3x3 patches (csrc/conv2d.hip: tfasr_conv1_gram / _stats_from_gram / _bn_bwd_onepass /
_bn_bwd_finalize) against the two-pass kernels they replace and against a plain torch restatement (Conv2dSubsampling's first block,
subsampling.py:163-230: causal 3x3 stride-2 conv, BatchNormalization, swish)."""
import numpy as np
import pytest
import torch

from tensorflowasr_amd import kernels as K

pytestmark = pytest.mark.gpu


def _ref(feats, w, b, gamma, beta, dy_s2d_to_dense):
    """torch f64 restatement: z = conv(feats), BN batch statistics, swish; returns stats, and the gradients for dy (dense [B,T1,F1,C])"""
    B, T0, F0 = feats.shape
    x = feats.double()[:, None]  # [B,1,T0,F0]
    xp = torch.nn.functional.pad(x, (2, 0, 2, 0))
    wt = w.double().permute(3, 2, 0, 1).contiguous().requires_grad_(True)  # [C,1,3,3]
    bb = b.double().clone().requires_grad_(True)
    z = torch.nn.functional.conv2d(xp, wt, bb, stride=2)  # [B,C,T1,F1]
    return z, wt, bb


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T0,F0,C", [(3, 37, 80, 64), (2, 50, 17, 256), (4, 9, 80, 192)])
def test_gram_route_matches_two_pass_and_reference(dev, dtype, B, T0, F0, C):
    g = torch.Generator().manual_seed(B * 100 + T0)
    feats = (torch.randn(B, T0, F0, generator=g) * 2.5 - 6.0).to(dev).to(dtype)  # log-mel like: mean^2 >> variance
    w = (torch.randn(3, 3, 1, C, generator=g) * 0.3).to(dev)
    b = (torch.randn(C, generator=g) * 0.1).to(dev)
    gamma, beta = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    T1, F1 = (T0 + 1) // 2, (F0 + 1) // 2
    T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
    N = B * T1 * F1
    # forward statistics: Gram route vs the channel pass vs torch
    gram = K.conv1_gram(feats, torch.empty(K.CONV1_GRAM_DOUBLES, dtype=torch.float64, device=dev))
    st_g = torch.zeros(2 * C + 1, device=dev)
    K.conv1_stats_from_gram(gram, w, b, st_g)
    st_p = torch.zeros(2 * C + 1, device=dev)
    K.conv1_stats(feats, w, b, st_p)
    z, wt, bb = _ref(feats.float().cpu(), w.cpu(), b.cpu(), gamma.cpu(), beta.cpu(), None)
    ref_s1, ref_s2 = z.sum((0, 2, 3)), (z * z).sum((0, 2, 3))
    assert float(gram.view(8, 96)[:, 90].sum()) == N
    np.testing.assert_allclose(st_g[:C].cpu().numpy(), ref_s1.detach().numpy(), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(st_g[C:2 * C].cpu().numpy(), ref_s2.detach().numpy(), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(st_g[:2 * C].cpu().numpy(), st_p[:2 * C].cpu().numpy(), rtol=1e-4, atol=1e-2)
    # BatchNorm coefficients from the statistics
    fin = torch.empty(4 * C, device=dev)
    K.bn_finalize(st_g, N, gamma, beta, fin, torch.zeros(C, device=dev), torch.ones(C, device=dev), 0.99, 1e-3, True)
    # a gradient in the S layout: write through the apply kernel's addressing by running it on a dense random tensor is not possible, so
    # build dy directly in the haloed space-to-depth layout with random values at every slot (the kernels only read the valid ones)
    rows = B * (T2 + 1) * (F2 + 1)
    dy = (torch.randn(rows, 4 * C, generator=g) * 0.5).to(dev).to(dtype)
    # two-pass route
    bst2 = torch.zeros(2 * C, device=dev)
    K.conv1_bn_bwd_stats_s2d(feats, w, b, fin, dy, bst2)
    dw2, db2 = torch.zeros(3, 3, 1, C, device=dev), torch.zeros(C, device=dev)
    K.conv1_bn_bwd_apply_s2d(feats, w, b, fin, bst2, N, dy, dw2, db2)
    # one-pass route
    bst1, pbuf = torch.zeros(2 * C, device=dev), torch.zeros(10 * C, device=dev)
    K.conv1_bn_bwd_onepass_s2d(feats, w, b, fin, dy, bst1, pbuf)
    dw1, db1 = torch.zeros(3, 3, 1, C, device=dev), torch.zeros(C, device=dev)
    K.conv1_bn_bwd_finalize(gram, w, b, fin, bst1, N, pbuf, dw1, db1)
    torch.cuda.synchronize()
    np.testing.assert_allclose(bst1.cpu().numpy(), bst2.cpu().numpy(), rtol=1e-4, atol=1e-3)
    scale = float(dw2.abs().max())
    np.testing.assert_allclose(dw1.cpu().numpy(), dw2.cpu().numpy(), rtol=2e-3, atol=2e-4 * scale + 1e-4)
    # the bias gradient through a BatchNorm is zero up to rounding on one rank: both routes must agree on "small"
    assert float(db1.abs().max()) <= 1e-3 * (1.0 + float(bst1[:C].abs().max())) and float(db2.abs().max()) <= 1e-3 * (1.0 + float(bst2[:C].abs().max()))
