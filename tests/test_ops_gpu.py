"""GPU numerics of the pointwise / reduction / frontend kernels vs torch-CPU fp32 (oracle/conformer_ref.py pieces)."""
import math

import numpy as np
import pytest
import torch

from oracle import conformer_ref as R
from tensorflowasr_amd import kernels as K
from tensorflowasr_amd.kernels import ACT_NONE, ACT_SWISH

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16]


def tol(dtype, f32=(1e-5, 1e-5), bf16=(2e-2, 2e-2)):
    return dict(rtol=f32[0], atol=f32[1]) if dtype == torch.float32 else dict(rtol=bf16[0], atol=bf16[1])


def rt(x, dtype):
    """round-trip through the storage dtype so the reference sees the same inputs"""
    return x.to(dtype).float()


def cmp(a, b, **kw):
    np.testing.assert_allclose(a.float().cpu().numpy(), b.float().cpu().numpy(), **kw)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C", [144, 256, 40])
def test_layernorm(dev, dtype, C):
    g = torch.Generator().manual_seed(C)
    rows = 77
    x = rt(torch.randn(rows, C, generator=g) * 2 + 0.5, dtype)
    gam, bet = torch.randn(C, generator=g), torch.randn(C, generator=g)
    dy = rt(torch.randn(rows, C, generator=g), dtype)
    add = rt(torch.randn(rows, C, generator=g), dtype)
    xr = x.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    yr = R.layer_norm(xr, gr, br)
    yr.backward(dy)
    y, mean, rstd = K.layernorm_fwd(x.to(dev).to(dtype), gam.to(dev), bet.to(dev))
    cmp(y, yr.detach(), **tol(dtype, (1e-5, 2e-5)))
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx = K.layernorm_bwd(dy.to(dev).to(dtype), x.to(dev).to(dtype), gam.to(dev), mean, rstd, dgam, dbet, add=add.to(dev).to(dtype))
    cmp(dx, xr.grad + add, **tol(dtype, (1e-4, 2e-5), (3e-2, 3e-2)))
    cmp(dgam, gr.grad, rtol=1e-4, atol=1e-3)
    cmp(dbet, br.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C", [256, 144, 40])
def test_layernorm_bwd_with_dropped_copy(dev, dtype, C):
    """tfasr_layernorm_bwd_drop == tfasr_layernorm_bwd followed by tfasr_dropout(dx) (bit-exact: same mask, same rounding), for the
    vectorised kernel (bf16, C % 8 == 0) and the fallback shapes."""
    g = torch.Generator().manual_seed(C)
    rows = 301
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(dev).to(dtype)
    dy = torch.randn(rows, C, generator=g).to(dev).to(dtype)
    add = torch.randn(rows, C, generator=g).to(dev).to(dtype)
    gam, bet = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    _, mean, rstd = K.layernorm_fwd(x, gam, bet)
    dg1, db1, dg2, db2 = (torch.zeros(C, device=dev) for _ in range(4))
    dx1 = K.layernorm_bwd(dy, x, gam, mean, rstd, dg1, db1, add=add)
    want = K.dropout(dx1, 0.1, 12345)
    dropped = torch.empty_like(x)
    dx2 = K.layernorm_bwd(dy, x, gam, mean, rstd, dg2, db2, add=add, dropped=dropped, drop_p=0.1, drop_seed=12345)
    assert torch.equal(dx1, dx2) and torch.equal(dropped, want)
    cmp(dg1, dg2, rtol=1e-5, atol=1e-4)
    frac = float((dropped == 0).float().mean())
    assert 0.05 < frac < 0.16


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C", [144, 1280])  # 1280 = ContextNet-L width: channel slabs in the statistics kernels
def test_batchnorm_swish(dev, dtype, C):
    if C > 1024 and dtype == torch.float32:
        pytest.skip("the f32 (parity-mode) statistics kernels cover C <= 1024")
    g = torch.Generator().manual_seed(0)
    rows = 500
    x = rt(torch.randn(rows, C, generator=g) * 1.5 + 0.3, dtype)
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    dy = rt(torch.randn(rows, C, generator=g), dtype)
    xr, gr, br = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    yr, mean, var = R.batch_norm_train(xr, gr, br)
    yr = R.swish(yr)
    yr.backward(dy)
    xd = x.to(dev).to(dtype)
    stats = torch.zeros(2 * C, device=dev)
    K.bn_stats(xd, stats)
    fin = torch.empty(4 * C, device=dev)
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    K.bn_finalize(stats, rows, gam.to(dev), bet.to(dev), fin, mm, mv)
    cmp(fin[:C], mean.detach(), rtol=1e-4, atol=1e-5)
    cmp(mm, mean.detach() * 0.01, rtol=1e-3, atol=1e-6)
    cmp(mv, 0.99 + var.detach() * 0.01, rtol=1e-4, atol=1e-6)
    y = K.bn_apply_fwd(xd, fin, ACT_SWISH)
    cmp(y, yr.detach(), **tol(dtype, (1e-4, 1e-5)))
    bstats = torch.zeros(2 * C, device=dev)
    dyd = dy.to(dev).to(dtype)
    K.bn_bwd_stats(xd, dyd, fin, bstats, ACT_SWISH)
    dx = K.bn_apply_bwd(xd, dyd, fin, bstats, rows, ACT_SWISH)
    cmp(dx, xr.grad, **tol(dtype, (1e-3, 2e-5), (3e-2, 3e-2)))
    cmp(bstats[C:], gr.grad, rtol=1e-3, atol=1e-3)
    cmp(bstats[:C], br.grad, rtol=1e-3, atol=1e-3)
    # inference mode uses the moving statistics
    K.bn_finalize(None, 1, gam.to(dev), bet.to(dev), fin, mm, mv, training=False)
    yi = K.bn_apply_fwd(xd, fin, ACT_NONE)
    cmp(yi, R.batch_norm_infer(x, gam, bet, mm.cpu(), mv.cpu()), **tol(dtype, (1e-4, 1e-5)))


@pytest.mark.parametrize("dtype", DT)
def test_glu_dwconv(dev, dtype):
    g = torch.Generator().manual_seed(1)
    B, T, C, Kk = 3, 53, 144, 31
    x = rt(torch.randn(B, T, 2 * C, generator=g), dtype)
    w, b = torch.randn(Kk, C, generator=g) * 0.2, torch.randn(C, generator=g) * 0.1
    dy = rt(torch.randn(B, T, C, generator=g), dtype)
    xr, wr, brr = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    a, bb = xr.chunk(2, -1)
    gl = a * torch.sigmoid(bb)
    gl.retain_grad()
    yr = R.depthwise_conv1d_causal(gl, wr, brr)
    yr.backward(dy)
    xd = x.to(dev).to(dtype)
    gd = K.glu_fwd(xd)
    cmp(gd, gl.detach(), **tol(dtype))
    gin = gl.detach().to(dev).to(dtype)
    y = K.dwconv_fwd(gin, w.to(dev), b.to(dev))
    ref_y = R.depthwise_conv1d_causal(rt(gl.detach(), dtype), w, b)
    cmp(y, ref_y, **tol(dtype, (1e-4, 1e-5)))
    dyd = dy.to(dev).to(dtype)
    dg = K.dwconv_bwd_data(dyd, w.to(dev))
    cmp(dg, gl.grad, **tol(dtype, (1e-4, 1e-5)))
    dw, db = torch.zeros(Kk, C, device=dev), torch.zeros(C, device=dev)
    K.dwconv_bwd_weight(gin, dyd, dw, db)
    # reference weight grad with the same (rounded) input
    g2 = rt(gl.detach(), dtype).requires_grad_(False)
    w2 = w.clone().requires_grad_(True)
    R.depthwise_conv1d_causal(g2, w2, b).backward(dy)
    cmp(dw, w2.grad, rtol=1e-3, atol=2e-3)
    cmp(db, dy.sum((0, 1)), rtol=1e-3, atol=1e-3)
    dx = K.glu_bwd(xd, gl.grad.to(dev).to(dtype))
    cmp(dx, xr.grad, **tol(dtype, (1e-4, 1e-5), (3e-2, 3e-2)))


@pytest.mark.parametrize("rows,C", [(23808, 256), (777, 144), (4096, 512)])
def test_layernorm_bwd_partial_sums_and_fold(dev, rows, C):
    """tfasr_layernorm_bwd_part + tfasr_layernorm_bwd_fold (per-block partial sums of the gamma / beta gradients, one writer per column)
    against tfasr_layernorm_bwd (atomics): same dx bitwise, same sums up to summation order; the fold ADDS to its destinations."""
    g = torch.Generator().manual_seed(rows + C)
    dt = torch.bfloat16
    x = torch.randn(rows, C, generator=g).to(dev).to(dt)
    dy = torch.randn(rows, C, generator=g).to(dev).to(dt)
    add = torch.randn(rows, C, generator=g).to(dev).to(dt)
    gamma, beta = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    _, mean, rstd = K.layernorm_fwd(x, gamma, beta)
    dg0, db0 = torch.full((C,), 0.5, device=dev), torch.full((C,), -0.25, device=dev)
    dg1, db1 = dg0.clone(), db0.clone()
    dx0 = K.layernorm_bwd(dy, x, gamma, mean, rstd, dg0, db0, add=add)
    dx1 = K.layernorm_bwd_fold(dy, x, gamma, mean, rstd, dg1, db1, add=add)
    assert dx1 is not None
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1)
    cmp(dg1, dg0.cpu(), rtol=2e-4, atol=2e-2)
    cmp(db1, db0.cpu(), rtol=2e-4, atol=2e-2)


@pytest.mark.parametrize("rows,Kd,p", [(23776, 1024, 0.1), (14784, 768, 0.1), (777, 512, 0.0), (40, 64, 0.25), (24500, 256, 0.1), (30000, 256, 0.1)])
def test_dense_ln_bwd_one_launch(dev, rows, Kd, p):
    """tfasr_dense_ln_bwd: the backward of `Dense(LayerNorm(x))` (FFModule encoders/conformer.py:66-109, the q/k/v projection
    multihead_attention.py:628-637, ConvModule's first pointwise conv convolution.py:159-228) in one launch, against (a) the oracle's
    autograd of the same expression (torch-CPU f32: dx incl. the residual `add`, gamma / beta gradients) and (b) the two-launch route
    (tfasr_gemm + tfasr_layernorm_bwd_drop: the product rounded to bf16 in between, so only close).  Row counts: the two bench batches
    (96- and 64-row tiles, one round of workgroups), a ragged last tile, fewer rows than one tile, 24500 rows (256 tiles of 96), and a
    row count outside the range (UNSUPPORTED: more 96-row tiles than partial-sum slots)."""
    g = torch.Generator().manual_seed(rows + Kd)
    dt, d = torch.bfloat16, 256
    x = rt(torch.randn(rows, d, generator=g) * 1.5 + 0.3, dt)
    W = rt(torch.randn(d, Kd, generator=g) / math.sqrt(d), dt)
    dy = rt(torch.randn(rows, Kd, generator=g) * 0.5, dt)
    add = rt(torch.randn(rows, d, generator=g), dt)
    gam, bet = torch.randn(d, generator=g) * 0.5 + 1.0, torch.randn(d, generator=g) * 0.1
    xr, gr, br = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    (R.layer_norm(xr, gr, br) @ W).backward(dy)
    xd, Wd, dyd, addd, gd, bd = x.to(dev).to(dt), W.to(dev).to(dt), dy.to(dev).to(dt), add.to(dev).to(dt), gam.to(dev), bet.to(dev)
    _, mean, rstd = K.layernorm_fwd(xd, gd, bd)
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    dropped = torch.empty_like(xd) if p > 0 else None
    dx = K.dense_ln_bwd(dyd, Wd, xd, gd, mean, rstd, dg, db, add=addd, dropped=dropped, drop_p=p, drop_seed=4711)
    if rows > 96 * 256:  # more 96-row tiles than partial-sum slots (= CUs): outside the kernel's range, the caller keeps the two launches
        assert dx is None
        return
    assert dx is not None
    torch.cuda.synchronize()
    cmp(dx, xr.grad + add, rtol=3e-2, atol=3e-2)
    rel = float((dx.float().cpu() - (xr.grad + add)).norm() / (xr.grad + add).norm())
    assert rel < 4e-3, rel  # (bf16 output rounding only: the product stays f32 inside the launch)
    scale_g = float(gr.grad.abs().max())
    cmp(dg, gr.grad, rtol=2e-3, atol=2e-3 * scale_g)
    cmp(db, br.grad, rtol=2e-3, atol=2e-3 * float(br.grad.abs().max()))
    if p > 0:
        assert torch.equal(dropped, K.dropout(dx, p, 4711))  # the same mask and rounding as tfasr_dropout(dx)
    # the two-launch route
    dln = torch.empty(rows, d, dtype=dt, device=dev)
    K.gemm(dyd, Wd, dln, rows, d, Kd, Kd, Kd, d, trans_b=True)
    dg2, db2 = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    dx2 = K.layernorm_bwd(dln, xd, gd, mean, rstd, dg2, db2, add=addd)
    rel2 = float((dx.float() - dx2.float()).norm() / dx2.float().norm())
    assert rel2 < 6e-3, rel2
    cmp(dg, dg2.cpu(), rtol=5e-3, atol=5e-3 * scale_g)


@pytest.mark.parametrize("Kk,C,T", [(5, 640, 150), (5, 80, 77), (3, 256, 64), (7, 144, 130), (8, 264, 65), (15, 144, 53), (32, 144, 53)])
def test_dwconv_kernel_sizes(dev, Kk, C, T):
    """The bf16 depthwise kernels are instantiated per window bound (8 / 32 taps) and per weight-gradient kernel size: every size the
    reference configs use (ContextNet 5, Conformer 31/32, small test sizes) against the torch restatement."""
    g = torch.Generator().manual_seed(100 + Kk)
    B = 3
    dtype = torch.bfloat16
    x = rt(torch.randn(B, T, C, generator=g), dtype)
    w, b = torch.randn(Kk, C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1
    dy = rt(torch.randn(B, T, C, generator=g), dtype)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = R.depthwise_conv1d_causal(xr, wr, b)
    yr.backward(dy)
    xd, dyd = x.to(dev).to(dtype), dy.to(dev).to(dtype)
    cmp(K.dwconv_fwd(xd, w.to(dev), b.to(dev)), yr.detach(), **tol(dtype))
    cmp(K.dwconv_bwd_data(dyd, w.to(dev)), xr.grad, **tol(dtype))
    dw, db = torch.zeros(Kk, C, device=dev), torch.zeros(C, device=dev)
    K.dwconv_bwd_weight(xd, dyd, dw, db)
    cmp(dw, wr.grad, rtol=1e-3, atol=2e-3)
    cmp(db, dy.sum((0, 1)), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("Kk,C,T,copies", [(32, 256, 157, 8), (31, 144, 53, 4), (5, 640, 150, 8), (32, 256, 33, 1)])
def test_dwconv_forward_with_batchnorm_statistics(dev, Kk, C, T, copies):
    """tfasr_dwconv_fwd_stats: the conv output is bitwise the plain forward's, the copies of the statistics add up to what tfasr_bn_stats
    takes of that output (f32 sums in another order), and the finalize + apply launch reading the copies gives the coefficients / output of
    the single-buffer route (ConvModule BatchNorm, conformer.py:305-333)."""
    g = torch.Generator().manual_seed(300 + Kk + T)
    B = 5
    x = (torch.randn(B, T, C, generator=g) * 1.5 + 0.3).to(dev).to(torch.bfloat16)
    w, b = (torch.randn(Kk, C, generator=g) * 0.3).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    y0 = K.dwconv_fwd(x, w, b)
    st0 = torch.zeros(2 * C, device=dev)
    K.bn_stats(y0.view(B * T, C), st0)
    st = torch.zeros(copies, 2 * C, device=dev)
    y1 = K.dwconv_fwd_stats(x, w, b, st)
    assert y1 is not None
    assert torch.equal(y0, y1)
    assert copies == 1 or int((st.abs().sum(1) > 0).sum()) > 1  # the workgroups really spread over the copies
    ref = y0.view(B * T, C).float()
    cmp(st.sum(0)[:C], ref.sum(0), rtol=1e-4, atol=1e-2)
    cmp(st.sum(0)[C:], (ref * ref).sum(0), rtol=1e-4, atol=1e-2)
    cmp(st.sum(0), st0, rtol=1e-4, atol=1e-2)
    gm, bt = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    outs = []
    for stats, cp in ((st0, 1), (st, copies)):
        fin, mm, mv = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
        o = K.bn_finalize_apply_fwd(y0.view(B * T, C), stats, float(B * T), gm, bt, fin, mm, mv, act=K.ACT_SWISH, copies=cp)
        if o is None:
            return  # channel count outside the row kernel (block.hip then keeps everything in copy 0)
        outs.append((o.float(), fin, mm, mv))
    cmp(outs[1][1], outs[0][1], rtol=1e-4, atol=1e-4)
    cmp(outs[1][2], outs[0][2], rtol=1e-4, atol=1e-5)
    cmp(outs[1][3], outs[0][3], rtol=1e-4, atol=1e-5)
    cmp(outs[1][0], outs[0][0], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("Kk,C,T,copies", [(32, 256, 157, 8), (31, 144, 53, 4), (15, 512, 33, 2)])
def test_glu_depthwise_conv_and_statistics_in_one_launch(dev, Kk, C, T, copies):
    """tfasr_glu_dwconv_fwd_stats against the three launches it replaces (tfasr_glu_fwd, tfasr_dwconv_fwd, tfasr_bn_stats): gated input and
    conv output bitwise, statistics up to the order of the f32 additions (ConvModule, encoders/conformer.py:300-333)."""
    g = torch.Generator().manual_seed(500 + Kk + T)
    B = 4
    gx = (torch.randn(B, T, 2 * C, generator=g) * 1.2).to(dev).to(torch.bfloat16)
    w, b = (torch.randn(Kk, C, generator=g) * 0.3).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    g0 = K.glu_fwd(gx.view(B * T, 2 * C)).view(B, T, C)
    y0 = K.dwconv_fwd(g0, w, b)
    st0 = torch.zeros(2 * C, device=dev)
    K.bn_stats(y0.view(B * T, C), st0)
    st = torch.zeros(copies, 2 * C, device=dev)
    out = K.glu_dwconv_fwd_stats(gx, w, b, st)
    assert out is not None
    g1, y1 = out
    assert torch.equal(g0, g1)
    assert torch.equal(y0, y1)
    cmp(st.sum(0), st0, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("Kk,C,T,copies", [(32, 256, 157, 8), (31, 144, 53, 1), (15, 512, 33, 4)])
def test_batchnorm_backward_depthwise_gradient_and_glu_backward_in_one_launch(dev, Kk, C, T, copies):
    """tfasr_bn_dwconv_bwd_data_glu against the two launches it replaces (tfasr_bn_apply_bwd_grads_copies with swish, then
    tfasr_dwconv_bwd_data_glu): dcv bitwise (same arithmetic per element), the GLU gradient bitwise from it, the BatchNorm parameter
    gradients equal (ConvModule backward, encoders/conformer.py:300-333)."""
    g = torch.Generator().manual_seed(700 + Kk + T)
    B = 3
    bf = torch.bfloat16
    x = (torch.randn(B, T, C, generator=g) * 1.3 + 0.2).to(dev).to(bf)
    dsw = (torch.randn(B, T, C, generator=g) * 0.4).to(dev).to(bf)
    gx = (torch.randn(B, T, 2 * C, generator=g)).to(dev).to(bf)
    w = (torch.randn(Kk, C, generator=g) * 0.3).to(dev)
    xf = x.float().view(-1, C)
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-3)
    gm, bt = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    fin = torch.cat([mean, rstd, gm * rstd, bt - mean * gm * rstd]).contiguous()
    st1 = torch.zeros(2 * C, device=dev)
    K.bn_bwd_stats(x.view(-1, C), dsw.view(-1, C), fin, st1, K.ACT_SWISH)
    st = torch.zeros(copies, 2 * C, device=dev)
    st[0] = st1 * 0.25
    st[-1] += st1 * 0.75
    count = float(B * T)
    dg0, db0 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dcv0 = K.bn_apply_bwd(x.view(-1, C), dsw.view(-1, C), fin, st, count, K.ACT_SWISH, dgamma=dg0, dbeta=db0, grad_scale=0.5, copies=copies) if _rows_ok(C) else None
    if dcv0 is None:
        dcv0 = K.bn_apply_bwd(x.view(-1, C), dsw.view(-1, C), fin, st.sum(0), count, K.ACT_SWISH, dgamma=dg0, dbeta=db0, grad_scale=0.5)
    dglu0 = K.dwconv_bwd_data_glu(dcv0.view(B, T, C), w, gx)
    dg1, db1 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    out = K.bn_dwconv_bwd_data_glu(x, dsw, fin, st, count, w, gx, dgamma=dg1, dbeta=db1, grad_scale=0.5)
    assert out is not None and dglu0 is not None
    dcv1, dglu1 = out
    cmp(dcv1.view(-1, C), dcv0, rtol=1e-2, atol=1e-2 * float(dcv0.float().abs().max()))
    cmp(dglu1, dglu0, rtol=2e-2, atol=2e-2 * float(dglu0.float().abs().max()))
    cmp(dg1, dg0, rtol=1e-5, atol=1e-5)
    cmp(db1, db0, rtol=1e-5, atol=1e-5)


def _rows_ok(C):
    return C % 8 == 0 and 64 <= C <= 2048 and 256 % (C // 8) == 0


@pytest.mark.parametrize("rows,C,copies", [(12 * 256 + 37, 256, 8), (9000, 256, 1), (40000, 128, 4)])
def test_gemm_epilogue_batchnorm_backward_sums(dev, rows, C, copies):
    """tfasr_gemm_args.bns_*: the data gradient of the ConvModule's second pointwise conv with the BatchNorm backward sums in its epilogue
    against the two-pass route (tfasr_gemm, then tfasr_bn_bwd_stats over x and the stored gradient): same product bitwise, same sums up
    to the order of the f32 additions, same apply pass from the copies (conformer.py:305-333)."""
    g = torch.Generator().manual_seed(rows + C)
    bf = torch.bfloat16
    dy = (torch.randn(rows, C, generator=g) * 0.5).to(dev).to(bf)
    W = (torch.randn(C, C, generator=g) / 16).to(dev).to(bf)        # [din, dout]: dx = dy @ W^T
    x = (torch.randn(rows, C, generator=g) * 1.3 + 0.2).to(dev).to(bf)
    mean, var = x.float().mean(0), x.float().var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-3)
    gm, bt = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    fin = torch.cat([mean, rstd, gm * rstd, bt - mean * gm * rstd]).contiguous()
    d0 = torch.empty(rows, C, dtype=bf, device=dev)
    K.gemm(dy, W, d0, rows, C, C, C, C, C, trans_b=True, alpha=0.5)
    st0 = torch.zeros(2 * C, device=dev)
    K.bn_bwd_stats(x, d0, fin, st0, K.ACT_SWISH)
    d1 = torch.empty(rows, C, dtype=bf, device=dev)
    st = torch.zeros(copies, 2 * C, device=dev)
    K.gemm(dy, W, d1, rows, C, C, C, C, C, trans_b=True, alpha=0.5, bns=(x, fin, st))
    assert torch.equal(d0, d1)
    scale = float(st0.abs().max())
    cmp(st.sum(0), st0, rtol=1e-3, atol=2e-4 * scale)
    assert copies == 1 or int((st.abs().sum(1) > 0).sum()) > 1
    dx0 = K.bn_apply_bwd(x, d0, fin, st0, float(rows), K.ACT_SWISH)
    dx1 = K.bn_apply_bwd(x, d0, fin, st, float(rows), K.ACT_SWISH, copies=copies)
    cmp(dx1, dx0, rtol=2e-2, atol=2e-3)


def test_gemm_epilogue_batchnorm_sums_with_position_channel_columns(dev):
    """tfasr_gemm_args.bns_c: the product's columns are (position, channel) pairs and its rows carry a wider stride (the subsampling's linear
    layer writing into the haloed layout, subsampling.py:197-230): sums per CHANNEL over rows and positions against tfasr_bn_bwd_stats over
    the [rows * positions, C] view, product unchanged."""
    g = torch.Generator().manual_seed(99)
    bf = torch.bfloat16
    rows, C, npos, Kd = 4000, 256, 5, 256
    N, ldd = npos * C, (npos + 1) * C
    dy = (torch.randn(rows, Kd, generator=g) * 0.5).to(dev).to(bf)
    W = (torch.randn(N, Kd, generator=g) / 16).to(dev).to(bf)              # [N, K] row-major: D = dy @ W^T
    xfull = (torch.randn(rows, ldd, generator=g) * 1.3 + 0.2).to(dev).to(bf)
    xv = xfull.float()[:, C:].reshape(rows * npos, C)
    mean, var = xv.mean(0), xv.var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-3)
    gm, bt = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    fin = torch.cat([mean, rstd, gm * rstd, bt - mean * gm * rstd]).contiguous()
    d0 = torch.zeros(rows, ldd, dtype=bf, device=dev)
    d1 = torch.zeros(rows, ldd, dtype=bf, device=dev)
    K.gemm(dy, W, d0.view(-1)[C:], rows, N, Kd, Kd, Kd, ldd, trans_b=True)
    st = torch.zeros(8, 2 * C, device=dev)
    K.gemm(dy, W, d1.view(-1)[C:], rows, N, Kd, Kd, Kd, ldd, trans_b=True, bns=(xfull.view(-1)[C:], fin, st, C))
    assert torch.equal(d0, d1)
    st0 = torch.zeros(2 * C, device=dev)
    K.bn_bwd_stats(xfull[:, C:].reshape(rows * npos, C).contiguous(), d0[:, C:].reshape(rows * npos, C).contiguous(), fin, st0, K.ACT_SWISH)
    cmp(st.sum(0), st0, rtol=1e-3, atol=2e-4 * float(st0.abs().max()))


@pytest.mark.parametrize("dtype", DT)
def test_bias2_embedding_colsum_cast(dev, dtype):
    g = torch.Generator().manual_seed(2)
    rows, C = 90, 144
    qkv = rt(torch.randn(rows, 3 * C, generator=g), dtype)
    u, v = torch.randn(C, generator=g), torch.randn(C, generator=g)
    qd = qkv.to(dev).to(dtype)
    y1, y2 = K.bias2_fwd(qd, 3 * C, u.to(dev), v.to(dev), rows, C)
    cmp(y1, qkv[:, :C] + u, **tol(dtype))
    cmp(y2, qkv[:, :C] + v, **tol(dtype))
    d1, d2 = rt(torch.randn(rows, C, generator=g), dtype), rt(torch.randn(rows, C, generator=g), dtype)
    dq = torch.zeros(rows, 3 * C, device=dev, dtype=dtype)
    du, dv = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    K.bias2_bwd(d1.to(dev).to(dtype), d2.to(dev).to(dtype), dq, 3 * C, du, dv, rows, C)
    cmp(dq[:, :C], d1 + d2, **tol(dtype))
    assert dq[:, C:].abs().max().item() == 0
    cmp(du, d1.sum(0), rtol=1e-4, atol=1e-3)
    cmp(dv, d2.sum(0), rtol=1e-4, atol=1e-3)
    # embedding
    V, E = 50, 24
    table = torch.randn(V, E, generator=g)
    idx = torch.randint(0, V, (4, 9), generator=g, dtype=torch.int32)
    out = K.embedding_fwd(idx.to(dev), table.to(dev), dtype)
    cmp(out, table[idx.long()], **tol(dtype))
    dout = rt(torch.randn(4, 9, E, generator=g), dtype)
    dt_ = torch.zeros(V, E, device=dev)
    K.embedding_bwd(idx.to(dev), dout.to(dev).to(dtype), dt_)
    ref = torch.zeros(V, E).index_add_(0, idx.view(-1).long(), dout.view(-1, E))
    cmp(dt_, ref, rtol=1e-4, atol=1e-4)
    # colsum over a strided view + cast
    x = rt(torch.randn(300, 1000, generator=g), dtype)
    o = torch.zeros(1000, device=dev)
    K.colsum(x.to(dev).to(dtype), o, scale=0.5)
    cmp(o, 0.5 * x.sum(0), rtol=1e-4, atol=2e-3)
    src = torch.randn(1003, generator=g)
    dst = torch.empty(1003, device=dev, dtype=torch.bfloat16)
    K.cast(src.to(dev), dst)
    cmp(dst, src.to(torch.bfloat16), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", DT)
def test_joint(dev, dtype):
    g = torch.Generator().manual_seed(3)
    B, T, U1, J = 2, 7, 5, 40
    e, p = rt(torch.randn(B, T, J, generator=g), dtype), rt(torch.randn(B, U1, J, generator=g), dtype)
    er, pr = e.clone().requires_grad_(True), p.clone().requires_grad_(True)
    hr = torch.tanh(er[:, :, None] + pr[:, None])
    dh = rt(torch.randn(B, T, U1, J, generator=g), dtype)
    hr.backward(dh)
    h = K.joint_fwd(e.to(dev).to(dtype), p.to(dev).to(dtype))
    cmp(h, hr.detach(), **tol(dtype, (1e-5, 1e-6)))
    denc, dpred = K.joint_bwd(hr.detach().to(dev).to(dtype), dh.to(dev).to(dtype))
    cmp(denc, er.grad, **tol(dtype, (1e-4, 1e-5), (3e-2, 5e-2)))
    cmp(dpred, pr.grad, **tol(dtype, (1e-4, 1e-5), (3e-2, 5e-2)))


def test_adam_specaug(dev):
    g = torch.Generator().manual_seed(4)
    n, n_reg = 1000, 600
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m, v = torch.rand(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.01
    step, lr, l2, wd, gs = 7, 3e-3, 1e-2, 1e-2, 0.5
    geff = gr * gs
    geff[:n_reg] += 2 * l2 * p[:n_reg]
    pr, mr, vr = R.adam_step(p, geff, m, v, step, lr, 0.9, 0.98, 1e-9, wd)
    pd, md, vd = p.to(dev), m.to(dev), v.to(dev)
    K.adam(pd, gr.to(dev), md, vd, n_reg, lr, step, 0.9, 0.98, 1e-9, wd, l2, gs)
    cmp(pd, pr, rtol=1e-5, atol=1e-6)
    cmp(md, mr, rtol=1e-5, atol=1e-7)
    cmp(vd, vr, rtol=1e-5, atol=1e-8)
    out = torch.zeros(1, device=dev)
    K.sumsq(pd, n_reg, out)
    cmp(out, (pd[:n_reg].cpu() ** 2).sum()[None], rtol=1e-5, atol=1e-4)
    # the optimizer also writes the bf16 shadow of the parameters: the same update, and bitwise the cast of the updated buffer
    p2, m2, v2 = p.to(dev), m.to(dev), v.to(dev)
    sh = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    K.adam(p2, gr.to(dev), m2, v2, n_reg, lr, step, 0.9, 0.98, 1e-9, wd, l2, gs, shadow=sh)
    assert torch.equal(p2, pd) and torch.equal(m2, md) and torch.equal(v2, vd)
    assert torch.equal(sh, K.cast(p2, torch.empty(n, dtype=torch.bfloat16, device=dev)))
    # specaugment
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((3, 120, 80)).astype(np.float32)
    fm, tm = R.specaugment_draw(rng, [120, 100, 60])
    ref = R.specaugment_apply(feat, fm, tm)
    x = torch.from_numpy(feat).to(dev)
    K.specaugment(x, torch.from_numpy(fm).to(dev), torch.from_numpy(tm).to(dev))
    np.testing.assert_array_equal(x.cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("lens", [[9, 9], [9, 5], [6, 1]])
def test_relattn_softmax(dev, dtype, lens):
    """scores = content + shifted positional with the per-sample rolled PE, masked query rows, softmax; then backward."""
    g = torch.Generator().manual_seed(5)
    B, H, T, dh, d = 2, 2, 9, 8, 16
    Rr = 2 * T - 1
    pe_b, table = R.relative_position_encoding(T, d, lens)
    Wp, bp = torch.randn(d, H, dh, generator=g) * 0.3, torch.randn(H, dh, generator=g) * 0.3
    qv = rt(torch.randn(B, T, H, dh, generator=g), dtype)
    content = rt(torch.randn(B, H, T, T, generator=g), dtype)
    # reference: project the per-sample PE, einsum, rel_left_shift, slice
    p_b = torch.einsum("brd,dhe->brhe", pe_b, Wp) + bp
    positional = torch.einsum("brhe,bthe->bhtr", p_b, qv)
    shifted = R.rel_left_shift(positional)[..., -T:]
    # device form: shared projected table + bias row, gathered inside the kernel
    p_ext = torch.cat([torch.einsum("rd,dhe->rhe", table, Wp) + bp, bp[None]], 0)  # [R+1,H,dh]
    pos = rt(torch.einsum("rhe,bthe->bhtr", p_ext, qv).contiguous(), dtype)  # [B,H,T,R+1]
    cr, pr = content.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    # reference scores built from the same rounded `pos` through the gather identity (checked vs `shifted` below)
    idx = torch.zeros(B, T, T, dtype=torch.long)
    for b in range(B):
        for i in range(T):
            for j in range(T):
                r = T - 1 - i + j
                idx[b, i, j] = (r + T - lens[b]) % Rr if r < 2 * lens[b] - 1 else Rr
    gathered = torch.gather(pr, 3, idx[:, None].expand(B, H, T, T))
    np.testing.assert_allclose(torch.gather(torch.einsum("rhe,bthe->bhtr", p_ext, qv), 3, idx[:, None].expand(B, H, T, T)).numpy(),
                               shifted.numpy(), atol=1e-5)
    scores = cr + gathered
    qmask = (torch.arange(T)[None] < torch.tensor(lens)[:, None])[:, None, :, None]
    probs_r = torch.softmax(torch.where(qmask, scores, torch.full_like(scores, -1e9)), -1)
    dP = rt(torch.randn(B, H, T, T, generator=g), dtype)
    probs_r.backward(dP)
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    probs = K.relattn_softmax_fwd(content.to(dev).to(dtype), pos.to(dev).to(dtype), ln, T)
    cmp(probs, probs_r.detach(), **tol(dtype, (1e-4, 1e-6), (2e-2, 4e-3)))
    dc, dpos = K.relattn_softmax_bwd(probs_r.detach().to(dev).to(dtype), dP.to(dev).to(dtype), ln, T, 2 * T)
    cmp(dc, cr.grad, **tol(dtype, (1e-4, 1e-6), (3e-2, 1e-2)))
    cmp(dpos, pr.grad, **tol(dtype, (1e-4, 1e-6), (3e-2, 1e-2)))


@pytest.mark.parametrize("dtype", DT)
def test_subsampling_pieces(dev, dtype):
    g = torch.Generator().manual_seed(6)
    B, T0, F0, C = 2, 21, 80, 32
    x = rt(torch.randn(B, T0, F0, generator=g), dtype)
    w1, b1 = torch.randn(3, 3, 1, C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1
    ref1 = R.conv2d_causal_s2(x[..., None], w1, b1)
    y1 = K.conv1_fwd(x.to(dev).to(dtype), w1.to(dev), b1.to(dev))
    assert y1.shape == ref1.shape == (B, 11, 40, C)
    cmp(y1, ref1, **tol(dtype, (1e-4, 1e-5)))
    dy1 = rt(torch.randn(*ref1.shape, generator=g), dtype)
    w1r = w1.clone().requires_grad_(True)
    R.conv2d_causal_s2(x[..., None], w1r, b1).backward(dy1)
    dw, db = torch.zeros(3, 3, 1, C, device=dev), torch.zeros(C, device=dev)
    K.conv1_bwd_weight(x.to(dev).to(dtype), dy1.to(dev).to(dtype), dw, db)
    cmp(dw, w1r.grad, rtol=1e-3, atol=2e-3)
    cmp(db, dy1.sum((0, 1, 2)), rtol=1e-3, atol=2e-3)
    # C == 256 takes the row-structured kernels (the flagship filter count)
    Cb = 256
    wb, bb = torch.randn(3, 3, 1, Cb, generator=g) * 0.3, torch.randn(Cb, generator=g) * 0.1
    xb = rt(torch.randn(3, 23, 80, generator=g), dtype)
    refb = R.conv2d_causal_s2(xb[..., None], wb, bb)
    yb = K.conv1_fwd(xb.to(dev).to(dtype), wb.to(dev), bb.to(dev))
    cmp(yb, refb, **tol(dtype, (1e-4, 1e-5)))
    dyb = rt(torch.randn(*refb.shape, generator=g), dtype)
    wbr = wb.clone().requires_grad_(True)
    R.conv2d_causal_s2(xb[..., None], wbr, bb).backward(dyb)
    dwb, dbb = torch.zeros(3, 3, 1, Cb, device=dev), torch.zeros(Cb, device=dev)
    K.conv1_bwd_weight(xb.to(dev).to(dtype), dyb.to(dev).to(dtype), dwb, dbb)
    cmp(dwb, wbr.grad, rtol=1e-3, atol=3e-3)
    cmp(dbb, dyb.sum((0, 1, 2)), rtol=1e-3, atol=3e-3)
    # conv2 = im2col + GEMM; data grad = GEMM + col2im
    x1 = rt(torch.randn(B, 11, 40, C, generator=g), dtype)
    w2, b2 = torch.randn(3, 3, C, C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    x1r, w2r = x1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    ref2 = R.conv2d_causal_s2(x1r, w2r, b2)
    dy2 = rt(torch.randn(*ref2.shape, generator=g), dtype)
    ref2.backward(dy2)
    col = K.im2col_3x3s2(x1.to(dev).to(dtype))
    w2d = w2.reshape(9 * C, C).to(dev).to(dtype)
    y2 = K.matmul(col, w2d, bias=b2.to(dev))
    cmp(y2.view(*ref2.shape), R.conv2d_causal_s2(x1, rt(w2, dtype), b2), **tol(dtype, (1e-4, 1e-4), (3e-2, 3e-2)))
    dyd = dy2.reshape(-1, C).to(dev).to(dtype)
    dcol = K.matmul(dyd, w2d, trans_b=True)
    dx1 = K.col2im_3x3s2(dcol, B, 11, 40, C)
    cmp(dx1, x1r.grad, **tol(dtype, (1e-4, 1e-4), (4e-2, 4e-2)))
    dw2 = torch.zeros(9 * C, C, device=dev)
    K.gemm(col, dyd, dw2, 9 * C, C, col.shape[0], col.stride(0), dyd.stride(0), C, trans_a=True, accumulate=True, split_k=4)
    cmp(dw2.view(3, 3, C, C), w2r.grad, **tol(dtype, (1e-4, 2e-4), (3e-2, 1e-1)))


@pytest.mark.parametrize("n", [160000, 4321, 160, 100])  # the last two: exactly one hop, shorter than one hop (a single padded frame)
def test_logmel(dev, n):
    cfg = R.conformer_config("S")
    rng = np.random.default_rng(n)
    sig = np.clip(rng.standard_normal((3, n)) * 0.1, -1, 1).astype(np.float32)
    ref = R.log_mel(sig, cfg)
    melw = R.mel_weight_matrix()
    out = K.logmel(torch.from_numpy(sig).to(dev), torch.from_numpy(R.hann_periodic(400)).to(dev), torch.from_numpy(melw).to(dev),
                   torch.from_numpy(R.mel_bands(melw)).to(dev), 160, 512, 0.97, 1e-6, torch.float32)
    assert out.shape == ref.shape
    # SURVEY 7.4 asks for <= 1e-4 abs in the log domain; the kernel transforms in f64 like the oracle (an f32 FFT leaves ~1e-7 x |frame| of
    # absolute error in every bin: 2e-4 in the log domain in near-silent bins), so what remains is the f32 mel sum and logf: 2e-6 measured
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5, rtol=0)


@pytest.mark.parametrize("nbins", [23, 40, 128])
def test_logmel_other_filterbank_sizes(dev, nbins):
    """23 / 40 bins: bands wider than the 16 taps the kernel keeps in LDS (the tail is read from the dense matrix); 128 bins: two full
    passes of 64 bins, empty bands (a mel bin between two FFT bins has no non-zero tap: log(eps))."""
    cfg = dict(R.conformer_config("S"))
    cfg["num_feature_bins"] = nbins
    rng = np.random.default_rng(nbins)
    sig = np.clip(rng.standard_normal((2, 6000)) * 0.1, -1, 1).astype(np.float32)
    ref = R.log_mel(sig, cfg)
    melw = R.mel_weight_matrix(nbins)
    out = K.logmel(torch.from_numpy(sig).to(dev), torch.from_numpy(R.hann_periodic(400)).to(dev), torch.from_numpy(melw).to(dev),
                   torch.from_numpy(R.mel_bands(melw)).to(dev), 160, 512, 0.97, 1e-6, torch.float32)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5, rtol=0)


def test_logmel_no_preemphasis_and_bf16_output(dev):
    """preemphasis = 0 (feature_extraction.py:170-175 returns the signal unchanged) and the bf16 feature map the bf16 models consume."""
    cfg = dict(R.conformer_config("S"))
    cfg["preemphasis"] = 0.0
    rng = np.random.default_rng(5)
    sig = np.clip(rng.standard_normal((2, 8000)) * 0.1, -1, 1).astype(np.float32)
    ref = R.log_mel(sig, cfg)
    melw = R.mel_weight_matrix()
    args = (torch.from_numpy(sig).to(dev), torch.from_numpy(R.hann_periodic(400)).to(dev), torch.from_numpy(melw).to(dev),
            torch.from_numpy(R.mel_bands(melw)).to(dev), 160, 512, 0.0, 1e-6)
    out = K.logmel(*args, torch.float32)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5, rtol=0)
    out16 = K.logmel(*args, torch.bfloat16)
    assert torch.equal(out16, out.to(torch.bfloat16))  # same f32 value, rounded once


def _relattn_reference(qkv, u, v, pext, lens, B, H, T, dh, scale, use_mask=True):
    """f32 reference of the fused kernel from the same (bf16-rounded) inputs, via the gather identity of attention.hip."""
    HD = H * dh
    Rr = 2 * T - 1
    q = qkv[:, :HD].view(B, T, H, dh)
    k = qkv[:, HD:2 * HD].view(B, T, H, dh)
    vv = qkv[:, 2 * HD:].view(B, T, H, dh)
    qu = (q + u.view(H, dh)).to(torch.bfloat16).float()
    qv = (q + v.view(H, dh)).to(torch.bfloat16).float()
    content = torch.einsum("bthe,bshe->bhts", qu, k)
    pos_all = torch.einsum("bthe,rhe->bhtr", qv, pext.view(2 * T, H, dh))
    idx = torch.zeros(B, T, T, dtype=torch.long)
    ii, jj = torch.meshgrid(torch.arange(T), torch.arange(T), indexing="ij")
    for b in range(B):
        r = T - 1 - ii + jj
        idx[b] = torch.where(r < 2 * lens[b] - 1, r + T - lens[b], torch.full_like(r, Rr))
    pos = torch.gather(pos_all, 3, idx[:, None].expand(B, H, T, T))
    s = (content + pos) * scale
    if use_mask:
        qmask = (torch.arange(T)[None] < torch.tensor(lens)[:, None])[:, None, :, None]
        s = torch.where(qmask, s, torch.zeros_like(s))
    lse = torch.logsumexp(s, -1)
    out = torch.einsum("bhts,bshe->bthe", torch.softmax(s, -1), vv).reshape(B * T, HD)
    return out, lse


@pytest.mark.parametrize("T,lens", [(75, [75, 75]), (130, [130, 97]), (64, [64, 1]), (200, [150, 200])])
def test_relattn_fused_forward(dev, T, lens):
    g = torch.Generator().manual_seed(T)
    B, H, dh = 2, 4, 64
    HD = H * dh
    qkv = (torch.randn(B * T, 3 * HD, generator=g) * 0.7).to(torch.bfloat16)
    u, v = torch.randn(HD, generator=g) * 0.3, torch.randn(HD, generator=g) * 0.3
    pext = (torch.randn(2 * T, HD, generator=g) * 0.7).to(torch.bfloat16)
    scale = 1.0 / 8.0
    ref_out, ref_lse = _relattn_reference(qkv.float(), u, v, pext.float(), lens, B, H, T, dh, scale)
    out, lse = K.relattn_fused_fwd(qkv.to(dev), u.to(dev), v.to(dev), pext.to(dev), torch.tensor(lens, dtype=torch.int32, device=dev), B, H, T, dh, scale)
    torch.cuda.synchronize()
    np.testing.assert_allclose(lse.cpu().numpy(), ref_lse.numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref_out.numpy(), rtol=2e-2, atol=2e-2)


def _relattn_ref_parts(qkv, u, v, pext, lens, B, H, T, dh, scale):
    """autograd-able pieces: leaves qu, kk, vv, pos_all -> out."""
    HD = H * dh
    Rr = 2 * T - 1
    q = qkv[:, :HD].view(B, T, H, dh)
    qu = (q + u.view(H, dh)).to(torch.bfloat16).float().requires_grad_(True)
    qv = (q + v.view(H, dh)).to(torch.bfloat16).float()
    k = qkv[:, HD:2 * HD].view(B, T, H, dh).clone().requires_grad_(True)
    vv = qkv[:, 2 * HD:].view(B, T, H, dh).clone().requires_grad_(True)
    pos_all = torch.einsum("bthe,rhe->bhtr", qv, pext.view(2 * T, H, dh)).detach().requires_grad_(True)
    ii, jj = torch.meshgrid(torch.arange(T), torch.arange(T), indexing="ij")
    idx = torch.zeros(B, T, T, dtype=torch.long)
    for b in range(B):
        r = T - 1 - ii + jj
        idx[b] = torch.where(r < 2 * lens[b] - 1, r + T - lens[b], torch.full_like(r, Rr))
    s = (torch.einsum("bthe,bshe->bhts", qu, k) + torch.gather(pos_all, 3, idx[:, None].expand(B, H, T, T))) * scale
    qmask = (torch.arange(T)[None] < torch.tensor(lens)[:, None])[:, None, :, None]
    s = torch.where(qmask, s, torch.zeros_like(s))
    out = torch.einsum("bhts,bshe->bthe", torch.softmax(s, -1), vv).reshape(B * T, HD)
    return out, torch.logsumexp(s, -1), qu, k, vv, pos_all


@pytest.mark.parametrize("T,lens,use_mask", [(75, [75, 75], True), (130, [130, 97], True), (64, [64, 1], True), (200, [150, 200], True),
                                             (333, [333, 20, 200], True), (130, [130, 60], False)])
def test_relattn_fused_backward_v2_no_skewed_gradient(dev, T, lens, use_mask):
    """Fused attention backward (tfasr_relattn_fused_bwd_q3 / _bwd_k / tfasr_relattn_dpext): dq = dqu + dqv and the u / v bias gradients finished
    inside the query-side kernel, the UNSKEWED dS stored, dpext accumulated by tfasr_relattn_dpext (+ the bias row's share from the query-side
    kernel): dq, du, dv, dpext, dk, dv against torch autograd of the same attention (multihead_attention.py:543-582)."""
    g = torch.Generator().manual_seed(T + 3)
    B, H, dh = len(lens), 4, 64
    HD = H * dh
    qkv = (torch.randn(B * T, 3 * HD, generator=g) * 0.7).to(torch.bfloat16)
    u, v = torch.randn(HD, generator=g) * 0.3, torch.randn(HD, generator=g) * 0.3
    pext = (torch.randn(2 * T, HD, generator=g) * 0.7).to(torch.bfloat16)
    dout = torch.randn(B * T, HD, generator=g).to(torch.bfloat16)
    scale = 1.0 / 8.0
    # reference: leaves qu, qv, k, v, pext -> out
    q = qkv.float()[:, :HD].view(B, T, H, dh)
    qu = (q + u.view(H, dh)).to(torch.bfloat16).float().requires_grad_(True)
    qv = (q + v.view(H, dh)).to(torch.bfloat16).float().requires_grad_(True)
    kk = qkv.float()[:, HD:2 * HD].view(B, T, H, dh).clone().requires_grad_(True)
    vv = qkv.float()[:, 2 * HD:].view(B, T, H, dh).clone().requires_grad_(True)
    pe = pext.float().view(2 * T, H, dh).clone().requires_grad_(True)
    Rr = 2 * T - 1
    ii, jj = torch.meshgrid(torch.arange(T), torch.arange(T), indexing="ij")
    idx = torch.zeros(B, T, T, dtype=torch.long)
    for b in range(B):
        r = T - 1 - ii + jj
        idx[b] = torch.where(r < 2 * lens[b] - 1, r + T - lens[b], torch.full_like(r, Rr))
    pos_all = torch.einsum("bthe,rhe->bhtr", qv, pe)
    s = (torch.einsum("bthe,bshe->bhts", qu, kk) + torch.gather(pos_all, 3, idx[:, None].expand(B, H, T, T))) * scale
    if use_mask:
        qmask = (torch.arange(T)[None] < torch.tensor(lens)[:, None])[:, None, :, None]
        s = torch.where(qmask, s, torch.zeros_like(s))
    s.retain_grad()
    out = torch.einsum("bhts,bshe->bthe", torch.softmax(s, -1), vv).reshape(B * T, HD)
    out.backward(dout.float())
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    qd, ud, vd, pd, dod = qkv.to(dev), u.to(dev), v.to(dev), pext.to(dev), dout.to(dev)
    o_d, lse_d = K.relattn_fused_fwd(qd, ud, vd, pd, ln, B, H, T, dh, scale, use_mask=use_mask)
    dpext = torch.zeros(2 * T, HD, dtype=torch.float32, device=dev)
    dqkv = torch.zeros(B * T, 3 * HD, dtype=torch.bfloat16, device=dev)
    du, dv = torch.zeros(HD, device=dev), torch.zeros(HD, device=dev)
    ds, dvec, qud, qvd = K.relattn_fused_bwd_q3(qd, ud, vd, pd, ln, o_d, dod, lse_d, dqkv, du, dv, dpext, B, H, T, dh, scale, use_mask=use_mask)
    qu_ref, qv_ref = K.bias2_fwd(qd, 3 * HD, ud, vd, B * T, HD)  # (the q + u / q + v the kernel writes for its two consumers; padded blocks excepted)
    K.relattn_fused_bwd_k(qd, qud, qvd, pd, ln, dod, lse_d, dvec, dqkv, B, H, T, dh, scale, use_mask=use_mask)
    K.relattn_dpext(ds, qvd, ln, dpext, B, H, T, dh, use_mask=use_mask)
    torch.cuda.synchronize()

    def close(a, b, what, tol=3e-2):
        a, b = a.float().cpu().numpy(), b.numpy()
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol * float(np.abs(b).max()), err_msg=what)

    gqu, gqv = qu.grad.reshape(B * T, HD), qv.grad.reshape(B * T, HD)
    close(dqkv[:, :HD], gqu + gqv, "dq = dqu + dqv")
    close(du, gqu.sum(0), "du")
    close(dv, gqv.sum(0), "dv")
    for b_, n_ in enumerate(lens):  # rows of live 64-query blocks
        live = -(-n_ // 64) * 64 if use_mask else T
        sl = slice(b_ * T, b_ * T + min(live, T))
        assert torch.equal(qud[sl], qu_ref[sl]) and torch.equal(qvd[sl], qv_ref[sl])
    close(dqkv[:, HD:2 * HD], kk.grad.reshape(B * T, HD), "dk")
    close(dqkv[:, 2 * HD:], vv.grad.reshape(B * T, HD), "dv")
    close(dpext, pe.grad.reshape(2 * T, HD), "dpext")
    # unskewed score gradient (scale folded in), only where the pair reads a real table row (the bias-row pairs are stored as 0)
    gs = (s.grad * scale)
    valid = (idx != Rr)[:, None].expand(B, H, T, T)
    if use_mask:  # masked query rows have constant scores: no gradient flows into them (s.grad above is taken after the fill)
        valid = valid & (torch.arange(T)[None] < torch.tensor(lens)[:, None])[:, None, :, None]
    # (rows of a 64-row query block that is entirely padding are NOT written: tfasr_relattn_dpext skips those tiles)
    close(torch.where(valid.to(dev), ds[..., :T].float(), torch.zeros((), device=dev)), gs * valid, "ds")


@pytest.mark.parametrize("dtype,C", [(torch.float32, 32), (torch.bfloat16, 64)])
@pytest.mark.parametrize("T0,F0", [(25, 80), (26, 78), (9, 6)])
def test_s2d_layout_conv2_equals_im2col_route(dev, dtype, C, T0, F0):
    """Haloed space-to-depth layout (csrc/conv2d.hip): conv1 written straight into it, halo / odd-edge slots zeroed, and conv2 as a
    GEMM over 9 K-segments of row-shifted operands == the im2col + GEMM route on the same activations (odd and even T1 / F1)."""
    g = torch.Generator().manual_seed(T0 * 100 + F0)
    B = 3
    x = torch.randn(B, T0, F0, generator=g).to(dev).to(dtype)
    w0 = (torch.randn(3, 3, 1, C, generator=g) * 0.3).to(dev)
    b0 = torch.randn(C, generator=g).to(dev)
    W = (torch.randn(9 * C, C, generator=g) * 0.1).to(dev).to(dtype)
    b1 = torch.randn(C, generator=g).to(dev)
    c1 = K.conv1_fwd(x, w0, b0)  # [B, T1, F1, C]
    T1, F1 = c1.shape[1], c1.shape[2]
    T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
    rows, slack = B * (T2 + 1) * (F2 + 1), F2 + 2
    full = torch.full(((rows + 2 * slack) * 4 * C,), float("nan"), dtype=dtype, device=dev)
    full[:slack * 4 * C] = 0
    full[(slack + rows) * 4 * C:] = 0
    s1 = full[slack * 4 * C:(slack + rows) * 4 * C].view(rows, 4 * C)
    K.conv1_fwd_s2d(x, w0, b0, s1)
    K.halo_zero(s1, B, T2, F2, 4 * C)
    K.s2d_edge_zero(s1, B, T1, F1, C)
    # the layout: element (b, t, f) at row (b, t/2+1, f/2+1), block (t%2, f%2); everything else zero
    want = torch.zeros(B, T2 + 1, F2 + 1, 2, 2, C, dtype=dtype, device=dev)
    for pt in range(2):
        for pf in range(2):
            sub = c1[:, pt::2, pf::2]
            want[:, 1:1 + sub.shape[1], 1:1 + sub.shape[2], pt, pf] = sub
    assert torch.equal(s1.view(B, T2 + 1, F2 + 1, 2, 2, C), want)
    # conv2 over the layout
    shift, blk = [], []
    for kh in range(3):
        for kw in range(3):
            dt, pt = (-1, kh) if kh < 2 else (0, 0)
            df, pf = (-1, kw) if kw < 2 else (0, 0)
            shift.append(dt * (F2 + 1) + df)
            blk.append(pt * 2 + pf)
    o = torch.empty(rows, C, dtype=dtype, device=dev)
    if dtype == torch.float32:
        base = slack * 4 * C
        for i in range(9):
            K.gemm(full[base + shift[i] * 4 * C + blk[i] * C:], W[i * C:(i + 1) * C], o, rows, C, C, 4 * C, C, C, bias=b1 if i == 0 else None,
                   accumulate=i > 0)
    else:
        tab = torch.tensor([sh * 4 * C + b * C for sh, b in zip(shift, blk)], dtype=torch.int64, device=dev)
        K.gemm(s1, W, o, rows, C, 9 * C, 4 * C, C, C, bias=b1, seg=(tab, None, C))
    ref = K.matmul(K.im2col_3x3s2(c1), W, bias=b1).view(B, T2, F2, C)
    got = o.view(B, T2 + 1, F2 + 1, C)[:, 1:, 1:]
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(got.float().cpu().numpy(), ref.float().cpu().numpy(), rtol=tol, atol=tol)


def test_depthwise_data_gradient_with_glu_backward_fused(dev):
    """tfasr_dwconv_bwd_data_glu = tfasr_dwconv_bwd_data followed by tfasr_glu_bwd (the fused launch keeps the depthwise gradient in f32)."""
    from tensorflowasr_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    B, T, C, Kw = 3, 77, 256, 31
    dy = torch.randn(B, T, C, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(Kw, C, generator=g) * 0.2).to(dev)
    gx = torch.randn(B, T, 2 * C, generator=g).to(dev).to(torch.bfloat16)
    fused = K.dwconv_bwd_data_glu(dy, w, gx)
    assert fused is not None
    ref = K.glu_bwd(gx, K.dwconv_bwd_data(dy, w))
    torch.cuda.synchronize()
    err = (fused.float() - ref.float()).abs().max().item()
    assert err <= 2e-2 * ref.float().abs().max().item() + 1e-3, err


@pytest.mark.parametrize("rows,N", [(9000, 768), (333, 512), (64, 128), (20011, 512), (15000, 1024)])
def test_layernorm_and_dense_in_one_launch(dev, rows, N):
    """tfasr_ln_dense_fwd against tfasr_layernorm_fwd + tfasr_gemm: ln / mean / rstd to the order of the row sums, the Dense output bitwise
    that of tfasr_gemm on the stored normalised rows (same bf16 operands, same k order), ragged last tile (MHSAModule / ConvModule heads, encoders/conformer.py:59-64)."""
    bf = torch.bfloat16
    d = 256
    g = torch.Generator().manual_seed(rows + N)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x, gm, bt = rnd(rows, d).to(bf), 1.0 + 0.1 * rnd(d), 0.1 * rnd(d)
    W, b = rnd(d, N, sc=1 / 16).to(bf), 0.1 * rnd(N)
    out = K.ln_dense_fwd(x, gm, bt, W, b)
    assert out is not None
    y1, ln1, mean1, rstd1 = out
    ln0, mean0, rstd0 = K.layernorm_fwd(x, gm, bt)
    cmp(ln1, ln0, rtol=1e-2, atol=1e-2)          # (the row sums are taken in another order: single bf16 ulps)
    cmp(mean1, mean0, rtol=1e-5, atol=1e-6)
    cmp(rstd1, rstd0, rtol=1e-5, atol=1e-6)
    y0 = torch.empty(rows, N, dtype=bf, device=dev)
    K.gemm(ln1, W, y0, rows, N, d, d, N, N, bias=b)   # the Dense layer on the launch's own normalised rows: bitwise
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("rows,F,p", [(9000, 1024, 0.1), (333, 512, 0.25), (20000, 1024, 0.0)])
def test_ffn_fused_forward_stores_the_backward_factor(dev, rows, F, p):
    """tfasr_ffn_fused_fwd2 with z_factor: y / ln / h bitwise those of the z-storing launch, the stored factor = swish'(z) * mask1 / (1 - p) of
    the same bf16 z, and the data gradient of the second Dense layer through it (dact = TFASR_ACT_FACTOR, no dropout term) = the one through
    z + swish' + the regenerated mask, up to the factor's bf16 rounding (FFModule backward, encoders/conformer.py:101-109)."""
    bf = torch.bfloat16
    d = 256
    g = torch.Generator().manual_seed(rows + F + 7)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x, gm, bt = rnd(rows, d).to(bf), 1.0 + 0.1 * rnd(d), 0.1 * rnd(d)
    W1, b1, W2, b2 = rnd(d, F, sc=1 / 16).to(bf), 0.1 * rnd(F), rnd(F, d, sc=1 / 32).to(bf), 0.1 * rnd(d)
    s1, s2, res = 201, 202, 0.5
    y0, ln0, _, _, z0, h0 = K.ffn_fused_fwd(x, gm, bt, W1, b1, W2, b2, res, p, s1, s2)
    y1, ln1, _, _, g1, h1 = K.ffn_fused_fwd(x, gm, bt, W1, b1, W2, b2, res, p, s1, s2, z_factor=True)
    assert torch.equal(y0, y1) and torch.equal(ln0, ln1) and torch.equal(h0, h1)
    ones = torch.ones(rows, F, dtype=bf, device=dev)
    m1 = K.dropout(ones, p, s1).float() if p > 0 else ones.float()
    zf = z0.float()
    sg = torch.sigmoid(zf)
    g_ref = sg * (1.0 + zf * (1.0 - sg)) * m1
    torch.testing.assert_close(g1.float(), g_ref, rtol=1e-2, atol=1e-2)
    dy = rnd(rows, d, sc=0.3).to(bf)
    dz0 = torch.empty(rows, F, dtype=bf, device=dev)
    dz1 = torch.empty(rows, F, dtype=bf, device=dev)
    K.gemm(dy, W2, dz0, rows, F, d, d, d, F, trans_b=True, alpha=res, dact_z=z0, dact=K.ACT_SWISH, drop_p=p, drop_seed=s1)
    K.gemm(dy, W2, dz1, rows, F, d, d, d, F, trans_b=True, alpha=res, dact_z=g1, dact=K.ACT_FACTOR)
    ref = res * (dy.float() @ W2.float().t()) * g_ref
    scale = float(ref.abs().max())
    torch.testing.assert_close(dz1.float(), ref, rtol=2e-2, atol=1e-2 * scale)
    torch.testing.assert_close(dz1.float(), dz0.float(), rtol=2e-2, atol=1e-2 * scale)


@pytest.mark.parametrize("rows,F,p", [(200, 1024, 0.1), (64, 256, 0.0), (333, 512, 0.25)])
def test_ffn_fused_fwd_matches_the_three_launch_arithmetic(dev, rows, F, p):
    """tfasr_ffn_fused_fwd (FFModule.call, encoders/conformer.py:101-109) against the same
    arithmetic written out in f32 torch on the bf16 operands, dropout masks regenerated with tfasr_dropout (a pure function of seed + index);
    ragged last tile, dropout off / on."""
    bf = torch.bfloat16
    d = 256
    g = torch.Generator().manual_seed(rows + F)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x, gm, bt = rnd(rows, d).to(bf), 1.0 + 0.1 * rnd(d), 0.1 * rnd(d)
    W1, b1, W2, b2 = rnd(d, F, sc=1 / 16).to(bf), 0.1 * rnd(F), rnd(F, d, sc=1 / 32).to(bf), 0.1 * rnd(d)
    s1, s2, s3, res = 101, 102, 103, 0.5
    out = K.ffn_fused_fwd(x, gm, bt, W1, b1, W2, b2, res, p, s1, s2)
    assert out is not None
    y, ln, mean, rstd, z, h = out
    ones = lambda n, m: torch.ones(n, m, dtype=bf, device=dev)
    m1 = K.dropout(ones(rows, F), p, s1).float() if p > 0 else torch.ones(rows, F, device=dev)   # (mask / (1 - p))
    m2 = K.dropout(ones(rows, d), p, s2).float() if p > 0 else torch.ones(rows, d, device=dev)
    xf = x.float()
    mu = xf.mean(1, keepdim=True)
    var = ((xf - mu) ** 2).mean(1, keepdim=True)
    ln_ref = ((xf - mu) * torch.rsqrt(var + 1e-3) * gm + bt)
    torch.testing.assert_close(ln.float(), ln_ref, rtol=2e-2, atol=2e-2)
    z_ref = ln.float() @ W1.float() + b1
    torch.testing.assert_close(z.float(), z_ref, rtol=2e-2, atol=2e-2)
    h_ref = torch.nn.functional.silu(z.float()) * m1
    torch.testing.assert_close(h.float(), h_ref, rtol=2e-2, atol=2e-2)
    y_ref = xf + res * ((h.float() @ W2.float() + b2) * m2)
    torch.testing.assert_close(y.float(), y_ref, rtol=2e-2, atol=3e-2)
