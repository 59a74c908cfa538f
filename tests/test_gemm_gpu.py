"""GPU numerics: the MFMA GEMM family vs a plain PyTorch fp32 (CPU) reference of the same op."""
import numpy as np
import pytest
import torch

from tensorflowasr_amd import kernels
from tensorflowasr_amd.kernels import ACT_NONE, ACT_SWISH, ACT_TANH

pytestmark = pytest.mark.gpu


def _tol(dtype):
    return (2e-2, 2e-2) if dtype == torch.bfloat16 else (1e-5, 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("mnk", [(128, 128, 64), (200, 144, 144), (77, 1000, 320), (300, 36, 250), (5, 7, 3), (256, 576, 2880),
                                 (595, 64, 600), (200, 64, 136), (130, 48, 64)])
def test_layouts(dev, dtype, ta, tb, mnk):
    M, N, K = mnk
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    Ad, Bd = A.to(dev).to(dtype), B.to(dev).to(dtype)
    ref = (Ad.float().cpu().T if ta else Ad.float().cpu()) @ (Bd.float().cpu().T if tb else Bd.float().cpu())
    out = kernels.matmul(Ad, Bd, trans_a=ta, trans_b=tb, out_dtype=torch.float32)
    torch.cuda.synchronize()
    rtol, atol = _tol(dtype)
    scale = float(np.sqrt(K))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_asymmetric_identity(dev, dtype):
    """A = I with an asymmetric B catches a row/col swap in the C write (guide rule 16)."""
    n = 160
    A = torch.eye(n)
    B = (torch.arange(n)[:, None] * 3 + torch.arange(n)[None, :] * 0.5).float() / 64.0
    out = kernels.matmul(A.to(dev).to(dtype), B.to(dev).to(dtype), out_dtype=torch.float32)
    np.testing.assert_allclose(out.cpu().numpy(), B.to(dtype).float().numpy(), rtol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_epilogue_bias_act_residual_prez(dev, dtype):
    M, N, K = 130, 200, 96
    g = torch.Generator().manual_seed(1)
    A, B = torch.randn(M, K, generator=g) * 0.3, torch.randn(K, N, generator=g) * 0.3
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd, resd = A.to(dev).to(dtype), B.to(dev).to(dtype), res.to(dev).to(dtype)
    z = Ad.float().cpu() @ Bd.float().cpu() * 0.5 + bias
    ref = resd.float().cpu() + 0.25 * torch.nn.functional.silu(z)
    prez = torch.empty(M, N, dtype=dtype, device=dev)
    out = kernels.matmul(Ad, Bd, bias=bias.to(dev), res=resd, prez=prez, alpha=0.5, beta=0.25, act=ACT_SWISH)
    rtol, atol = _tol(dtype)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol * 4)
    np.testing.assert_allclose(prez.float().cpu().numpy(), z.numpy(), rtol=rtol, atol=atol * 4)
    # backward-style epilogue: multiply by swish'(z)
    dz = kernels.matmul(Ad, Bd, dact_z=prez, dact=ACT_SWISH)
    zz = prez.float().cpu()
    s = torch.sigmoid(zz)
    ref2 = (Ad.float().cpu() @ Bd.float().cpu()) * (s * (1 + zz * (1 - s)))
    np.testing.assert_allclose(dz.float().cpu().numpy(), ref2.numpy(), rtol=rtol * 2, atol=atol * 4)
    t = kernels.matmul(Ad, Bd, act=ACT_TANH)
    np.testing.assert_allclose(t.float().cpu().numpy(), torch.tanh(Ad.float().cpu() @ Bd.float().cpu()).numpy(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_strided_heads(dev, dtype):
    """scores[b,h] = q[b,:,h,:] @ k[b,:,h,:]^T with q,k laid out [B,T,H,dh] (the attention layout)."""
    Bn, T, H, dh = 3, 70, 4, 36
    g = torch.Generator().manual_seed(2)
    q, k = torch.randn(Bn, T, H, dh, generator=g), torch.randn(Bn, T, H, dh, generator=g)
    qd, kd = q.to(dev).to(dtype), k.to(dev).to(dtype)
    out = torch.empty(Bn, H, T, T, dtype=torch.float32, device=dev)
    kernels.gemm(qd, kd, out, T, T, dh, H * dh, H * dh, T, trans_b=True, nb1=Bn, nb2=H, sA=(T * H * dh, dh),
                 sB=(T * H * dh, dh), sD=(H * T * T, T * T), alpha=0.125)
    ref = torch.einsum("bthd,bshd->bhts", qd.float().cpu(), kd.float().cpu()) * 0.125
    rtol, atol = _tol(dtype)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol * 6)


@pytest.mark.parametrize("split", [8, 5, 24])  # multiples of 8 take the slice-per-XCD workgroup mapping
def test_split_k_accumulate(dev, split):
    M, N, K = 144, 576, 5000
    g = torch.Generator().manual_seed(3)
    X, dY = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    for dtype in (torch.float32, torch.bfloat16):
        Xd, Yd = X.to(dev).to(dtype), dY.to(dev).to(dtype)
        out = torch.ones(M, N, dtype=torch.float32, device=dev)
        kernels.gemm(Xd, Yd, out, M, N, K, M, N, N, trans_a=True, accumulate=True, split_k=split)
        ref = 1.0 + Xd.float().cpu().T @ Yd.float().cpu()
        rtol, atol = _tol(dtype)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol * 70)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_product_shapes_narrow_tile(dev, dtype):
    """probs @ v and probs^T @ dctx with head size 64 and a padded score stride (the N<=64 tile variant + tr-read path)."""
    Bn, T, H, dh = 2, 75, 4, 64
    Tp = 80
    g = torch.Generator().manual_seed(5)
    probs = torch.zeros(Bn, H, T, Tp)
    probs[..., :T] = torch.rand(Bn, H, T, T, generator=g)
    qkv = torch.randn(Bn * T, 3 * H * dh, generator=g)
    pd, qd = probs.to(dev).to(dtype), qkv.to(dev).to(dtype)
    HD = H * dh
    vv = qd[:, 2 * HD:]
    att = torch.empty(Bn * T, HD, device=dev, dtype=dtype)
    kernels.gemm(pd, vv, att, T, dh, T, Tp, 3 * HD, HD, nb1=Bn, nb2=H, sA=(H * T * Tp, T * Tp), sB=(T * 3 * HD, dh), sD=(T * HD, dh))
    v4 = qd.float().cpu()[:, 2 * HD:].view(Bn, T, H, dh)
    ref = torch.einsum("bhts,bshe->bthe", pd.float().cpu()[..., :T], v4).reshape(Bn * T, HD)
    rtol, atol = _tol(dtype)
    np.testing.assert_allclose(att.float().cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol * 6)
    # dv = probs^T @ datt  (trans_a on the padded-stride score matrix)
    datt = torch.randn(Bn * T, HD, generator=g).to(dev).to(dtype)
    dv = torch.zeros(Bn * T, 3 * HD, device=dev, dtype=dtype)
    kernels.gemm(pd, datt, dv[:, 2 * HD:], T, dh, T, Tp, HD, 3 * HD, trans_a=True, nb1=Bn, nb2=H, sA=(H * T * Tp, T * Tp), sB=(T * HD, dh), sD=(T * 3 * HD, dh))
    ref = torch.einsum("bhts,bthe->bshe", pd.float().cpu()[..., :T], datt.float().cpu().view(Bn, T, H, dh)).reshape(Bn * T, HD)
    np.testing.assert_allclose(dv[:, 2 * HD:].float().cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol * 6)
    assert dv[:, :2 * HD].abs().max().item() == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,split", [(144, 576, 5000, 8), (256, 1024, 4096, 1), (40, 48, 333, 3)])
def test_weight_gradient_with_fused_bias_gradient(dev, dtype, M, N, K, split):
    """tfasr_gemm_args.colsum: gW += alpha x^T dy and gb += alpha colsum(dy) from one launch (bf16 fast path: an all-ones MFMA
    row in the first row of tiles; f32 / narrow shapes: the library falls back to a separate pass)."""
    g = torch.Generator().manual_seed(11)
    X, dY = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    Xd, Yd = X.to(dev).to(dtype), dY.to(dev).to(dtype)
    out = torch.ones(M, N, dtype=torch.float32, device=dev)
    gb = torch.full((N,), 2.0, dtype=torch.float32, device=dev)
    kernels.gemm(Xd, Yd, out, M, N, K, M, N, N, trans_a=True, accumulate=True, split_k=split, alpha=0.5, colsum=gb)
    ref = 1.0 + 0.5 * (Xd.float().cpu().T @ Yd.float().cpu())
    refb = 2.0 + 0.5 * Yd.float().cpu().sum(0)
    rtol, atol = _tol(dtype)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol * 70)
    np.testing.assert_allclose(gb.cpu().numpy(), refb.numpy(), rtol=1e-3, atol=atol * 70)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_grouped_weight_gradients_one_launch(dev, dtype):
    """tfasr_gemm_group: the Dense-layer weight gradients of a Conformer block (different shapes, one K = rows except the positional
    projection, with and without a fused bias gradient, ragged edges) as ONE launch (bf16) / one by one (f32 falls back):
    same results as separate tfasr_gemm calls."""
    g = torch.Generator().manual_seed(21)
    rows = 3000
    shapes = [(1024, 256, rows, True), (256, 1024, rows, True), (256, 256, rows, True), (256, 512, rows, False), (256, 768, rows, True),
              (256, 256, 150, False), (144, 576, rows - 7, True), (200, 72, 1000, True)]
    calls, refs, outs = [], [], []
    for i, (M, N, K, with_bias) in enumerate(shapes):
        X, dY = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
        Xd, Yd = X.to(dev).to(dtype), dY.to(dev).to(dtype)
        out = torch.full((M, N), float(i), dtype=torch.float32, device=dev)
        gb = torch.full((N,), 1.0, dtype=torch.float32, device=dev) if with_bias else None
        alpha = 0.5 if i % 2 else 1.0
        calls.append(dict(A=Xd, B=Yd, out=out, M=M, N=N, K=K, lda=M, ldb=N, ldd=N, trans_a=True, accumulate=True, alpha=alpha, colsum=gb))
        refs.append((float(i) + alpha * (Xd.float().cpu().T @ Yd.float().cpu()), None if gb is None else 1.0 + alpha * Yd.float().cpu().sum(0)))
        outs.append((out, gb))
    kernels.gemm_group(calls)
    torch.cuda.synchronize()
    rtol, atol = _tol(dtype)
    for (out, gb), (ref, refb), (M, N, K, _) in zip(outs, refs, shapes):
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol * 70, err_msg=str((M, N, K)))
        if gb is not None:
            np.testing.assert_allclose(gb.cpu().numpy(), refb.numpy(), rtol=1e-3, atol=atol * 70)


@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("mnk", [(140001, 1000, 320), (140001, 320, 1000), (133000, 256, 2304), (131072 + 77, 448, 200)])
def test_256_row_tiles_for_products_with_thousands_of_tiles(dev, tb, mnk, monkeypatch):
    """gemm_big.h: 256 x 256 / 256 x 320 tiles, one 8-wave workgroup per CU (joint projection, its data gradient, conv2).  Sizes of the
    joint network (V = 1000, J = 320), a ragged last row tile (shifted, not clamped), a K tail (1000 = 15 * 64 + 40) and a partial
    last column tile; vs torch's f32 product of the same bf16 operands and (bias case) bitwise vs the 128-row route."""
    M, N, K = mnk
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    B = (torch.randn((N, K) if tb else (K, N), generator=g) * 0.5).to(dev).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    out = kernels.matmul(A, B, trans_b=tb, bias=bias)
    ref = A.float() @ (B.float().T if tb else B.float()) + bias
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * float(np.sqrt(K)) * 0.25 + 0.05 * ref.abs().max().item() * 2 ** -7, err
    # rows of the shifted last tile are written exactly once with the same values: compare with the product of the last rows alone
    tail_rows = kernels.matmul(A[-300:].contiguous(), B, trans_b=tb, bias=bias)
    assert torch.equal(out[-300:], tail_rows)


@pytest.mark.parametrize("M", [300, 140001])
def test_data_gradient_times_tanh_prime_from_the_output(dev, M):
    """dact = TANH_OUT: dx = (dy @ W^T) * (1 - h^2) with h = tanh's OUTPUT (the joint network keeps h, not its argument); the 128-row
    route (generic epilogue) and the 256 x 320 tiles."""
    from tensorflowasr_amd.kernels import ACT_TANH_OUT
    g = torch.Generator().manual_seed(M)
    N, K = 640, 1000
    dy = (torch.randn(M, K, generator=g) * 0.3).to(dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * 0.3).to(dev).to(torch.bfloat16)
    h = torch.tanh(torch.randn(M, N, generator=g)).to(dev).to(torch.bfloat16)
    out = kernels.matmul(dy, W, trans_b=True, dact_z=h, dact=ACT_TANH_OUT)
    ref = (dy.float() @ W.float().T) * (1 - h.float() ** 2)
    torch.cuda.synchronize()
    assert (out.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item() + 1e-2


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False)])
@pytest.mark.parametrize("mnk", [(23808, 1024, 256), (23808, 256, 1024), (19200 + 8, 768, 256), (4100, 512, 2304), (12096, 256, 256)])
def test_pipeline_race_screen_bitwise_repeatable(dev, ta, tb, mnk):
    """The LDS-DMA pieces of the bf16 GEMMs go out through inline asm and are ordered by hand-counted vmcnt waits + barriers only (the
    compiler no longer drains the queue in front of the fragment reads): a mis-counted wait would show up as a tile that depends on
    timing.  Many multi-slab, multi-tile-per-workgroup products (persistent workgroups walk several tiles: the cross-tile prefetch is
    in play), each run 6 times with other work in between: every run bitwise equal, and equal to the f32 reference within bf16 rounding."""
    M, N, K = mnk
    if ta and K > 4096:
        return  # (weight-gradient layouts of that size accumulate with atomics: not bitwise repeatable by design)
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + ta + 2 * tb)
    A = (torch.randn((K, M) if ta else (M, K), generator=g) * 0.5).to(dev).to(torch.bfloat16)
    B = (torch.randn((N, K) if tb else (K, N), generator=g) * 0.5).to(dev).to(torch.bfloat16)
    noise = torch.randn(4096, 4096, device=dev)
    outs = []
    for it in range(6):
        outs.append(kernels.matmul(A, B, trans_a=ta, trans_b=tb))
        noise = noise * 1.0001 + 0.5  # unrelated traffic between the launches
        if it % 2:
            kernels.matmul(B if tb else B.t().contiguous(), B if not tb else B.t().contiguous())  # another GEMM on the same stream
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    rows = torch.randint(0, M, (64,), generator=g)
    Af = (A.float().t() if ta else A.float())[rows.to(dev)]
    ref = Af @ (B.float().t() if tb else B.float())
    np.testing.assert_allclose(outs[0][rows.to(dev)].float().cpu().numpy(), ref.cpu().numpy(), rtol=2e-2, atol=2e-2 * float(np.sqrt(K)))


@pytest.mark.parametrize("N,K", [(256, 1024), (1024, 256)])
def test_epilogues_on_the_one_tile_per_workgroup_kernel(dev, N, K):
    """gemm_fast_one_kernel (64-column tiles, one per workgroup, epilogue strips over the dead stage buffers, three workgroups per CU) takes
    every bf16 product without split-K / accumulation that has more tiles than resident slots: bias + swish + pre-activation, swish' of a
    stored pre-activation, bias + residual, and the plain NT data-gradient form at a Conformer-M shape (19 072 rows, ragged last row tile)."""
    M = 19072 - 40
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(N + K)
    A = (torch.randn(M, K, generator=g) * 0.3).to(dev).to(bf)
    B = (torch.randn(K, N, generator=g) * 0.3).to(dev).to(bf)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev).to(bf)
    rows = torch.randint(0, M, (96,), generator=g).to(dev)
    rows[:4] = torch.tensor([0, 1, M - 2, M - 1], device=dev)
    Af, Bf = A.float()[rows], B.float()
    prez = torch.empty(M, N, dtype=bf, device=dev)
    out = kernels.matmul(A, B, bias=bias, prez=prez, alpha=0.5, act=ACT_SWISH)
    z = Af @ Bf * 0.5 + bias
    tol = dict(rtol=3e-2, atol=3e-2 * float(np.sqrt(K)) * 0.3)
    np.testing.assert_allclose(prez[rows].float().cpu().numpy(), z.cpu().numpy(), **tol)
    np.testing.assert_allclose(out[rows].float().cpu().numpy(), torch.nn.functional.silu(prez[rows].float()).cpu().numpy(), rtol=2e-2, atol=2e-2)
    dz = kernels.matmul(A, B, dact_z=prez, dact=ACT_SWISH)
    zz = prez[rows].float()
    s = torch.sigmoid(zz)
    np.testing.assert_allclose(dz[rows].float().cpu().numpy(), ((Af @ Bf) * (s * (1 + zz * (1 - s)))).cpu().numpy(), **tol)
    out2 = kernels.matmul(A, B, bias=bias, res=res, beta=0.5)
    np.testing.assert_allclose(out2[rows].float().cpu().numpy(), (res[rows].float() + 0.5 * (Af @ Bf + bias)).cpu().numpy(), **tol)
    Bt = B.t().contiguous()
    out3 = kernels.matmul(A, Bt, trans_b=True)
    np.testing.assert_allclose(out3[rows].float().cpu().numpy(), (Af @ Bf).cpu().numpy(), **tol)
