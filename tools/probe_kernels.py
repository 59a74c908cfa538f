"""GPU micro-benchmarks of individual kernels (prints JSON lines). Usage: python tools/probe_kernels.py [rnnt] [gemm]"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorflowasr_amd.kernels import ACT_SWISH

from tensorflowasr_amd import kernels

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def probe_rnnt():
    for (B, T, U, V, dt) in [(32, 250, 64, 1000, torch.bfloat16), (32, 250, 64, 1000, torch.float32), (1, 743, 200, 1000, torch.float32),
                             (32, 744, 230, 1000, torch.bfloat16)]:
        lg = torch.randn(B, T, U + 1, V, device=dev, dtype=dt)
        labels = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32)
        ul = torch.full((B,), U, device=dev, dtype=torch.int32)
        tl = torch.full((B,), T, device=dev, dtype=torch.int32)
        grads = torch.empty_like(lg)
        ms = timeit(lambda: kernels.rnnt_loss_fwd_bwd(lg, labels, ul, tl, grads=grads), iters=5)
        bytes_alg = lg.numel() * lg.element_size() * 2  # logits in + grads out
        print(json.dumps({"kernel": "rnnt_loss_fwd_bwd", "shape": [B, T, U + 1, V], "dtype": str(dt), "ms": ms,
                          "alg_GBps": bytes_alg / ms / 1e6}))
        del lg, grads


def probe_gemm():
    for dt in (torch.bfloat16, torch.float32):
        for (M, N, K, ta, tb) in [(4096, 4096, 4096, False, True), (4096, 4096, 4096, False, False), (24000, 1024, 256, False, False),
                                  (24000, 256, 1024, False, False), (256, 1024, 24000, True, False), (65536, 1000, 640, False, False)]:
            A = torch.randn((K, M) if ta else (M, K), device=dev).to(dt)
            B = torch.randn((N, K) if tb else (K, N), device=dev).to(dt)
            if ta:
                out = torch.zeros(M, N, device=dev, dtype=torch.float32)
                fn = lambda: kernels.gemm(A, B, out, M, N, K, M, B.stride(0), N, trans_a=True, trans_b=tb, accumulate=True, split_k=16)
            else:
                out = torch.empty(M, N, device=dev, dtype=dt)
                fn = lambda: kernels.matmul(A, B, trans_a=ta, trans_b=tb, out=out)
            ms = timeit(fn, iters=10)
            print(json.dumps({"kernel": "gemm", "mnk": [M, N, K], "ta": ta, "tb": tb, "dtype": str(dt), "ms": ms,
                              "TFLOPs": 2.0 * M * N * K / ms / 1e9}))




def probe_modelgemm():
    """The GEMM shapes of one Conformer-M step (B=32, T'=595): (M, N, K, ta, tb, batch, count per step)."""
    dt = torch.bfloat16
    R = 19040
    shapes = [
        ("ffn1 fwd", R, 1024, 256, 0, 0, 1, 32), ("ffn2 fwd", R, 256, 1024, 0, 0, 1, 32),
        ("ffn2 dgrad", R, 1024, 256, 0, 1, 1, 32), ("ffn1 dgrad", R, 256, 1024, 0, 1, 1, 32),
        ("ffn1 wgrad", 256, 1024, R, 1, 0, 1, 32), ("ffn2 wgrad", 1024, 256, R, 1, 0, 1, 32),
        ("qkv fwd", R, 768, 256, 0, 0, 1, 16), ("proj fwd (o/pw2)", R, 256, 256, 0, 0, 1, 32), ("pw1 fwd", R, 512, 256, 0, 0, 1, 16),
        ("proj wgrad", 256, 256, R, 1, 0, 1, 32), ("qkv wgrad", 256, 768, R, 1, 0, 1, 16),
        ("attn content (NT)", 595, 595, 64, 0, 1, 128, 48), ("attn PV (NN)", 595, 64, 595, 0, 0, 128, 48), ("attn dV (TN)", 595, 64, 595, 1, 0, 128, 32),
        ("attn pos (NT)", 595, 1190, 64, 0, 1, 128, 16), ("attn dqv (NN)", 595, 64, 1190, 0, 0, 128, 16),
        ("conv2 fwd", 380800, 256, 2304, 0, 0, 1, 1), ("conv2 dgrad", 380800, 2304, 256, 0, 1, 1, 1), ("conv2 wgrad", 2304, 256, 380800, 1, 0, 1, 1),
        ("linear fwd", R, 256, 5120, 0, 0, 1, 1), ("joint vocab fwd", 590000, 1000, 640, 0, 0, 1, 1),
        ("joint vocab dgrad", 590000, 640, 1000, 0, 1, 1, 1), ("joint vocab wgrad", 640, 1000, 590000, 1, 0, 1, 1),
    ]
    tot = 0.0
    for name, M, N, Kd, ta, tb, nb, cnt in shapes:
        A = torch.randn(nb, *((Kd, M) if ta else (M, Kd)), device=dev).to(dt)
        B = torch.randn(nb, *((N, Kd) if tb else (Kd, N)), device=dev).to(dt)
        acc = bool(ta) and nb == 1
        out = torch.zeros(nb, M, N, device=dev, dtype=torch.float32 if acc else dt)
        tiles = -(-M // 128) * -(-N // 128)
        sk = 1 if (not acc or tiles >= 512 or Kd <= 2048) else int(max(1, min(-(-1024 // tiles), Kd // 1024, 64)))
        lda, ldb = A.shape[2], B.shape[2]
        fn = lambda: kernels.gemm(A, B, out, M, N, Kd, lda, ldb, N, trans_a=bool(ta), trans_b=bool(tb), nb1=nb, sA=(A.shape[1] * A.shape[2], 0),
                                  sB=(B.shape[1] * B.shape[2], 0), sD=(M * N, 0), accumulate=acc, split_k=sk)
        ms = timeit(fn, iters=10)
        tot += ms * cnt
        print(json.dumps({"gemm": name, "mnk": [M, N, Kd], "batch": nb, "split_k": sk, "us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * nb * M * N * Kd / ms / 1e9, 1),
                          "ms_per_step": round(ms * cnt, 2)}))
    print(json.dumps({"sum_ms_per_step": round(tot, 2)}))


def probe_attn():
    """fused forward vs the unfused GEMM + softmax sequence at Conformer-M bench shape."""
    import math
    B, H, T, dh = 32, 4, 595, 64
    HD = H * dh
    dt = torch.bfloat16
    qkv = torch.randn(B * T, 3 * HD, device=dev).to(dt)
    u, v = torch.randn(HD, device=dev) * 0.1, torch.randn(HD, device=dev) * 0.1
    pext = torch.randn(2 * T, HD, device=dev).to(dt)
    ln = torch.full((B,), T, dtype=torch.int32, device=dev)
    ms = timeit(lambda: kernels.relattn_fused_fwd(qkv, u, v, pext, ln, B, H, T, dh, 0.125), iters=10)
    fl = 2.0 * B * H * T * T * dh * 4  # QK, QP(window ~2x counted as 1 useful), PV
    print(json.dumps({"kernel": "relattn_fused_fwd", "ms": ms, "useful_TFLOPs": 2.0 * B * H * T * T * dh * 3 / ms / 1e9}))


def probe_epi():
    """Cost of each fused epilogue term on the ffn1-forward / ffn2-dgrad shapes."""
    dt = torch.bfloat16
    M = 23808
    x = torch.randn(M, 256, device=dev).to(dt)
    w = torch.randn(256, 1024, device=dev).to(dt)
    b = torch.randn(1024, device=dev)
    out = torch.empty(M, 1024, device=dev, dtype=dt)
    z = torch.empty(M, 1024, device=dev, dtype=dt)
    zz = torch.randn(M, 1024, device=dev).to(dt)
    variants = [("plain", {}), ("bias", dict(bias=b)), ("bias+prez", dict(bias=b, prez=z)), ("bias+swish", dict(bias=b, act=ACT_SWISH)),
                ("bias+drop", dict(bias=b, drop_p=0.1, drop_seed=1234)), ("bias+prez+swish+drop", dict(bias=b, prez=z, act=ACT_SWISH, drop_p=0.1, drop_seed=1234)),
                ("dact_z", dict(dact_z=zz, dact=ACT_SWISH)), ("dact_z+drop", dict(dact_z=zz, dact=ACT_SWISH, drop_p=0.1, drop_seed=77))]
    for name, kw in variants:
        ms = timeit(lambda: kernels.matmul(x, w, out=out, **kw), iters=20)
        print(json.dumps({"epilogue": name, "us": round(ms * 1e3, 1)}))
    h = torch.randn(M, 1024, device=dev).to(dt)
    w2 = torch.randn(1024, 256, device=dev).to(dt)
    o2 = torch.empty(M, 256, device=dev, dtype=dt)
    for name, kw in [("ffn2 plain", {}), ("ffn2 bias+res", dict(bias=b[:256], res=x, beta=0.5)), ("ffn2 bias+res+drop", dict(bias=b[:256], res=x, beta=0.5, drop_p=0.1, drop_seed=5))]:
        ms = timeit(lambda: kernels.matmul(h, w2, out=o2, **kw), iters=20)
        print(json.dumps({"epilogue": name, "us": round(ms * 1e3, 1)}))


def probe_splitk():
    dt = torch.bfloat16
    R = 19040
    for M, N in [(256, 1024), (1024, 256), (256, 256), (256, 768), (256, 512), (640, 1000)]:
        Kd = R if M != 640 else 343000
        A = torch.randn(Kd, M, device=dev).to(dt)
        B = torch.randn(Kd, N, device=dev).to(dt)
        out = torch.zeros(M, N, device=dev, dtype=torch.float32)
        for sk in (8, 12, 18, 24, 32, 48, 64, 96):
            ms = timeit(lambda: kernels.gemm(A, B, out, M, N, Kd, M, N, N, trans_a=True, accumulate=True, split_k=sk), iters=20)
            print(json.dumps({"wgrad": [M, N, Kd], "split_k": sk, "us": round(ms * 1e3, 1)}))


def probe_pointwise():
    """HBM-bound pointwise / reduction kernels at the Conformer-M bench shape: achieved GB/s of algorithmic traffic."""
    dt = torch.bfloat16
    B, T, d = 32, 744, 256
    rows = B * T
    x = torch.randn(rows, d, device=dev).to(dt)
    dy = torch.randn(rows, d, device=dev).to(dt)
    g = torch.randn(d, device=dev)
    b = torch.randn(d, device=dev)
    e2 = 2.0 * rows * d  # bytes of one [rows, d] bf16 tensor

    def rep(name, fn, nbytes):
        ms = timeit(fn, iters=20)
        print(json.dumps({"kernel": name, "us": round(ms * 1e3, 1), "GBps": round(nbytes / ms / 1e6, 0)}))

    y, mean, rstd = kernels.layernorm_fwd(x, g, b)
    rep("ln_fwd", lambda: kernels.layernorm_fwd(x, g, b), 2 * e2)
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    rep("ln_bwd(+add)", lambda: kernels.layernorm_bwd(dy, x, g, mean, rstd, dg, db, add=dy), 4 * e2)
    stats = torch.zeros(2 * d + 1, device=dev)
    rep("bn_stats", lambda: kernels.bn_stats(x, stats), e2)
    fin = torch.empty(4 * d, device=dev)
    mm, mv = torch.zeros(d, device=dev), torch.ones(d, device=dev)
    kernels.bn_finalize(stats, rows * 24, g, b, fin, mm, mv)
    rep("bn_apply_fwd(swish)", lambda: kernels.bn_apply_fwd(x, fin, ACT_SWISH), 2 * e2)
    bst = torch.zeros(2 * d, device=dev)
    rep("bn_bwd_stats", lambda: kernels.bn_bwd_stats(x, dy, fin, bst, ACT_SWISH), 2 * e2)
    rep("bn_apply_bwd", lambda: kernels.bn_apply_bwd(x, dy, fin, bst, rows, ACT_SWISH), 3 * e2)
    w = torch.randn(31, d, device=dev)
    x3, dy3 = x.view(B, T, d), dy.view(B, T, d)
    rep("dwconv_fwd", lambda: kernels.dwconv_fwd(x3, w, b), 2 * e2)
    rep("dwconv_bwd_data", lambda: kernels.dwconv_bwd_data(dy3, w), 2 * e2)
    dw = torch.zeros(31, d, device=dev)
    rep("dwconv_bwd_weight", lambda: kernels.dwconv_bwd_weight(x3, dy3, dw, db), 2 * e2)
    a2 = torch.randn(rows, 2 * d, device=dev).to(dt)
    rep("glu_fwd", lambda: kernels.glu_fwd(a2), 3 * e2)
    rep("glu_bwd", lambda: kernels.glu_bwd(a2, dy), 5 * e2)
    for C in (256, 1024):
        z = torch.randn(rows, C, device=dev).to(dt)
        o = torch.zeros(C, device=dev)
        rep(f"colsum C={C}", lambda: kernels.colsum(z, o), 2.0 * rows * C)
    rep("dropout", lambda: kernels.dropout(x, 0.1, 123), 2 * e2)
    q3 = torch.randn(rows, 3 * d, device=dev).to(dt)
    rep("bias2_fwd", lambda: kernels.bias2_fwd(q3, 3 * d, g, b, rows, d), 3 * e2)
    dq = torch.empty(rows, 3 * d, device=dev, dtype=dt)
    rep("bias2_bwd", lambda: kernels.bias2_bwd(x, dy, dq, 3 * d, dg, db, rows, d), 3 * e2)


if __name__ == "__main__":
    which = sys.argv[1:] or ["rnnt", "gemm"]
    for w in which:
        globals()["probe_" + w]()
