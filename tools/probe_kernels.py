"""GPU micro-benchmarks of individual kernels (prints JSON lines). Usage: python tools/probe_kernels.py [rnnt] [gemm]"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

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


if __name__ == "__main__":
    which = sys.argv[1:] or ["rnnt", "gemm"]
    for w in which:
        globals()["probe_" + w]()
