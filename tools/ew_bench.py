"""Isolated timings of the HBM-bound row kernels at the Conformer-M step's shapes (rows = 23808, d = 256):
    python tools/ew_bench.py [rows] [C] [nsets]
Every op runs over `nsets` rotating operand sets (default 8: ~0.3 GB, more than the 256 MB Infinity Cache, so the operands come from
HBM; nsets = 1 measures the cache-resident case the step often sees) and is timed with HIP events over 200 launches.  A plain
device copy of the same byte count is the reference for what the memory system delivers at this size."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tensorflowasr_amd import kernels as K  # noqa: E402


def timed(fn, nsets, iters=200):
    for i in range(10):
        fn(i % nsets)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 23808
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    nsets = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(dev).to(bf)

    X = [rnd(rows, C) for _ in range(nsets)]
    DY = [rnd(rows, C) for _ in range(nsets)]
    ADD = [rnd(rows, C) for _ in range(nsets)]
    OUT = [torch.empty(rows, C, dtype=bf, device=dev) for _ in range(nsets)]
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    _, mean, rstd = K.layernorm_fwd(X[0], gamma, beta)
    stats = torch.zeros(2 * C + 1, device=dev)
    fin = torch.empty(4 * C, device=dev)
    K.bn_stats(X[0], stats)
    K.bn_finalize(stats, rows, gamma, beta, fin, torch.zeros(C, device=dev), torch.ones(C, device=dev))
    MB = rows * C * 2 / 1e6

    def run(name, nbuf, fn):
        us = timed(fn, nsets)
        print("%-34s %8.2f us   %6.2f TB/s (%d x %.1f MB)" % (name, us, nbuf * MB / us, nbuf, MB), flush=True)

    run("copy (torch)", 2, lambda i: OUT[i].copy_(X[i]))
    run("add (torch, 2 reads 1 write)", 3, lambda i: torch.add(X[i], DY[i], out=OUT[i]))
    run("layernorm_fwd", 2, lambda i: K.layernorm_fwd(X[i], gamma, beta))
    run("layernorm_bwd", 3, lambda i: K.layernorm_bwd(DY[i], X[i], gamma, mean, rstd, dgam, dbet, dx=OUT[i]))
    run("layernorm_bwd (no dgamma/dbeta)", 3, lambda i: K.layernorm_bwd(DY[i], X[i], gamma, mean, rstd, None, None, dx=OUT[i]))
    run("layernorm_bwd + add", 4, lambda i: K.layernorm_bwd(DY[i], X[i], gamma, mean, rstd, dgam, dbet, add=ADD[i], dx=OUT[i]))
    run("bn_stats", 1, lambda i: K.bn_stats(X[i], stats))
    run("bn_bwd_stats (swish)", 2, lambda i: K.bn_bwd_stats(X[i], DY[i], fin, stats[:2 * C], K.ACT_SWISH))
    run("bn_apply_fwd (swish)", 2, lambda i: K.bn_apply_fwd(X[i], fin, K.ACT_SWISH, y=OUT[i]))
    run("bn_apply_bwd (swish)", 3, lambda i: K.bn_apply_bwd(X[i], DY[i], fin, stats[:2 * C], rows, K.ACT_SWISH, dx=OUT[i]))
    run("add_act_fwd", 3, lambda i: K.add_act_fwd(X[i], DY[i], K.ACT_SWISH))
    X3 = [x.view(32, rows // 32, C) for x in X]
    D3 = [x.view(32, rows // 32, C) for x in DY]
    for ks in (31, 5):
        w = torch.randn(ks, C, device=dev) * 0.1
        dw = torch.zeros(ks, C, device=dev)
        run("dwconv_fwd k=%d" % ks, 2, lambda i: K.dwconv_fwd(X3[i], w, None))
        run("dwconv_bwd_data k=%d" % ks, 2, lambda i: K.dwconv_bwd_data(D3[i], w))
        run("dwconv_bwd_weight k=%d" % ks, 2, lambda i: K.dwconv_bwd_weight(X3[i], D3[i], dw, None))


if __name__ == "__main__":
    main()
