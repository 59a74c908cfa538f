"""Fused Dense data gradient + LayerNorm backward (tfasr_dense_ln_bwd) against the two launches it replaces, HIP-event timed:
python tools/dense_ln_timing.py   (rows of the two bench batches x the three K of a Conformer-M block)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorflowasr_amd import kernels as K

dev = torch.device("cuda", 0)
dt, d = torch.bfloat16, 256
for rows in (14784, 19264, 23776):
    for Kd in (1024, 768, 512):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(rows, d, generator=g).to(dev).to(dt)
        W = (torch.randn(d, Kd, generator=g) / 16).to(dev).to(dt)
        dy = torch.randn(rows, Kd, generator=g).to(dev).to(dt)
        add = torch.randn(rows, d, generator=g).to(dev).to(dt)
        gam, bet = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
        _, mean, rstd = K.layernorm_fwd(x, gam, bet)
        dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        dropped = torch.empty_like(x)
        dln = torch.empty(rows, d, dtype=dt, device=dev)

        def fused():
            K.dense_ln_bwd(dy, W, x, gam, mean, rstd, dg, db, add=add, dropped=dropped, drop_p=0.1, drop_seed=5)

        def two():
            K.gemm(dy, W, dln, rows, d, Kd, Kd, Kd, d, trans_b=True)
            K.layernorm_bwd_fold(dln, x, gam, mean, rstd, dg, db, add=add)

        res = []
        for f in (fused, two):
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                f()
            b.record()
            torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 50 * 1e3)
        print(f"rows {rows} K {Kd}: fused (+ fold) {res[0]:.1f} us   gemm + ln_bwd (+ fold) {res[1]:.1f} us")
