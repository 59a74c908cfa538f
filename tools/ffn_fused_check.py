import sys, torch, numpy as np, os
sys.path.insert(0, '.')
from tensorflowasr_amd import kernels as K
from tensorflowasr_amd.kernels import ACT_SWISH
dev = torch.device('cuda:0'); bf = torch.bfloat16
def run(rows, d=256, F=1024, p=0.1, iters=20):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(rows, d, generator=g)).to(dev).to(bf)
    gm = (1 + 0.1*torch.randn(d, generator=g)).to(dev); bt = (0.1*torch.randn(d, generator=g)).to(dev)
    W1 = (torch.randn(d, F, generator=g)/16).to(dev).to(bf); b1 = (0.1*torch.randn(F, generator=g)).to(dev)
    W2 = (torch.randn(F, d, generator=g)/32).to(dev).to(bf); b2 = (0.1*torch.randn(d, generator=g)).to(dev)
    s1, s2 = 12345, 67890
    os.environ.pop("TFASR_FFN_FUSED", None)
    got = K.ffn_fused_fwd(x, gm, bt, W1, b1, W2, b2, 0.5, p, s1, s2)
    assert got is not None
    y, ln, mean, rstd, z, h = got
    ln0, mean0, rstd0 = K.layernorm_fwd(x, gm, bt)
    z0 = torch.empty(rows, F, dtype=bf, device=dev)
    h0 = K.matmul(ln0, W1, bias=b1, act=ACT_SWISH, prez=z0, drop_p=p, drop_seed=s1)
    y0 = K.matmul(h0, W2, bias=b2, res=x, beta=0.5, drop_p=p, drop_seed=s2)
    torch.cuda.synchronize()
    def cmp(a, b, name):
        a, b = a.float(), b.float()
        err = (a - b).abs().max().item(); rel = ((a - b).norm() / b.norm()).item()
        nbad = int((a != b).sum())
        print(f"  {name:5s} maxabs {err:.3e} rel {rel:.3e} mismatching {nbad}/{a.numel()}")
        return rel
    print(f"rows {rows} p {p}")
    ok = cmp(ln, ln0, 'ln') < 1e-4 and cmp(mean, mean0, 'mean') < 1e-6 and cmp(rstd, rstd0, 'rstd') < 1e-6
    ok &= cmp(z, z0, 'z') < 1e-3
    ok &= cmp(h, h0, 'h') < 1e-2
    ok &= cmp(y, y0, 'y') < 1e-2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): K.ffn_fused_fwd(x, gm, bt, W1, b1, W2, b2, 0.5, p, s1, s2)
    e0.record()
    for _ in range(iters): K.ffn_fused_fwd(x, gm, bt, W1, b1, W2, b2, 0.5, p, s1, s2)
    e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1)/iters*1e3
    e0.record()
    for _ in range(iters):
        ln0, mean0, rstd0 = K.layernorm_fwd(x, gm, bt)
        h0 = K.matmul(ln0, W1, bias=b1, act=ACT_SWISH, prez=z0, drop_p=p, drop_seed=s1)
        y0 = K.matmul(h0, W2, bias=b2, res=x, beta=0.5, drop_p=p, drop_seed=s2)
    e1.record(); torch.cuda.synchronize()
    t3 = e0.elapsed_time(e1)/iters*1e3
    print(f"  fused {tf:.1f} us   three-launch {t3:.1f} us   (python launch overhead included in both)  ok={ok}")
for nwr in ():
    if nwr: os.environ["TFASR_FFN_NWR"] = nwr
    print("NWR", nwr or "auto")
    if nwr: break
for rows in (14784, 19072, 23776, 1000, 77):
    run(rows)
run(19072, p=0.0)
