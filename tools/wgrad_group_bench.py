"""One grouped weight-gradient launch of a Conformer-M block (8 products, d = 256, dff = 1024) at several row counts: the slope is the main
loop (3.9 us per 1000 rows = 780 TFLOP/s), the intercept (~18 us) the launch + the f32-atomic epilogue.  python tools/wgrad_group_bench.py"""
import sys, torch
sys.path.insert(0, '.')
from tensorflowasr_amd import kernels as K
dev = torch.device('cuda:0')
bf = torch.bfloat16
def run(rows, with_bias=True, iters=30):
    g = torch.Generator().manual_seed(0)
    shapes = [(256,1024),(1024,256),(256,1024),(1024,256),(256,768),(256,256),(256,512),(256,256)]
    xs = [ (torch.randn(rows, m, generator=g)*0.1).to(dev).to(bf) for m,n in shapes]
    dys = [ (torch.randn(rows, n, generator=g)*0.1).to(dev).to(bf) for m,n in shapes]
    outs = [ torch.zeros(m, n, device=dev) for m,n in shapes]
    bs = [ torch.zeros(n, device=dev) for m,n in shapes]
    calls = [dict(A=xs[i], B=dys[i], out=outs[i], M=shapes[i][0], N=shapes[i][1], K=rows, lda=shapes[i][0], ldb=shapes[i][1], ldd=shapes[i][1],
                  trans_a=True, accumulate=True, split_k=8, colsum=bs[i] if with_bias else None) for i in range(len(shapes))]
    for _ in range(3): K.gemm_group(calls)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): K.gemm_group(calls)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = sum(2.0*rows*m*n for m,n in shapes)
    print("rows %6d bias %d: %7.1f us  %6.0f TFLOP/s" % (rows, with_bias, us, fl/us/1e6))
for rows in (2048, 4768, 9536, 19072, 38144):
    run(rows)
run(19072, False)
