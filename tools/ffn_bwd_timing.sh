#!/bin/bash
# per-phase cycle sums of the fused FFN backward (probe build with -DTFASR_FFN_TIMING, see tools/README.md) + event timing of both routes
TFASR_LIB=$PWD/tools/hwprobe/libtfasr_probe.so TFASR_FFN_DBG_DUMP=1 timeout 200 python - <<'PY' 2>&1 | grep -E "ffn_bwd_timing|us per" | tail -4
import sys, ctypes, torch
sys.path.insert(0, '.')
from tensorflowasr_amd import kernels as K, _lib
L = K._L()
dev = torch.device('cuda:0'); bf = torch.bfloat16
rows, d, F = 19072, 256, 1024
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).to(dev).to(bf)
dyd, z, x, add = rnd(rows, d), rnd(rows, F), rnd(rows, d), rnd(rows, d)
W1, W2 = rnd(d, F, sc=1/16), rnd(F, d, sc=1/32)
gm = torch.ones(d, device=dev); mean = torch.zeros(rows, device=dev); rstd = torch.ones(rows, device=dev)
dz = torch.empty(rows, F, dtype=bf, device=dev); dx = torch.empty(rows, d, dtype=bf, device=dev); dxd = torch.empty_like(dx)
tiles = L.tfasr_ffn_fused_bwd_tiles(rows)
part = torch.empty(tiles * 512, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def run():
    return L.tfasr_ffn_fused_bwd(p(dyd), p(z), p(W1), p(W2), p(x), p(gm), p(mean), p(rstd), p(add), p(dz), p(dx), p(dxd), p(part), rows, d, F, 0.5, 0.1, 11, 12, 1, K._stream())
for _ in range(3): assert run() == 0
torch.cuda.synchronize()
PY
