#!/bin/bash
# per-phase cycle sums of the fused FFN forward (probe build with -DTFASR_FFN_TIMING, see tools/README.md)
for n in 2; do for rows in 19072 14784; do
echo "=== MR $n rows $rows"
TFASR_LIB=$PWD/tools/hwprobe/libtfasr_probe.so TFASR_FFN_MR=$n TFASR_FFN_DBG_DUMP=1 timeout 120 python - <<PY 2>&1 | grep ffn_timing | tail -2
import sys, torch
sys.path.insert(0, '.')
from tensorflowasr_amd import kernels as K
dev = torch.device('cuda:0'); bf = torch.bfloat16
rows, d, F = $rows, 256, 1024
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, d, generator=g).to(dev).to(bf)
gm = torch.ones(d, device=dev); bt = torch.zeros(d, device=dev)
W1 = (torch.randn(d, F, generator=g)/16).to(dev).to(bf); b1 = torch.zeros(F, device=dev)
W2 = (torch.randn(F, d, generator=g)/32).to(dev).to(bf); b2 = torch.zeros(d, device=dev)
for _ in range(4): K.ffn_fused_fwd(x, gm, bt, W1, b1, W2, b2, 0.5, 0.1, 1, 2)
torch.cuda.synchronize()
PY
done; done
