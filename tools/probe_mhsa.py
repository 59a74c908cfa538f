"""Run only the MHSA module forward+backward of Conformer-M (B=32, T'=595) a few times (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorflowasr_amd import configs
from tensorflowasr_amd.conformer import ConformerTransducer

dev = torch.device("cuda:0")
cfg = configs.conformer_m()
cfg.num_blocks = 1
m = ConformerTransducer(cfg, dev, dtype=torch.bfloat16)
B, T = 32, 595
x = torch.randn(B * T, cfg.dmodel, device=dev).to(torch.bfloat16)
elen = torch.full((B,), T, dtype=torch.int32, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "mhsa"
for it in range(6):
    ctx = {}
    if which == "mhsa":
        y = m._mhsa_fwd(x, "enc/block0/mhsa/", B, T, elen, ctx, 18, True)
        dx = m._mhsa_bwd(y, "enc/block0/mhsa/", B, T, elen, ctx)
    elif which == "ffm":
        y = m._ffm_fwd(x, "enc/block0/ff1/", ctx, 16, True)
        dx = m._ffm_bwd(y, "enc/block0/ff1/", ctx)
    else:
        y = m._convm_fwd(x, "enc/block0/conv/", B, T, True, ctx, 19)
        dx = m._convm_bwd(y, "enc/block0/conv/", B, T, ctx)
torch.cuda.synchronize()
print("done")
