"""One-rank RCCL sanity run of the data-parallel hooks (the only NCCL configuration a 1-GPU box allows): process-group init with
device_id, sync-BN statistic all-reduce, bucketed async gradient all-reduce + wait, scalar reductions, on a tiny model."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
import numpy as np
import torch
import torch.distributed as dist

from tensorflowasr_amd import configs, dp as dpmod
from tensorflowasr_amd.conformer import ConformerTransducer

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dp = dpmod.DataParallel(bucket_bytes=1 << 16)
cfg = configs.conformer_tiny()
ref = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
mod = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0, dp=dp)
dp.attach(mod.ps.grad)
import importlib.util

spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
b = importlib.util.module_from_spec(spec)
sys.argv = ["x"]
spec.loader.exec_module(b)
data = b.to_train_data(b.make_batch(cfg, 4, seed=3, padding="batch", size="S-10s"), dev)
for m in (ref, mod):
    m.cfg.dropout = 0.0
    m.cfg.time_masking, m.cfg.freq_masking = {}, {}
l0 = ref.train_step(data)["loss"].float().cpu().numpy()
l1 = mod.train_step(data)["loss"].float().cpu().numpy()
t = torch.ones(8, device=dev)
dp.allreduce_stats_(t)
dp.barrier()
mx = dp.max_scalar(1.5, dev)
torch.cuda.synchronize()
same = bool(torch.allclose(ref.ps.flat, mod.ps.flat, rtol=1e-3, atol=1e-4))
print("nccl selfcheck:", "loss", np.round(l0, 3), np.round(l1, 3), "params match", same, "max", mx, "stats", float(t.sum()))
assert np.allclose(l0, l1, rtol=1e-3) and same and mx == 1.5 and float(t.sum()) == 8.0
dist.destroy_process_group()
print("nccl selfcheck ok")
