"""RCCL sanity run of the data-parallel hooks: process-group init with device_id, sync-BN statistic all-reduce, bucketed async
gradient all-reduce + wait, scalar reductions, on a tiny model.

    python tools/nccl_selfcheck.py [--gpus N]      N > 1 re-executes itself under torch.distributed.run (one rank per GPU);
                                                   every rank asserts world == N.  N = 1 is the only configuration a 1-GPU box allows.
Every rank trains on the SAME utterances, so the all-reduced mean gradient (and the synchronised BatchNorm moments) must equal the
single-process ones: parameters after one step are compared against an un-distributed twin model."""
import os
import sys

N = int(sys.argv[sys.argv.index("--gpus") + 1]) if "--gpus" in sys.argv else 1
if N > 1 and "WORLD_SIZE" not in os.environ:
    import socket
    import subprocess

    _s = socket.socket()
    _s.bind(("127.0.0.1", 0))
    _port = _s.getsockname()[1]
    _s.close()
    raise SystemExit(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={N}", "--master-addr", "127.0.0.1",
                                     "--master-port", str(_port), os.path.abspath(__file__), "--gpus", str(N)],
                                    env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
import numpy as np
import torch
import torch.distributed as dist

from tensorflowasr_amd import configs, dp as dpmod
from tensorflowasr_amd.conformer import ConformerTransducer

RANK, WORLD, LOCAL = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
assert WORLD == N, f"--gpus {N} but WORLD_SIZE={WORLD}"
torch.cuda.set_device(LOCAL)
dev = torch.device("cuda", LOCAL)
dist.init_process_group("nccl", rank=RANK, world_size=WORLD, device_id=dev)
assert dist.get_world_size() == N
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
print(f"nccl selfcheck rank {RANK}/{WORLD}:", "loss", np.round(l0, 3), np.round(l1, 3), "params match", same, "max", mx, "stats", float(t.sum()))
assert np.allclose(l0, l1, rtol=1e-3) and same and mx == 1.5 and float(t.sum()) == 8.0 * WORLD
dist.destroy_process_group()
print(f"nccl selfcheck ok (world {WORLD})")
