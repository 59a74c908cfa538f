"""Where the HOST spends its time enqueueing a data-parallel train step (one-rank RCCL group on one GPU): cProfile over 20 steps.
    python tools/dp_host_profile.py [M|S] [dp|nodp]"""
import cProfile
import importlib.util
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
size = sys.argv[1] if len(sys.argv) > 1 else "M"
use_dp = (sys.argv[2] if len(sys.argv) > 2 else "dp") == "dp"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
import torch
import torch.distributed as dist

from tensorflowasr_amd import configs, dp as dpmod
from tensorflowasr_amd.conformer import ConformerTransducer

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dp = None
if use_dp:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dp = dpmod.DataParallel()
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
sys.argv = ["x"]
spec.loader.exec_module(b)
cfg = configs.conformer_m() if size == "M" else configs.conformer_s()
model = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0, dp=dp)
if dp:
    dp.attach(model.ps.grad)
data = [b.to_train_data(b.make_batch(cfg, 32, seed=10 + 13 * i, padding="batch", size="LibriSpeech-shaped"), dev) for i in range(2)]
for i in range(5):
    model.train_step(data[i % 2])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(20):
    model.train_step(data[i % 2])
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{size} dp={use_dp}: host enqueue {t_host / 20 * 1e3:.2f} ms/step (under cProfile), wall {dt / 20 * 1e3:.2f} ms/step")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
