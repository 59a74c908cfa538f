"""Which operations of a train step make the host wait for the GPU: torch's sync-debug mode (warn) over three steps.
    python tools/sync_trace.py [M|S]"""
import importlib.util
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tensorflowasr_amd import configs
from tensorflowasr_amd.conformer import ConformerTransducer

dev = torch.device("cuda", 0)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
size = sys.argv[1] if len(sys.argv) > 1 else "M"
sys.argv = ["x"]
spec.loader.exec_module(b)
cfg = configs.conformer_m() if size == "M" else configs.conformer_s()
model = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
data = [b.to_train_data(b.make_batch(cfg, 32, seed=10 + 13 * i, padding="batch", size="LibriSpeech-shaped"), dev) for i in range(2)]
for i in range(4):
    model.train_step(data[i % 2])
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for i in range(2):
        model.train_step(data[i % 2])
torch.cuda.set_sync_debug_mode("default")
print(len(w), "synchronizing operations in 2 steps")
import traceback
for x in w[:12]:
    print("--", str(x.message)[:120], "at", x.filename.split("/")[-1], x.lineno)
