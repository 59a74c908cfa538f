"""Host-side cost of queueing one train step from an EMPTY stream (no back-pressure): python tools/host_enqueue.py [M|S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
from tensorflowasr_amd import configs
from tensorflowasr_amd.conformer import ConformerTransducer
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
b = importlib.util.module_from_spec(spec); sys.argv = [sys.argv[0]] + sys.argv[1:]; spec.loader.exec_module(b)
which = sys.argv[1] if len(sys.argv) > 1 else "M"
dev = torch.device("cuda", 0)
cfg = configs.conformer_m() if which == "M" else configs.conformer_s()
m = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
data = b.to_train_data(b.make_batch(cfg, 32, seed=10, padding="batch", size="S-10s" if which == "S" else "LibriSpeech-shaped"), dev)
for _ in range(5):
    m.train_step(data)
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.train_step(data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print(f"{which}: host enqueue {1e3 * sum(enq) / len(enq):.2f} ms per step from an empty stream; step incl. drain {1e3 * sum(tot) / len(tot):.2f} ms")
